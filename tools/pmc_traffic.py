"""HBM-side bytes per launch from rocprofv3 --pmc passes, filed under the hash of the kernel sources
so that bench.py can quote `roofline.traffic` for THIS build only (profiles/pmc_traffic.json).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/f -- python3 tools/run_frozen.py c4 4 filtered
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/w -- python3 tools/run_frozen.py c4 4 filtered
    python tools/pmc_traffic.py c4 out/f out/w

FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
reads (MI355X_MICROARCH.md, HBM): doubled here.  Steady-state launches only (the first launch of a
kernel also pays cold caches: dropped when there are at least three)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KEYS = {"sweep": "sweep4_i8_kernel<0", "prepass": "sweep4_i8_kernel<1", "subset_exact": "subset_exact_",
        "segsum": "segsum_kernel", "bmu_dma": "bmu_dma_kernel", "smooth_gemm": "smooth_gemm_kernel",
        "prune": "prune_mark_kernel", "proto_gap": "proto_gap_kernel", "refine": "refine_i8_kernel",
        "pair_exact": "pair_exact_kernel"}


def per_kernel(root, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "dbgsom" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: (sum(v[1:]) / len(v[1:]) if len(v) >= 3 else sum(v) / len(v)) for k, v in acc.items()}


def main():
    workload, fdir, wdir = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    tag = os.environ.get("CAMPAIGN_TAG", "r04")
    entry = {"source": f"profiles/{tag}_{workload}_pmc_traffic.txt"}
    lines = [f"# workload {workload}, build {bench.source_hash()}: HBM bytes per launch = FETCH_SIZE KiB x 1024 x 2 "
             f"(gfx950 correction) + WRITE_SIZE KiB x 1024"]
    total = 0.0
    for name in sorted(set(fetch) | set(write)):
        b = fetch.get(name, 0.0) * 1024 * 2 + write.get(name, 0.0) * 1024
        total += b
        lines.append(f"{name[-64:]:66s} fetch {fetch.get(name, 0.0):12.1f} KiB  write {write.get(name, 0.0):12.1f} KiB  "
                     f"-> {b / 1e9:8.4f} GB")
    for key, sub in KEYS.items():
        # the exact stage is three launches (list-length classes): their sum is the stage's traffic
        hits = [n for n in set(fetch) | set(write) if sub in n]
        if hits:
            entry[key] = sum(fetch.get(n, 0.0) * 2048 + write.get(n, 0.0) * 1024 for n in hits)
    lines.append(f"# sum over one launch of every kernel: {total / 1e9:.3f} GB")
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    table.setdefault(bench.source_hash(), {})[workload] = entry
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    open(os.path.join(ROOT, "profiles", f"{tag}_{workload}_pmc_traffic.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
