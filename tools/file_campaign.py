"""Copy the results of tools/campaign.sh (gpurun_out/camp) into profiles/ under a version tag and
print the numbers DESIGN.md quotes.  usage: python tools/file_campaign.py v4"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1]
C = os.path.join(ROOT, "gpurun_out", "camp")
P = os.path.join(ROOT, "profiles")
for old in glob.glob(os.path.join(P, "r02_v[0-9]*_*")):
    if "_v1_" not in old and "_v4_" not in old and f"_{tag}_" not in old:   # v1: before the float32 epilogue; v4: last build whose default path swept
        os.remove(old)
for w in ("c4", "c3", "c2", "c5"):
    shutil.copy(os.path.join(C, f"r02_{w}_pmc_traffic.txt"), P)
    shutil.copy(os.path.join(C, f"{w}_bench.json"), os.path.join(P, f"r02_{tag}_{w}_bench.json"))
shutil.copy(os.path.join(C, "c4_bench_via_ctx.json"), os.path.join(P, f"r02_{tag}_c4_bench_via_ctx.json"))
stats = sorted(glob.glob(os.path.join(C, "stats_c4", "runc", "*_kernel_stats.csv")), key=os.path.getmtime)[-1]
shutil.copy(stats, os.path.join(P, f"r02_{tag}_c4_kernel_stats.csv"))
for w in ("c4", "c2"):
    shutil.copy(os.path.join(C, f"epoch_trace_{w}.txt"), os.path.join(P, f"r02_{tag}_{w}_epoch_trace.txt"))
for w in ("c4", "c3"):
    shutil.copy(os.path.join(C, f"sq_{w}_summary.txt"), os.path.join(P, f"r02_{tag}_{w}_pmc_sq_summary.txt"))
t = json.load(open(os.path.join(C, "pmc_traffic.json")))
h = bench.source_hash()
json.dump({h: t[h]}, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print("build", h)
for w in ("c4", "c3", "c2", "c5"):
    js = json.loads(open(os.path.join(C, f"{w}_bench.json")).read().strip().splitlines()[-1])
    assert js["build"] == h
    print(w, round(js["value"] / 1e6, 1), "M", round(js["ms_per_step"], 3), "ms",
          {k: round(v, 3) for k, v in js["phases_ms"].items()})
    for r in js["rooflines"]:
        print("   ", r["stage"][:34].ljust(34), "frac", round(r["frac"], 3), "exec",
              r.get("executed_frac") and round(r["executed_frac"], 3), "ms", round(r["kernel_ms"], 3), "traffic GB",
              r.get("traffic") and round(r["traffic"] / 1e9, 3), "achieved", round(r["achieved"], 1))
    fa = js["fine_phase"]["auto"]
    print("    exact", round(js["exact"]["value"] / 1e6, 2), round(js["exact"]["ms_per_step"], 2),
          round(js["exact"]["roofline"]["frac"], 3), "fine", round(fa["value"] / 1e6, 1), round(fa["ms_per_step"], 3),
          fa.get("candidates_per_workgroup", {}).get("mean"), js["fine_phase"]["prototypes_identical"],
          js["exact"]["prototypes_identical_to_headline"])
js = json.loads(open(os.path.join(C, "c4_bench.json")).read().strip().splitlines()[-1])
for k, v in js["other_data"].items():
    print(k, {a: (round(v[a]["value"] / 1e6, 1), round(v[a]["ms_per_step"], 3),
                  v[a].get("candidates_per_workgroup", {}).get("mean"), v[a].get("sweep_planes"))
              for a in ("auto", "filtered", "exact")}, v["prototypes_identical"])
cb = js["cpu_baseline"]
print("cpu", cb["value"], cb["cores"], cb["value_reference_faithful_smoothing"], cb["smoothing_s"]["broadcast_MMd"],
      js["gpu_vs_cpu"])
ctx = json.loads(open(os.path.join(C, "c4_bench_via_ctx.json")).read().strip().splitlines()[-1])
print("via ctx", ctx["value"] / 1e6, ctx["ms_per_step"], ctx["torch_imported"], ctx["host_s"])
for r in csv.DictReader(open(os.path.join(P, f"r02_{tag}_c4_kernel_stats.csv"))):
    n = r["Name"].split("(")[0][-50:]
    if any(k in n for k in ("sweep4", "subset_exact", "segsum", "bmu_dma")):
        print(f"{n:52s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:9.1f} us")
for w in ("c4", "c2"):
    print(w, [ln for ln in open(os.path.join(P, f"r02_{tag}_{w}_epoch_trace.txt")) if "epoch span" in ln][0].strip())
