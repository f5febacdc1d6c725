# What the driver runs at the end of a round, in one gpurun call: the GPU suite, smoke(), the default bench line.
#   gpurun -- bash tools/final_check.sh
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu.log 2>&1; tail -1 gpurun_out/final/gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/final/bench.json").read().strip().split("\n")[-1])
print(j["metric"], round(j["value"]/1e6,1), round(j["ms_per_step"],4), round(j["roofline"]["frac"],3), j["roofline"]["traffic"], j["parity"]["ok"], round(j["cpu_baseline"]["value"]))
PY
