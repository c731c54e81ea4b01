set -e
mkdir -p gpurun_out/bench
python bench.py > gpurun_out/bench/c2.json 2> gpurun_out/bench/c2.err
python bench.py --workload c1 --no-cpu-baseline > gpurun_out/bench/c1.json 2> gpurun_out/bench/c1.err
python bench.py --batch 2048 --no-cpu-baseline > gpurun_out/bench/c2batch2048.json 2>/dev/null
python bench.py --batch 2048 --ticks --no-cpu-baseline > gpurun_out/bench/c2batch2048_ticks.json 2>/dev/null
python bench.py --batch 4096 --no-cpu-baseline > gpurun_out/bench/c2batch4096.json 2>/dev/null
python bench.py --batch 4096 --ticks --no-cpu-baseline > gpurun_out/bench/c2batch4096_ticks.json 2>/dev/null
python bench.py --workload c3 --no-cpu-baseline > gpurun_out/bench/c3.json 2>/dev/null
python bench.py --workload c5 --no-cpu-baseline > gpurun_out/bench/c5.json 2>/dev/null
for f in c2 c1 c2batch2048 c2batch2048_ticks c2batch4096 c2batch4096_ticks c3 c5; do python - <<PY
import json
d=json.loads(open("gpurun_out/bench/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("kernel_ms"), d.get("max_abs_du_vs_oracle"))
PY
done
