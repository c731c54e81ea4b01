#!/bin/bash
# The bench lines of a round's record, one after the other on one box: gpurun_out/bench/<name>.json (copy into profiles/rNN_bench_<name>.json).
set -e
mkdir -p gpurun_out/bench
O=gpurun_out/bench
python bench.py > $O/c2.json 2> $O/c2.err
python bench.py > $O/c2_second_run.json 2>/dev/null
python bench.py --workload c1 --no-cpu-baseline > $O/c1.json 2> $O/c1.err
for b in 2048 4096 8192 16384 32768 65536; do python bench.py --batch $b --no-cpu-baseline > $O/c2batch$b.json 2>/dev/null; done
python bench.py --batch 2048 --ticks --no-cpu-baseline > $O/c2batch2048_ticks.json 2>/dev/null
python bench.py --batch 4096 --ticks --no-cpu-baseline > $O/c2batch4096_ticks.json 2>/dev/null
python bench.py --workload c3 --no-cpu-baseline > $O/c3.json 2>/dev/null
python bench.py --workload c3 --repeat-batch --no-cpu-baseline > $O/c3_repeat.json 2>/dev/null
python bench.py --workload c5 --no-cpu-baseline > $O/c5.json 2>/dev/null
python bench.py --workload c5 --ticks --no-cpu-baseline > $O/c5_ticks.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/c2_torchrun_world1.json
for w in blur occ frame plan; do python bench.py --workload $w --no-cpu-baseline > $O/$w.json 2>/dev/null; done
for k in 1 16 64; do python bench.py --workload warp --frames $k --no-cpu-baseline > $O/warp_k$k.json 2>/dev/null; done
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], "value %.6g %s" % (d["value"], d["unit"]), "ms_per_step %.4f" % d["ms_per_step"], r.get("kernel"), "kernel_ms", r.get("kernel_ms"), d.get("max_abs_du_vs_oracle"))
PY
done
