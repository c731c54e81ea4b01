#!/bin/bash
# Kernel family crossover: config-2 scenes at batch sizes around and above one solve per SIMD, grouped family (default choice)
# against the one-wavefront-per-solve family forced (CILQR_FORCE_G=64).  Prints kernel ms per launch.
for b in 2048 4096 6144 8192 12288 16384 32768; do
  d=$(python bench.py --workload c2 --batch $b --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['roofline']['kernel_ms'],4), d['roofline']['kernel'])")
  w=$(CILQR_FORCE_G=64 python bench.py --workload c2 --batch $b --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['roofline']['kernel_ms'],4))")
  echo "B=$b default: $d | wavefront family: $w"
done
