#!/bin/bash
for b in 12288 16384; do
  d=$(python bench.py --workload c2 --batch $b --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['roofline']['kernel_ms'],4), d['roofline']['kernel'])")
  w=$(CILQR_FORCE_G=64 python bench.py --workload c2 --batch $b --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['roofline']['kernel_ms'],4))")
  echo "c2 B=$b default: $d | wavefront family: $w"
done
for b in 2048 4096 8192; do
  d=$(python bench.py --workload c5 --batch $b --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['roofline']['kernel_ms'],4), d['roofline']['kernel'])")
  w=$(CILQR_FORCE_G=64 python bench.py --workload c5 --batch $b --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['roofline']['kernel_ms'],4))")
  echo "c5 B=$b default: $d | wavefront family: $w"
done
