#!/bin/bash
# A/B of two builds of the library on one box: bench.py with each (CILQR_LIB selects the build the Python loader takes), twice, in turns.
#   bash tools/ab_lib.sh lib/libcilqr_hip.so lib/libcilqr_variant.so ["--workload c5" "--batch 65536" ...]
set -e
P=uncertainty-aware-cilqr-for-trajectory-optimization_amd
A=${1:-$P/lib/libcilqr_hip.so}
B=${2:-$P/lib/libcilqr_variant.so}
shift 2 || true
if [ $# -eq 0 ]; then set -- "--workload c5" "--batch 65536" "--batch 16384"; fi
for rep in 1 2; do
for lib in $A $B; do
  for wl in "$@"; do
    CILQR_LIB=$PWD/$lib python bench.py $wl --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$wl', 'value %.0f' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])"
  done
done
done
