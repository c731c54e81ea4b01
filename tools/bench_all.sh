#!/bin/bash
# Runs every bench workload once and leaves one JSON line per workload under gpurun_out/bench_all/ (copied to profiles/ by hand).
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/bench_all
rm -rf $OUT && mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" 2>$OUT/$name.err | grep "^{" > $OUT/$name.json || echo "FAILED $name"; echo "$name: $(python -c "import json;d=json.load(open('$OUT/$name.json'));print(round(d['value'],1),d['unit'],'kernel_ms',round(d['roofline']['kernel_ms'],4))" 2>/dev/null)"; }
run c2
run c3 --workload c3 --steps 10 --warmup 2
run c3_materialised --workload c3 --materialised --steps 10 --warmup 2 --no-cpu-baseline
run c5 --workload c5 --steps 10 --warmup 2
run c1 --workload c1
run warp --workload warp --steps 100 --warmup 10
run warp_k16 --workload warp --frames 16 --steps 50 --warmup 5 --no-cpu-baseline
run blur --workload blur
run occ --workload occ --steps 30 --warmup 3
run frame --workload frame
run frame_1024 --workload frame --batch 1024 --steps 20 --warmup 3
run plan --workload plan
run plan_65536 --workload plan --batch 65536 --steps 20 --warmup 3
for b in 2048 4096 8192 16384 65536; do run c2_B$b --workload c2 --batch $b --steps 10 --warmup 2 --no-cpu-baseline; done
