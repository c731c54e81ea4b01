#!/bin/bash
# Profiles the widened-row workloads on the GPU box: kernel-trace stats, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes
# (never combined with other trace domains).  Writes raw output under gpurun_out/prof_wide/ and a summary next to it.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_wide
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in warp occ frame plan; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${w}_stats -o ${w} -- python3 $ROOT/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${w}_stats.log 2>&1
  echo "stats $w done"
done
# the warp with 16 frames per launch (cilqr_warp_costmap_batch_device)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/warp16_stats -o warp16 -- python3 $ROOT/bench.py --workload warp --frames 16 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/warp16_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/warp16_pmc_$c -o warp16 -- python3 $ROOT/bench.py --workload warp --frames 16 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/warp16_pmc_$c.log 2>&1
done
echo "warp16 done"
for w in warp occ; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/${w}_pmc_$c -o ${w} -- python3 $ROOT/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $OUT/${w}_pmc_$c.log 2>&1
    echo "pmc $w $c done"
  done
done
cd $ROOT
python3 tools/prof_summary.py "${PROF_TITLE:-r02 — widened rows: warp (config 4; single frame and 16 frames per launch), occupancy conversions, one-call frame, batched LocalPlanner}" \
  $OUT/warp_stats $OUT/warp16_stats $OUT/warp16_pmc_FETCH_SIZE $OUT/warp16_pmc_WRITE_SIZE $OUT/occ_stats $OUT/frame_stats $OUT/plan_stats $OUT/warp_pmc_FETCH_SIZE $OUT/warp_pmc_WRITE_SIZE \
  $OUT/occ_pmc_FETCH_SIZE $OUT/occ_pmc_WRITE_SIZE > $ROOT/gpurun_out/prof_wide_summary.md
# keep only the small CSVs for merging back
find $OUT -name "*.db" -delete 2>/dev/null || true
du -sh $OUT
