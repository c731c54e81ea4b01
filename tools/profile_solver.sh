#!/bin/bash
# Profiles the solver configurations on the GPU box: kernel-trace stats, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes
# (never combined with other trace domains).  Raw output under gpurun_out/prof_solver/, summary next to it.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_solver
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in c2 c3 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${w}_stats -o ${w} -- python3 $ROOT/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${w}_stats.log 2>&1
  echo "stats $w done"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/${w}_pmc_$c -o ${w} -- python3 $ROOT/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline > $OUT/${w}_pmc_$c.log 2>&1
    echo "pmc $w $c done"
  done
done
# issue and memory-wait counters, and the L2's hit / miss counts, for every configuration (8 SQ slots / 4 TCC slots per pass)
for w in c2 c3 c5; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${w}_pmc_SQ -o ${w} -- python3 $ROOT/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline > $OUT/${w}_pmc_SQ.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $OUT/${w}_pmc_SQ2 -o ${w} -- python3 $ROOT/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline > $OUT/${w}_pmc_SQ2.log 2>&1 || echo "SQ2 pass failed for $w (a counter name?)"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/${w}_pmc_TCC -o ${w} -- python3 $ROOT/bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline > $OUT/${w}_pmc_TCC.log 2>&1 || echo "TCC pass failed for $w"
  echo "counters $w done"
done
cd $ROOT
python3 tools/prof_summary.py "${PROF_TITLE:-r03 — solver configurations 2, 3 (compact sampled obstacles), 5: bench.py --workload cN --no-cpu-baseline}" \
  $OUT/c2_stats $OUT/c3_stats $OUT/c5_stats $OUT/c2_pmc_FETCH_SIZE $OUT/c2_pmc_WRITE_SIZE $OUT/c3_pmc_FETCH_SIZE $OUT/c3_pmc_WRITE_SIZE \
  $OUT/c5_pmc_FETCH_SIZE $OUT/c5_pmc_WRITE_SIZE $OUT/c2_pmc_SQ $OUT/c3_pmc_SQ $OUT/c5_pmc_SQ $OUT/c2_pmc_SQ2 $OUT/c3_pmc_SQ2 $OUT/c5_pmc_SQ2 \
  $OUT/c2_pmc_TCC $OUT/c3_pmc_TCC $OUT/c5_pmc_TCC > $ROOT/gpurun_out/prof_solver_summary.md
# the record bench.py reads its recorded (not live) figures from; PROF_HEAD = `git rev-parse --short HEAD` of the profiled build
# (the box has no .git), PROF_ROUND = rNN.  Copy both files into profiles/ afterwards.
python3 tools/prof_summary.py --json $ROOT/gpurun_out/${PROF_ROUND:-r03}_pmc.json --head "${PROF_HEAD:-unknown}" --from-md $ROOT/gpurun_out/prof_solver_summary.md $(ls $ROOT/profiles/r*_wide_rows_summary.md | tail -1)
# keep only the small files that travel back (csv of stats and counters), not the raw traces
find $OUT -name "*.db" -delete 2>/dev/null || true
du -sh $OUT
