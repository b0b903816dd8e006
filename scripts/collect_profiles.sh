#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline object cites.  Run on the GPU box from the repo root:
#   bash scripts/collect_profiles.sh r01_v7
# Writes gpurun_out/<tag>_* ; copy the two summaries into profiles/ afterwards.
set -e
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# pass 1: kernel trace + stats of the default bench command (no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o run -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-pipelined --no-ppo-loop --no-g1 --no-physics-only > $OUT/${TAG}_trace.log 2>&1
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
# passes 2..: counters, each group in its own run, kernel trace only
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
         "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/${TAG}_pmc$i -o run -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-pipelined --no-ppo-loop --no-g1 --no-physics-only > $OUT/${TAG}_pmc$i.log 2>&1
done
python3 scripts/summarize_pmc.py $OUT/${TAG}_pmc_dm_step_kernel.csv dm_step_kernel $OUT/${TAG}_pmc1 $OUT/${TAG}_pmc2 $OUT/${TAG}_pmc3 $OUT/${TAG}_pmc4 $OUT/${TAG}_pmc5
head -5 $OUT/${TAG}_kernel_stats.csv
