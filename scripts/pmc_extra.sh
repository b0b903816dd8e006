#!/bin/bash
# Extra latency-oriented counters (instruction fetch, LDS/VMEM/SMEM in-flight levels).  Usage: bash scripts/pmc_extra.sh TAG
set -e
TAG=${1:-x}
OUT=$(pwd)/gpurun_out
export TMPDIR=/tmp
i=0
for G in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" \
         "SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/${TAG}_x$i -o run -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-pipelined --no-ppo-loop --no-g1 > $OUT/${TAG}_x$i.log 2>&1
done
python3 scripts/summarize_pmc.py $OUT/${TAG}_pmc_extra.csv dm_step_kernel $OUT/${TAG}_x1 $OUT/${TAG}_x2 $OUT/${TAG}_x3 $OUT/${TAG}_x4 $OUT/${TAG}_x5
