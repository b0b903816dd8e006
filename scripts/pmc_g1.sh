#!/bin/bash
# rocprofv3 counters of g1_step_kernel (separate --pmc passes, kernel trace only).  bash scripts/pmc_g1.sh <tag> [skipbits]
set -e
TAG=${1:-g1}
export DMG1_SKIP=${2:-0}
OUT=$(pwd)/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_FLAT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/${TAG}_pmc$i -o run -- python3 scripts/bench_g1.py 4096 12 > $OUT/${TAG}_pmc$i.log 2>&1 || echo "pass $i failed"
done
python3 scripts/summarize_pmc.py $OUT/${TAG}_pmc_g1_step_kernel.csv g1_step_kernel $OUT/${TAG}_pmc1 $OUT/${TAG}_pmc2 $OUT/${TAG}_pmc3 $OUT/${TAG}_pmc4 $OUT/${TAG}_pmc5 $OUT/${TAG}_pmc6 $OUT/${TAG}_pmc7
cat $OUT/${TAG}_pmc_g1_step_kernel.csv
