#!/bin/bash
# rocprofv3 counters of the split G1 pipeline (g1_env_kernel, g1_pair_kernel): separate --pmc passes, kernel trace only.
#   bash scripts/pmc_g1_split.sh <tag>       (from the repo root, on the GPU box)
TAG=${1:-g1split}
OUT=$(pwd)/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
i=0
DIRS=""
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/${TAG}_pmc$i -o run -- python3 scripts/bench_g1.py 4096 12 walk 2 > $OUT/${TAG}_pmc$i.log 2>&1 || echo "pass $i ($G) failed"
  DIRS="$DIRS $OUT/${TAG}_pmc$i"
done
python3 scripts/summarize_pmc.py $OUT/${TAG}_pmc_g1_env_kernel.csv g1_env_kernel $DIRS > /dev/null
python3 scripts/summarize_pmc.py $OUT/${TAG}_pmc_g1_pair_kernel.csv g1_pair_kernel $DIRS > /dev/null
echo "== g1_env_kernel (mean per dispatch over the last 12 = two steps; x 6 per step)"; cat $OUT/${TAG}_pmc_g1_env_kernel.csv
echo "== g1_pair_kernel (mean per dispatch over the last 10 = two steps; x 5 per step)"; cat $OUT/${TAG}_pmc_g1_pair_kernel.csv
