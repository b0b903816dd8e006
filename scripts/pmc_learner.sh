#!/bin/bash
# rocprofv3 counter passes over the learner micro-benchmark (scripts/bench_learner.py); run on the GPU box from the repo root:
#   bash scripts/pmc_learner.sh TAG     -> gpurun_out/TAG_pmc_<kernel>.csv for the fused learner kernels
TAG=${1:-lrn}
OUT=$(pwd)/gpurun_out
export TMPDIR=/tmp
i=0
for G in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  echo "pass $i: $G"
  timeout -k 10 150 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/${TAG}_pmc$i -o run -- python3 scripts/bench_learner.py 256,128 30 > $OUT/${TAG}_pmc$i.log 2>&1
done
for K in "(anonymous namespace)::mlp_fwdbwd_kernel" "(anonymous namespace)::mlp_wgrad_kernel" "(anonymous namespace)::mlp_pack_kernel"; do
  N=$(echo $K | sed 's/.*:://')
  python3 scripts/summarize_pmc.py $OUT/${TAG}_pmc_$N.csv "$K" $OUT/${TAG}_pmc1 $OUT/${TAG}_pmc2 $OUT/${TAG}_pmc3 $OUT/${TAG}_pmc4
  echo "== $N"; cat $OUT/${TAG}_pmc_$N.csv
done
