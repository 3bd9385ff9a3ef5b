#!/bin/bash
# rocprofv3 --pmc MfmaUtil VALUBusy (own pass, no tracing domains) of the prompt pass: the weight GEMMs (f16 weight-stationary from 32 rows, int8 below) and the fast step's flash attention.
# usage: gpurun --timeout 900 -- 'bash tools/gpu_profile_prefill.sh TAG'
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp; cd $REPO  # (the tools put "." on sys.path)
rm -rf $OUT/pmc_prefill_$TAG $OUT/pmc_prefill_long_$TAG
timeout 400 rocprofv3 --kernel-trace --pmc MfmaUtil VALUBusy -d $OUT/pmc_prefill_$TAG -- python $REPO/tools/prefill_bench.py --chunks 512 --loop 2 --layers 4 > $OUT/pmc_prefill_$TAG.out 2> $OUT/pmc_prefill_$TAG.err
timeout 400 rocprofv3 --kernel-trace --pmc MfmaUtil VALUBusy -d $OUT/pmc_prefill_long_$TAG -- python $REPO/tools/long_prefill_bench.py --n 4096 --layers 2 --only-default > $OUT/pmc_prefill_long_$TAG.out 2>> $OUT/pmc_prefill_$TAG.err
{
  echo "# $TAG PMC MfmaUtil / VALUBusy of the prompt pass (rocprofv3 --kernel-trace --pmc MfmaUtil VALUBusy, own pass): mean per dispatch"
  echo "# (a) tools/prefill_bench.py --chunks 512 --loop 2 --layers 4   (Llama-3-8B shape Q4_0, 512 rows per pass)"
  python $REPO/tools/rocpd_pmc.py $(find $OUT/pmc_prefill_$TAG -name "*_results.db" | head -1) | grep -E "kernel \||---|gemm_mfma|gemm_f16w|flash_rows|attn_tile"
  echo
  echo "# (b) tools/long_prefill_bench.py --n 4096 --layers 2 --only-default   (4096-token prompt: the attention kernel at full length)"
  python $REPO/tools/rocpd_pmc.py $(find $OUT/pmc_prefill_long_$TAG -name "*_results.db" | head -1) | grep -E "kernel \||---|gemm_mfma|gemm_f16w|flash_rows|attn_tile"
} > $OUT/pmc_mfma_util_prefill_$TAG.md 2>> $OUT/pmc_prefill_$TAG.err
cat $OUT/pmc_mfma_util_prefill_$TAG.md
find $OUT/pmc_prefill_$TAG $OUT/pmc_prefill_long_$TAG -name "*.db" -size +20M -delete
