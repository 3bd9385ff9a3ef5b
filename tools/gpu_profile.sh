#!/bin/bash
# rocprofv3 passes of the default bench line (run on the GPU box through gpurun): kernel trace, then FETCH_SIZE in its OWN pass.
# usage: gpurun --timeout 900 -- 'bash tools/gpu_profile.sh TAG [extra bench args]'
TAG=${1:-r02}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--steps 24 --warmup 8 --repeats 1 --no-cpu-baseline --no-parity-check --no-context --no-prefill --no-c3 --no-gemv-points --no-trait $*"
rm -rf $OUT/prof_$TAG $OUT/pmc_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -- python $REPO/bench.py $B > $OUT/bench_under_rocprof_$TAG.json 2> $OUT/prof_$TAG.err
DB=$(find $OUT/prof_$TAG -name "*_results.db" | head -1)
echo "# $TAG fused decode step, kernel trace: rocprofv3 --kernel-trace --stats -- python bench.py $B" > $OUT/kernel_trace_$TAG.md
echo "# (under the tracer the graph replay is serialized per kernel node: the per-kernel durations are what this file is for)" >> $OUT/kernel_trace_$TAG.md
echo >> $OUT/kernel_trace_$TAG.md
python $REPO/tools/rocpd_summary.py "$DB" >> $OUT/kernel_trace_$TAG.md 2>> $OUT/prof_$TAG.err
head -22 $OUT/kernel_trace_$TAG.md
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_$TAG -- python $REPO/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-parity-check --no-context --no-prefill --no-c3 --no-gemv-points --no-trait $* > /dev/null 2>> $OUT/prof_$TAG.err
DB2=$(find $OUT/pmc_$TAG -name "*_results.db" | head -1)
cd $REPO && python tools/pmc_traffic.py "$DB2" $OUT/${TAG}_pmc_fetch_size.md --trace="$DB" --trace-md=${TAG}_fused_path_kernel_trace.md && cp profiles/pmc_traffic.json $OUT/pmc_traffic_$TAG.json
# the reference's API: Llama2Runner<HipTensor> unchanged, one Tensor call after the other -- which kernels do its calls become?
cd /tmp
rm -rf $OUT/prof_trait_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_trait_$TAG -- python $REPO/tools/trait_trace.py 24 > $OUT/trait_trace_$TAG.json 2>> $OUT/prof_$TAG.err
DB3=$(find $OUT/prof_trait_$TAG -name "*_results.db" | head -1)
echo "# $TAG the reference's unchanged runner (Llama2Runner<HipTensor>::forward x 24 tokens, 8B shape Q4_0) under rocprofv3 --kernel-trace --stats:" > $OUT/${TAG}_trait_path_kernel_trace.md
echo "# its ~840 Tensor calls per token are recorded and served by the fused step -- 5 launches per layer (k_qkv, k_attn_s, k_gemv_res_nq x 2, k_gateup_q)" >> $OUT/${TAG}_trait_path_kernel_trace.md
echo "# $(cat $OUT/trait_trace_$TAG.json)" >> $OUT/${TAG}_trait_path_kernel_trace.md
echo >> $OUT/${TAG}_trait_path_kernel_trace.md
python $REPO/tools/rocpd_summary.py "$DB3" >> $OUT/${TAG}_trait_path_kernel_trace.md 2>> $OUT/prof_$TAG.err
head -16 $OUT/${TAG}_trait_path_kernel_trace.md
cd $REPO
# keep the merge small
find $OUT/prof_$TAG $OUT/pmc_$TAG $OUT/prof_trait_$TAG -name "*.db" -size +20M -delete
