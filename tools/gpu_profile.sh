#!/bin/bash
# rocprofv3 passes of the default bench line (run on the GPU box through gpurun): kernel trace, then FETCH_SIZE in its OWN pass.
# usage: gpurun --timeout 900 -- 'bash tools/gpu_profile.sh TAG [extra bench args]'
TAG=${1:-r02}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--steps 24 --warmup 8 --repeats 1 --no-cpu-baseline --no-parity-check --no-context --no-prefill --no-c3 --no-gemv-points $*"
rm -rf $OUT/prof_$TAG $OUT/pmc_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -- python $REPO/bench.py $B > $OUT/bench_under_rocprof_$TAG.json 2> $OUT/prof_$TAG.err
DB=$(find $OUT/prof_$TAG -name "*_results.db" | head -1)
echo "# $TAG fused decode step, kernel trace: rocprofv3 --kernel-trace --stats -- python bench.py $B" > $OUT/kernel_trace_$TAG.md
echo "# (under the tracer the graph replay is serialized per kernel node: the per-kernel durations are what this file is for)" >> $OUT/kernel_trace_$TAG.md
echo >> $OUT/kernel_trace_$TAG.md
python $REPO/tools/rocpd_summary.py "$DB" >> $OUT/kernel_trace_$TAG.md 2>> $OUT/prof_$TAG.err
head -22 $OUT/kernel_trace_$TAG.md
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_$TAG -- python $REPO/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-parity-check --no-context --no-prefill --no-c3 --no-gemv-points $* > /dev/null 2>> $OUT/prof_$TAG.err
DB2=$(find $OUT/pmc_$TAG -name "*_results.db" | head -1)
cd $REPO && python tools/pmc_traffic.py "$DB2" $OUT/${TAG}_pmc_fetch_size.md && cp profiles/pmc_traffic.json $OUT/pmc_traffic_$TAG.json
# keep the merge small
find $OUT/prof_$TAG $OUT/pmc_$TAG -name "*.db" -size +20M -delete
