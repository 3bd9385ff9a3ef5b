#!/bin/bash
# One GPU-box visit (gpurun): GPU test suite, the DMA lab, the default bench line, A/B bench lines.  Logs under gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_call.sh TAG [tests] [lab] [bench] [ab]'
TAG=${1:-r2}; shift
WHAT="${*:-tests lab bench ab}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in $WHAT; do
  case $w in
    tests) timeout 1100 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/gpu_tests_$TAG.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/gpu_tests_$TAG.log ;;
    lab) timeout 300 ./build/dma_lab > gpurun_out/dma_lab_$TAG.log 2>&1; echo "lab rc=$?"; cat gpurun_out/dma_lab_$TAG.log ;;
    bench) timeout 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err ;;
    ab) for f in 16 32; do timeout 300 python bench.py --flags $f --steps 48 --repeats 3 --no-cpu-baseline --no-parity-check --no-context --no-prefill > gpurun_out/bench_${TAG}_flags$f.json 2>> gpurun_out/bench_$TAG.err; python - <<PY
import json
d=json.load(open("gpurun_out/bench_${TAG}_flags$f.json"))
print("flags $f:", d["value"], {k:v["avg_us"] for k,v in d["roofline"]["per_stage"].items()})
PY
    done ;;
  esac
done
