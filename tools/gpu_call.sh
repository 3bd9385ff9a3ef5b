#!/bin/bash
# One GPU-box visit (gpurun).  Logs under gpurun_out/.
# usage: gpurun --timeout 1800 -- 'bash tools/gpu_call.sh TAG [newtests] [tests] [alltests] [bench] [fault] [tptests] [smoke]'
TAG=${1:-r3}; shift
WHAT="${*:-alltests bench}"
mkdir -p gpurun_out
export TMPDIR=/tmp
PT="python -m pytest -q -p no:cacheprovider"
for w in $WHAT; do
  case $w in
    newtests) timeout 1500 $PT tests/test_hip_long_context_oracle.py tests/test_hip_c5_shape.py "tests/test_hip_tp_p2p.py::test_q4_1_body_with_a_q6_k_classifier_over_the_p2p_group" > gpurun_out/new_tests_$TAG.log 2>&1; echo "newtests rc=$?"; tail -25 gpurun_out/new_tests_$TAG.log ;;
    tests) timeout 1500 $PT tests -m gpu --maxfail=25 --deselect tests/test_hip_long_context_oracle.py --deselect tests/test_hip_c5_shape.py > gpurun_out/gpu_tests_$TAG.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/gpu_tests_$TAG.log ;;
    alltests) timeout 2400 $PT tests -m gpu --maxfail=25 > gpurun_out/gpu_tests_all_$TAG.log 2>&1; echo "alltests rc=$?"; tail -8 gpurun_out/gpu_tests_all_$TAG.log ;;
    bench) timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err ;;
    fault) timeout 900 $PT tests/test_hip_fault_paths.py > gpurun_out/fault_tests_$TAG.log 2>&1; echo "fault rc=$?"; tail -15 gpurun_out/fault_tests_$TAG.log ;;
    tptests) timeout 900 $PT tests/test_hip_tp.py tests/test_hip_tp_p2p.py > gpurun_out/tp_tests_$TAG.log 2>&1; echo "tptests rc=$?"; tail -15 gpurun_out/tp_tests_$TAG.log ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke_$TAG.log ;;
  esac
done
