#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --tp-dry 8 --model llama3-70b --tp-split-vocab --steps 16 --warmup 4 > gpurun_out/r06_tp_dry_run_70b_tp8_vocab_split.json 2> gpurun_out/r06_tp_dry.err; echo "rc=$?"
timeout 1500 python -m pytest -q -p no:cacheprovider tests/test_hip_tp.py tests/test_hip_tp_p2p.py tests/test_hip_c5_shape.py tests/test_hip_tp_bench.py -m gpu -x -q 2>&1 | tail -3
