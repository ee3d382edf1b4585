#!/bin/bash
# round-2 check: new parity tests at full size, then the default bench line
set -o pipefail
mkdir -p gpurun_out/r2b
timeout -k 10 500 python -u -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r2b/pytest.log 2>&1 || { tail -30 gpurun_out/r2b/pytest.log; exit 1; }
tail -3 gpurun_out/r2b/pytest.log
timeout -k 10 500 python bench.py --steps 2 --warmup 1 > gpurun_out/r2b/bench_default.json 2> gpurun_out/r2b/bench_default.err
echo rc=$?
tail -c 2500 gpurun_out/r2b/bench_default.json
tail -5 gpurun_out/r2b/bench_default.err
