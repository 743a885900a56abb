#!/bin/bash
# first GPU call of round 3: environment probe, the new analytic / device-build / CLI tests, then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python tools/env_probe.py > gpurun_out/env_probe.txt 2>&1; echo "env_probe rc=$?"
cat gpurun_out/env_probe.txt
timeout -k 10 900 python -u -m pytest tests/test_gpu_analytic.py tests/test_gpu_device_build.py tests/test_cli_gpu.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_new_tests.log 2>&1
rc=$?; tail -25 gpurun_out/r3_new_tests.log; echo "new tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_gpu_tests.log 2>&1
rc=$?; tail -8 gpurun_out/r3_gpu_tests.log; echo "gpu suite rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 5 > gpurun_out/r3_bench0.json 2> gpurun_out/r3_bench0.err
rc=$?; cat gpurun_out/r3_bench0.json; echo "bench rc=$rc"
