set -x
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_driver.py tests/test_boundary.py -x -q -m gpu > gpurun_out/r06/driver_tests.log 2>&1
tail -3 gpurun_out/r06/driver_tests.log
E2E_CHECK_FUSION=1 timeout 1500 bash tools/e2e_timing.sh gpurun_out/r06 > gpurun_out/r06/e2e_console.log 2>&1
grep -n "^pass\|real\|fusion" gpurun_out/r06/e2e_apd.txt
grep "\[main\]" gpurun_out/r06/e2e_apd.log | tail -4
