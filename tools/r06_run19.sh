set -x
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py tests/test_boundary.py -x -q -m gpu > gpurun_out/r06/reserve_tests.log 2>&1
tail -3 gpurun_out/r06/reserve_tests.log
for i in 1 2; do
E2E_CHECK_FUSION= timeout 1500 bash tools/e2e_timing.sh gpurun_out/r06 > gpurun_out/r06/e2e_console.log 2>&1
grep -n "^pass\|real" gpurun_out/r06/e2e_apd.txt
grep "GPU RunPatchMatch" gpurun_out/r06/e2e_apd.log | sed 's/.*RunPatchMatch \([0-9.]*\) ms.*/\1/' | tr '\n' ' ' | fold -w 200
cp gpurun_out/r06/e2e_apd.txt gpurun_out/r06/e2e_apd_run$i.txt
done
