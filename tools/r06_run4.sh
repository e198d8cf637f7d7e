# round 6, GPU call 4: fusion on the device — tests, then the ten-view full-size schedule WITH fusion, device .ply vs host .ply
set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_host_oracles.py -x -q -m gpu -k fusion -s > gpurun_out/r06/fusion_tests.log 2>&1
tail -8 gpurun_out/r06/fusion_tests.log
E2E_CHECK_FUSION=1 timeout 1500 bash tools/e2e_timing.sh gpurun_out/r06 > gpurun_out/r06/e2e_console.log 2>&1
tail -30 gpurun_out/r06/e2e_apd.txt
