# round 6, GPU call 5: ten-view full-size schedule WITH fusion (pool eviction as in round 5, fusion rounds until the tail), device .ply vs host .ply
set -x
mkdir -p gpurun_out/r06
E2E_CHECK_FUSION=1 timeout 1500 bash tools/e2e_timing.sh gpurun_out/r06 > gpurun_out/r06/e2e_console.log 2>&1
grep -n "^pass\|real\|fusion" gpurun_out/r06/e2e_apd.txt
timeout 600 python -m pytest tests/test_host_oracles.py tests/test_boundary.py -x -q -m gpu -k "fusion" > gpurun_out/r06/fusion_tests.log 2>&1; tail -3 gpurun_out/r06/fusion_tests.log
