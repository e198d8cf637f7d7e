# round 6 final measurement pass: whole GPU suite, PMC tables, trace, bench lines, regimes, e2e
set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06_gpu_suite.log 2>&1; tail -3 gpurun_out/r06_gpu_suite.log
bash tools/final_measure.sh r06 > gpurun_out/final_measure_r06.log 2>&1
bash tools/weak_regimes.sh r06 > gpurun_out/r06_weak_regimes.txt 2>&1
tail -5 gpurun_out/r06_weak_regimes.txt
tail -c 600 gpurun_out/r06_steps20_bench.json
