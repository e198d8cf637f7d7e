set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep or fused or cfg3" > gpurun_out/r06/sweep_tests.log 2>&1
tail -3 gpurun_out/r06/sweep_tests.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/bench_layout.json 2> gpurun_out/r06/bench_layout.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_layout.json')); print(d['value'], d['stage_ms_per_step'])"
