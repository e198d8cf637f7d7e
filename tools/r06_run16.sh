set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -x -q -m gpu -k "sweep or fused or cfg3_shaped or cfg3_full or cfg5" > gpurun_out/r06/persist_tests.log 2>&1
tail -3 gpurun_out/r06/persist_tests.log
for v in 1 0; do
DVP_SWEEP_PERSIST=$v timeout 600 python bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_persist$v.json 2> gpurun_out/r06/ab_persist$v.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_persist$v.json')); print('persist $v', d['value'], d['stage_ms_per_step']['depth_to_weak'])"
done
DVP_SWEEP_SLOTS=4096 timeout 600 python bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_persist4096.json 2> gpurun_out/r06/ab_persist4096.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_persist4096.json')); print('persist slots 4096', d['value'], d['stage_ms_per_step']['depth_to_weak'])"
