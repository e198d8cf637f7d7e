set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py tests/test_random_configs.py -x -q -m gpu -k "ransac or cfg3_shaped or weak or random or large" > gpurun_out/r06/ransac_tests.log 2>&1
tail -3 gpurun_out/r06/ransac_tests.log
for v in wave lane; do
e=1; [ $v = lane ] && e=0
DVP_RANSAC_WAVE=$e timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_ransac_$v.json 2> gpurun_out/r06/ab_ransac_$v.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_ransac_$v.json')); print('ransac $v', d['value'], d['stage_ms_per_step']['ransac_fit'])"
DVP_RANSAC_WAVE=$e timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration --weak-frac 0.25 > gpurun_out/r06/ab_ransac25_$v.json 2> gpurun_out/r06/ab_ransac25_$v.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_ransac25_$v.json')); print('ransac 25% $v', d['value'], d['stage_ms_per_step']['ransac_fit'])"
done
