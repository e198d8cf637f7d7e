set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_host_oracles.py tests/test_boundary.py -x -q -m gpu -k "fusion" > gpurun_out/r06/fusion_tests.log 2>&1
tail -5 gpurun_out/r06/fusion_tests.log
for lb in 2 3; do
DVP_MVS_LIB=$PWD/build/probe/lbd$lb.so timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_lbd$lb.json 2> gpurun_out/r06/ab_lbd$lb.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_lbd$lb.json')); print('lb_decide $lb', d['value'], d['stage_ms_per_step']['strong_update'])"
done
