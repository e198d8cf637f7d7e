set -x
mkdir -p gpurun_out/r06
DVP_MVS_LIB=$PWD/build/probe/geom_inside.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -x -q -m gpu -k "sweep or fused or cfg3_shaped or cfg1" > gpurun_out/r06/gi_tests.log 2>&1
tail -3 gpurun_out/r06/gi_tests.log
for v in base inside; do
lib=$PWD/dvp-mvs_amd/libdvp_mvs_hip.so; [ $v = inside ] && lib=$PWD/build/probe/geom_inside.so
DVP_MVS_LIB=$lib timeout 600 python bench.py --steps 5 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_gi_$v.json 2> gpurun_out/r06/ab_gi_$v.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_gi_$v.json')); print('geom $v', d['value'], d['stage_ms_per_step']['depth_to_weak'])"
done
