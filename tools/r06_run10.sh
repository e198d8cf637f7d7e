set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -x -q -m gpu -k "sweep or fused or cfg3_shaped" > gpurun_out/r06/sweep_tests.log 2>&1
tail -3 gpurun_out/r06/sweep_tests.log
for v in tiled untiled; do
lib=$PWD/dvp-mvs_amd/libdvp_mvs_hip.so; [ $v = untiled ] && lib=$PWD/build/probe/untiled.so
DVP_MVS_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_$v.json 2> gpurun_out/r06/ab_$v.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_$v.json')); print('$v', d['value'], d['stage_ms_per_step']['depth_to_weak'])"
done
rocprofv3 --kernel-trace --stats -d gpurun_out/r06/prof_sweep -o sweep -- python bench.py --steps 1 --warmup 0 --no-secondary --no-cpu-baseline --no-per-iteration > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/r06/prof_sweep/sweep_results.db | grep -i "sweep\|calls"
