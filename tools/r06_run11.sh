set -x
mkdir -p gpurun_out/r06
for v in 1 2 4 8; do
lib=$PWD/dvp-mvs_amd/libdvp_mvs_hip.so; [ $v != 1 ] && lib=$PWD/build/probe/xw$v.so
DVP_MVS_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_xw$v.json 2> gpurun_out/r06/ab_xw$v.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_xw$v.json')); print('xw $v', d['value'], d['stage_ms_per_step']['depth_to_weak'])"
done
