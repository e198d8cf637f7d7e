set -x
mkdir -p gpurun_out/r06
for r in 14 12 10; do
lib=$PWD/dvp-mvs_amd/libdvp_mvs_hip.so; [ $r != 14 ] && lib=$PWD/build/probe/rows$r.so
DVP_MVS_LIB=$lib timeout 600 python bench.py --steps 4 --warmup 1 --no-secondary --no-cpu-baseline --no-per-iteration > gpurun_out/r06/ab_rows$r.json 2> gpurun_out/r06/ab_rows$r.err
python -c "
import json; d=json.load(open('gpurun_out/r06/ab_rows$r.json')); print('rows $r', d['value'], d['stage_ms_per_step']['depth_to_weak'])"
done
