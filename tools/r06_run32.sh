mkdir -p gpurun_out/r06
for v in "$@"; do
  DVP_MVS_LIB=$PWD/build/variants/$v.so timeout 600 bash tools/profile_bench.sh gpurun_out/r06 $v --no-cpu-baseline > /dev/null
  rm -rf gpurun_out/r06/trace_$v
  python -c "
import json; d=json.load(open('gpurun_out/r06/${v}_bench.json')); print('$v', d['value'], d['ms_per_step'], {k: round(x,1) for k,x in d['stage_ms_per_step'].items() if x > 5})"
  grep "gen_candidates" gpurun_out/r06/${v}_kernel_stats.txt | head -2
done
