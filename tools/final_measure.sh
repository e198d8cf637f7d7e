# The round's measurement pass on the GPU box (repo root): PMC table for the three bench configurations (keyed by the csrc
# hash), kernel trace of the default bench, the secondary bench lines, end-to-end apd timing.  Results land in gpurun_out/
# (copy to profiles/).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r06}
rm -f profiles/pmc_r06.json
for cfg in cfg3 cfg2 cfg5; do
  timeout 1500 bash tools/pmc_collect.sh gpurun_out/pmc_$cfg --config $cfg
  python tools/pmc_table.py gpurun_out/pmc_$cfg profiles/pmc_r06.json
  rm -rf gpurun_out/pmc_$cfg   # raw per-dispatch counter CSVs: tens of MB
done
cp profiles/pmc_r06.json gpurun_out/pmc_r06.json
timeout 900 bash tools/profile_bench.sh gpurun_out $T
rm -rf gpurun_out/trace_$T
timeout 600 python bench.py --config cfg2 --steps 5 --warmup 1 > gpurun_out/${T}_cfg2_bench.json 2> gpurun_out/${T}_cfg2.err
timeout 600 python bench.py --config cfg5 --steps 5 --warmup 1 > gpurun_out/${T}_cfg5_bench.json 2> gpurun_out/${T}_cfg5.err
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_steps20_bench.json 2> gpurun_out/${T}_steps20.err
timeout 600 python bench.py --steps 5 --warmup 1 --rig axis --no-cpu-baseline > gpurun_out/${T}_axis_rig_bench.json 2> gpurun_out/${T}_axis.err
DVP_NO_IMAGES8=1 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${T}_float_planes_bench.json 2> gpurun_out/${T}_float.err
E2E_CHECK_FUSION=1 timeout 900 bash tools/e2e_timing.sh gpurun_out > gpurun_out/e2e.out 2>&1
tail -c 900 gpurun_out/${T}_steps20_bench.json
