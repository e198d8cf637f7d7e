# The round's measurement pass on the GPU box (repo root): PMC table for the three bench configurations, kernel trace
# of the default bench, the secondary bench lines, the gather calibration.  Results land in gpurun_out/ (copy to profiles/).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=z
rm -f profiles/pmc_r02.json
for cfg in cfg3 cfg2 cfg5; do
  bash tools/pmc_collect.sh gpurun_out/pmc_$cfg --config $cfg
  python tools/pmc_table.py gpurun_out/pmc_$cfg profiles/pmc_r02.json
  rm -rf gpurun_out/pmc_$cfg   # raw per-dispatch counter CSVs: tens of MB
done
cp profiles/pmc_r02.json gpurun_out/pmc_r02.json
bash tools/profile_bench.sh gpurun_out r02$T
rm -rf gpurun_out/trace_r02$T
python bench.py --config cfg2 --steps 5 --warmup 1 > gpurun_out/r02${T}_cfg2_bench.json 2> gpurun_out/r02${T}_cfg2.err
python bench.py --config cfg5 --steps 5 --warmup 1 > gpurun_out/r02${T}_cfg5_bench.json 2> gpurun_out/r02${T}_cfg5.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r02${T}_steps20_bench.json 2> gpurun_out/r02${T}_steps20.err
./build/gather_peak > gpurun_out/r02_gather_peak.txt 2>&1
tail -c 600 gpurun_out/r02${T}_steps20_bench.json
