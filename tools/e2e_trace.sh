# kernel trace of the `apd` driver on a small folder (GPU box, repo root): per (kernel, grid) times of every level of the schedule
#   bash tools/e2e_trace.sh <tag> [W H VIEWS SRC]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=$1; W=${2:-3104}; H=${3:-2064}; NV=${4:-4}; NS=${5:-3}
DS=/tmp/ds_trace
rm -rf $DS
python tools/make_dataset.py $DS $W $H $NV $NS --jpg --torch > /dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/trace_$T -o $T -- ./dvp-mvs_amd/apd $DS 0 --iters 3 --passes 1 --min-scale 1 --seed 3 --no-fusion > gpurun_out/${T}_apd.log 2>&1
DB=$(find gpurun_out/trace_$T -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" > gpurun_out/${T}_kernel_stats.txt
rm -rf gpurun_out/trace_$T
head -40 gpurun_out/${T}_kernel_stats.txt
