#!/bin/bash
# End-to-end per-view time of the C++ driver (`apd`) at full resolution: a 6208x4128, 10-view, 9-sources-per-view folder
# of JPEG images through the coarse-to-fine schedule down to scale 1, with the wall time of every host step of every view
# (DVP_HOST_TIMING) next to the GPU time of RunPatchMatch.  GPU box, repo root.
# usage: tools/e2e_timing.sh [OUT=gpurun_out] [W=6208] [H=4128] [VIEWS=10] [SRC=9]
set -e
OUT=${1:-gpurun_out}; W=${2:-6208}; H=${3:-4128}; NV=${4:-10}; NS=${5:-9}
mkdir -p "$OUT"
DS=/tmp/ds_e2e
rm -rf $DS
python tools/make_dataset.py $DS $W $H $NV $NS --jpg --torch > /dev/null
( time DVP_HOST_TIMING=1 ./dvp-mvs_amd/apd $DS 0 --iters 3 --passes 1 --min-scale 1 --seed 3 --no-fusion ) > "$OUT/e2e_apd.log" 2>&1
python tools/e2e_summary.py "$OUT/e2e_apd.log" $W $H > "$OUT/e2e_apd.txt"
cat "$OUT/e2e_apd.txt"
