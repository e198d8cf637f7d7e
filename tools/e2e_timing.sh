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
# round 6: WITH the fusion (on the GPU, dvp_fuse_*); E2E_NO_FUSION=1 gives the rounds 3-5 command
FUSE=""; [ -n "$E2E_NO_FUSION" ] && FUSE="--no-fusion"
( time DVP_HOST_TIMING=1 ./dvp-mvs_amd/apd $DS 0 --iters 3 --passes 1 --min-scale 1 --seed 3 $FUSE ) > "$OUT/e2e_apd.log" 2>&1
python tools/e2e_summary.py "$OUT/e2e_apd.log" $W $H > "$OUT/e2e_apd.txt"
grep -E "^Fusion:|\[fusion\]" "$OUT/e2e_apd.log" >> "$OUT/e2e_apd.txt" || true
cat "$OUT/e2e_apd.txt"
if [ -z "$E2E_NO_FUSION" ] && [ -n "$E2E_CHECK_FUSION" ]; then
	# the same folder fused on the host's cores: the .ply must be the same file
	cp $DS/APD/APD.ply /tmp/e2e_device.ply
	( time DVP_FUSION_ON=host ./tests/host/test_host --fuse $DS ) > "$OUT/e2e_fusion_host.log" 2>&1
	if cmp -s /tmp/e2e_device.ply $DS/APD/APD.ply; then echo "fusion on the device == fusion on the host: APD.ply identical ($(stat -c %s /tmp/e2e_device.ply) bytes)"; else echo "FUSION MISMATCH: device and host .ply differ"; fi | tee -a "$OUT/e2e_apd.txt"
	grep -E "\[fusion\]|^real" "$OUT/e2e_fusion_host.log" | sed 's/^/host fusion: /' | tee -a "$OUT/e2e_apd.txt"
fi
