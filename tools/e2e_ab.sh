# A/B of the whole `apd` schedule on the e2e folder (GPU box, repo root): per-pass kernel ms for the tree's library, for environment
# switches and for variant libraries (copied over dvp-mvs_amd/libdvp_mvs_hip.so for the run: apd links it by rpath)
cd $GRAFT_REPO_ROOT
W=${W:-6208}; H=${H:-4128}; NV=${NV:-6}; NS=${NS:-5}
DS=/tmp/ds_ab
rm -rf $DS
python tools/make_dataset.py $DS $W $H $NV $NS --jpg --torch > /dev/null
cp dvp-mvs_amd/libdvp_mvs_hip.so /tmp/tree.so
run() { tag=$1; shift
  rm -rf $DS/APD
  ( time env "$@" DVP_HOST_TIMING=1 ./dvp-mvs_amd/apd $DS 0 --iters 3 --passes 1 --min-scale 1 --seed 3 --no-fusion ) > gpurun_out/ab_$tag.log 2>&1
  python tools/e2e_summary.py gpurun_out/ab_$tag.log $W $H | grep -E "^pass|real" | sed "s/^/[$tag] /" | cut -c1-110
}
run warmup X=1 > /dev/null
run tree X=1
run onewave DVP_WEAK_PHASED=0
for v in "$@"; do cp build/variants/$v.so dvp-mvs_amd/libdvp_mvs_hip.so; run $v X=1; cp /tmp/tree.so dvp-mvs_amd/libdvp_mvs_hip.so; done
for t in tree onewave "$@"; do echo "== $t: [gpu] lines of view 0, passes 2 4 6 7"; grep -n "\[gpu\]" gpurun_out/ab_$t.log | awk -v n=$NV 'NR==2*n+1 || NR==4*n+1 || NR==6*n+1 || NR==7*n+1' | cut -c1-400; done
