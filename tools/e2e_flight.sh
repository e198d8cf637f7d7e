# views in flight, A/B over the whole `apd` schedule on the e2e folder (GPU box, repo root): per-pass time per view
#   bash tools/e2e_flight.sh [extra apd arguments, e.g. --jacobi]
cd $GRAFT_REPO_ROOT
W=${W:-6208}; H=${H:-4128}; NV=${NV:-10}; NS=${NS:-9}
DS=/tmp/ds_fl
rm -rf $DS
python tools/make_dataset.py $DS $W $H $NV $NS --jpg --torch > /dev/null
run() { tag=$1; shift
  rm -rf $DS/APD
  ( time DVP_HOST_TIMING=1 ./dvp-mvs_amd/apd $DS 0 --iters 3 --passes 1 --min-scale 1 --seed 3 --no-fusion "$@" ) > gpurun_out/fl_$tag.log 2>&1
  python tools/e2e_summary.py gpurun_out/fl_$tag.log $W $H | grep -E "^pass|real" | sed "s/^/[$tag] /" | cut -c1-150
  ( cd $DS/APD && find . -name "*.dmb" -o -name "*.bin" | sort | xargs md5sum | md5sum )
}
run warmup --views-in-flight 1 "$@" > /dev/null
run one --views-in-flight 1 "$@"
run default "$@"
run three --views-in-flight 3 "$@"
run two_to_7mpx --in-flight-pixels 7000000 "$@"
run one_again --views-in-flight 1 "$@"
