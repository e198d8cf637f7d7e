# GenNeighbours under parameter ablations (GPU box, repo root): which part of the search the launch waits for
cd $GRAFT_REPO_ROOT
for p in "" "--param use_limit=0" "--param use_label=0" "--param use_limit=0 --param use_label=0"; do
  BENCH_ARGS="$p" bash tools/ab_variant_trace.sh gn_abl "dvp_gen_neighbours|dvp_ransac" 2>&1 | grep -v "^==" | sed "s/^/[$p] /"
done
