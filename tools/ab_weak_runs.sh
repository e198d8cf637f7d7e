# per-kernel times of the phased weak update for several XCD run lengths (GPU box, repo root): bash tools/ab_weak_runs.sh <tag> "a,b,c,d" ...
cd $GRAFT_REPO_ROOT
T=$1; shift
for g in "$@"; do
  export DVP_WEAK_RUNS=$g
  n=$(echo $g | tr , _)
  bash tools/profile_bench.sh gpurun_out ${T}_$n --no-cpu-baseline $BENCH_ARGS > /dev/null 2>&1; rm -rf gpurun_out/trace_${T}_$n
  echo "== runs $g"; grep -E "dvp_weak_eval" gpurun_out/${T}_${n}_kernel_stats.txt | awk '{print $1, $2, $4}' | sort | awk '{a[$1]+=$3; c[$1]++} END {for (k in a) printf "%s %.3f\n", k, a[k]/c[k]}' | sort
done
