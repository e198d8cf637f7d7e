# per-kernel times of the tree's library and of variant builds (build/variants/<name>.so), GPU box, repo root:
#   bash tools/ab_variant_trace.sh <tag> <kernel regex> [variant ...]
cd $GRAFT_REPO_ROOT
T=$1; RE=$2; shift 2
one() { n=$1; lib=$2
  DVP_MVS_LIB=$lib bash tools/profile_bench.sh gpurun_out ${T}_$n --no-cpu-baseline --no-secondary $BENCH_ARGS > /dev/null 2>&1; rm -rf gpurun_out/trace_${T}_$n
  echo "== $n"; python -c "
import json; d=json.load(open('gpurun_out/${T}_${n}_bench.json')); print(d['value'], d['ms_per_step'], {k: round(v,1) for k,v in d['stage_ms_per_step'].items() if v > 5})"
  grep -E "$RE" gpurun_out/${T}_${n}_kernel_stats.txt | awk '{a[$1]+=$3; c[$1]+=$2} END {for (k in a) printf "%s %.3f ms/launch (%d launches)\n", k, a[k]/c[k], c[k]}' | sort
}
one tree ""
for v in "$@"; do one $v $PWD/build/variants/$v.so; done
