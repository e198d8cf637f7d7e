# per-kernel times of variant libraries (GPU box, repo root): tools/ab_trace.sh PATTERN v1 v2 ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PAT=$1; shift
for v in "$@"; do
  LIBV=$PWD/build/variants/$v.so; [ "$v" = tree ] && LIBV=""; DVP_MVS_LIB=$LIBV timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/trace_ab -o ab -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== $v"; python tools/rocpd_summary.py $(find gpurun_out/trace_ab -name "*.db" | head -1) | grep -E "$PAT"; rm -rf gpurun_out/trace_ab
done
