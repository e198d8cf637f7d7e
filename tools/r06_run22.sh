set -x
mkdir -p gpurun_out/r06
timeout 900 bash tools/profile_bench.sh gpurun_out/r06 layout --no-cpu-baseline
rm -rf gpurun_out/r06/trace_layout
grep "sweep\|strong_eval\|decide" gpurun_out/r06/layout_kernel_stats.txt | head -12
