set -x
mkdir -p gpurun_out/r06
free -g | head -2 > gpurun_out/r06/mem.txt; nproc >> gpurun_out/r06/mem.txt
for lb in 2 3 4; do
  DVP_MVS_LIB=$PWD/build/probe/probe_lb$lb.so timeout 900 python tools/sweep_probe.py > gpurun_out/r06/sweep_probe_lb$lb.txt 2>&1
done
timeout 1500 python -m pytest tests/test_fullsize_sampled_parity.py -x -q -m gpu -s > gpurun_out/r06/fullsize_parity.log 2>&1
tail -5 gpurun_out/r06/fullsize_parity.log
