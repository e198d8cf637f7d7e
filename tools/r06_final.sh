# round 6 final measurement pass: whole GPU suite, PMC tables, trace, bench lines, regimes, e2e
set -x
mkdir -p gpurun_out
# the loaded library must be the tree's (a comment edited after the build makes bench.py refuse the PMC table)
python - <<'PY' || exit 1
import ctypes, subprocess, sys
l = ctypes.CDLL('dvp-mvs_amd/libdvp_mvs_hip.so'); l.dvp_build_id.restype = ctypes.c_char_p
tree = subprocess.check_output([sys.executable, 'tools/csrc_hash.py'], text=True).strip()
lib = l.dvp_build_id().decode().split()[0]
print('library', lib[:12], 'tree', tree[:12])
sys.exit(0 if lib == tree else 1)
PY
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r06_gpu_suite.log 2>&1; tail -3 gpurun_out/r06_gpu_suite.log
bash tools/final_measure.sh r06 > gpurun_out/final_measure_r06.log 2>&1
bash tools/weak_regimes.sh r06 > gpurun_out/r06_weak_regimes.txt 2>&1
tail -5 gpurun_out/r06_weak_regimes.txt
tail -c 600 gpurun_out/r06_steps20_bench.json
