set -e
python tools/make_dataset.py /tmp/ds_e2e 3104 2064 6 5 > /dev/null
( time DVP_HOST_TIMING=1 ./dvp-mvs_amd/apd /tmp/ds_e2e 0 --iters 3 --passes 1 --seed 3 ) > gpurun_out/e2e_apd.log 2>&1
grep -E "Cost time|Round|host|Fusion|real" gpurun_out/e2e_apd.log | tail -42
