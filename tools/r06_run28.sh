mkdir -p gpurun_out/r06
timeout 1200 bash tools/e2e_timing.sh gpurun_out/r06 > gpurun_out/r06/e2e_console.log 2>&1
grep "^pass\|real" gpurun_out/r06/e2e_apd.txt
grep -n "Cost time" gpurun_out/r06/e2e_apd.log | awk -F'Cost time: ' '{print $2}' | awk '{printf "%s ", $1} END{print ""}'
