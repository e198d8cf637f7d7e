#!/bin/bash
# Start one `apd` process per GPU of this node (one rank per MI355X, RCCL over xGMI between them).
#   tools/run_node.sh <dense_folder> [N ranks, default: every GPU rocm-smi / HIP shows] [apd options...]
# Rank r runs on HIP_VISIBLE_DEVICES=r with views v % N == r; the ranks find each other through the
# job id (a nonce-tagged rendezvous file in the dense folder).  The script waits for all ranks and returns the first
# non-zero exit status; a failing rank takes the others down through the abort marker (host/comm.cpp).
# The reference has no counterpart: /root/reference/main.cpp:430-434 selects one device for the whole program.
set -u
here="$(cd "$(dirname "$0")/.." && pwd)"
folder="${1:?usage: run_node.sh <dense_folder> [N] [apd options]}"; shift
if [ $# -gt 0 ] && [[ "$1" =~ ^[0-9]+$ ]]; then n="$1"; shift; else
	n=$(ls -d /sys/class/kfd/kfd/topology/nodes/*/ 2>/dev/null | while read d; do grep -q '^simd_count [1-9]' "$d/properties" 2>/dev/null && grep -q '^gfx_target_version [1-9]' "$d/properties" && echo x; done | wc -l)
	[ "$n" -ge 1 ] || { echo "run_node.sh: no GPU found on this node" >&2; exit 2; }
fi
job="${DVP_JOB_ID:-node$$-$(date +%s)}"
cores=$(nproc)
export HSA_ENABLE_IPC_MODE_LEGACY=0              # dmabuf IPC (RCCL across processes on this host driver)
export DVP_HOST_THREADS="${DVP_HOST_THREADS:-$(( cores / n > 1 ? cores / n : 1 ))}"   # host-thread budget per rank
pids=()
for ((r = 0; r < n; r++)); do
	HIP_VISIBLE_DEVICES=$r "$here/dvp-mvs_amd/apd" "$folder" --rank $r --world $n --job "$job" "$@" &
	pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || { s=$?; [ $rc -eq 0 ] && rc=$s; }; done
exit $rc
