#!/bin/bash
# round-3 GPU call AE: is the sharded run's upload rate a NUMA matter?  The same run free, and pinned to each NUMA node's CPUs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
lscpu | grep -i -E "numa|socket|model name" | head -8
for d in /sys/class/drm/card*/device /sys/class/kfd/kfd/topology/nodes/*; do [ -f $d/numa_node ] && echo "$d numa_node $(cat $d/numa_node)"; done 2>/dev/null | head
ls /sys/devices/system/node/ | grep node
export READS=32000000 MATRIX="4:2"
echo "== free"; timeout 300 python scripts/e2e_parts.py 2>&1 | grep -E "^parts|input"
for n in $(ls /sys/devices/system/node/ | grep -E "^node[0-9]+$"); do
  cpus=$(cat /sys/devices/system/node/$n/cpulist)
  echo "== $n cpus $cpus"; timeout 300 taskset -c $cpus python scripts/e2e_parts.py 2>&1 | grep -E "^parts"
done
