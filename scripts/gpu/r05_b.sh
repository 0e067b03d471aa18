#!/bin/bash
# round 5, call b: ONE output file on tmpfs, second look -- allocation (fallocate) taken out of the copies, the threads on the CPUs of one NUMA node as the
# tools bind themselves; the read side bound the same way; then the time line of today's one-stream and four-part runs on 64 M reads.
O=gpurun_out/r05b; mkdir -p $O
N0=$(cat /sys/devices/system/node/node0/cpulist)
{
echo "node0 cpus: $N0"; ls /sys/devices/system/node/ | grep node
for t in 4 8 16; do for m in p a A B Q w P r f; do taskset -c $N0 ./scripts/ubench/one_file_write $m /dev/shm 10 $t; done; done
echo "--- unbound"
for t in 8 16; do for m in a A B Q r; do ./scripts/ubench/one_file_write $m /dev/shm 10 $t; done; done
echo "--- 2 MiB blocks, bound"
for m in A B Q r; do taskset -c $N0 ./scripts/ubench/one_file_write $m /dev/shm 10 16 2; done
} > $O/one_file_write.txt 2>&1
READS=64000000 MATRIX="1:2,4:2" timeout 600 python scripts/e2e_parts.py > $O/e2e_timeline.txt 2>&1
cat $O/one_file_write.txt; cat $O/e2e_timeline.txt
