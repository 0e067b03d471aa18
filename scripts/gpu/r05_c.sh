#!/bin/bash
# round 5, call c: ONE tmpfs file filled by staged threads (allocate / populate / copy), bound to the CPUs of NUMA node 0
O=gpurun_out/r05c; mkdir -p $O
N0=$(cat /sys/devices/system/node/node0/cpulist)
{
for cfg in "1 0 16 8 8 1" "1 0 16 8 8 0" "1 0 4 8 8 1" "1 0 4 8 8 0" "1 1 8 8 8 1" "1 1 16 8 8 1" "1 2 8 8 8 1" "1 4 8 8 8 1" "0 1 8 8 8 1" "0 2 8 8 8 1" "0 4 8 8 8 1" "0 8 8 8 8 1" \
           "1 1 8 8 32 1" "1 2 8 8 32 1" "1 1 8 2 32 1" "1 2 8 2 32 1" "0 2 8 2 32 1" "1 1 8 8 8 0" "0 2 8 8 8 0"; do
  taskset -c $N0 ./scripts/ubench/one_file_stages /dev/shm 10 $cfg
done
} > $O/one_file_stages.txt 2>&1
cat $O/one_file_stages.txt
