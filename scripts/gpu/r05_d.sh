#!/bin/bash
# round 5, call d: ONE tmpfs file -- allocation and copies taking turns (gate), with and without the head start the device start-up gives the allocator
O=gpurun_out/r05d; mkdir -p $O
N0=$(cat /sys/devices/system/node/node0/cpulist)
{
for cfg in "1 0 8 8 8 1 1 0" "1 0 16 8 8 1 1 0" "1 0 8 8 32 1 1 0" "1 0 16 8 32 1 1 0" "1 0 16 8 64 1 1 0" "1 0 8 8 32 1 1 250" "1 0 16 8 32 1 1 250" "1 0 16 8 64 1 1 250" "1 0 16 2 128 1 1 250" "1 0 16 8 32 1 0 250" "1 0 32 8 32 1 1 250"; do
  taskset -c $N0 ./scripts/ubench/one_file_stages /dev/shm 9.3 $cfg
done
} > $O/one_file_gate.txt 2>&1
cat $O/one_file_gate.txt
