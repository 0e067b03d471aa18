#!/bin/bash
# round 5, call a: what the box gives ONE output file -- parallel pwrite / mapping / per-part files / parallel pread, and DMA straight
# from / into the page cache (hipHostRegister over file mappings); then the e2e lines of the tree as it stands, as the baseline.
O=gpurun_out/r05a; mkdir -p $O
{
nproc; grep -m1 "model name" /proc/cpuinfo; df -h /dev/shm | tail -1; uname -r; cat /sys/kernel/mm/transparent_hugepage/shmem_enabled
for t in 1 4 8 16 32; do for m in p f m d r; do ./scripts/ubench/one_file_write $m /dev/shm 10 $t; done; done
for b in 1 64; do for m in p m d; do ./scripts/ubench/one_file_write $m /dev/shm 10 16 $b; done; done
} > $O/one_file_write.txt 2>&1
timeout 300 ./scripts/ubench/register_filemap /dev/shm 4 > $O/register_filemap.txt 2>&1
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err
tail -c 1500 $O/one_file_write.txt; cat $O/register_filemap.txt; tail -c 600 $O/bench_cfg2.err
