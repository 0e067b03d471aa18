#!/bin/bash
# round 6, call ad: a read-only stream of 15 GB by the way it is asked for (scripts/ubench/read_stream.hip)
O=gpurun_out/r06ad; mkdir -p $O
timeout 600 scripts/ubench/read_stream > $O/read_stream.txt 2>&1; cat $O/read_stream.txt
