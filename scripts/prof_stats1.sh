#!/bin/bash
# One rocprofv3 kernel-stats pass of the single-stream step (bench.py --profile --no-overlap) into gpurun_out/$1; extra args go to bench.py
set -e
export TMPDIR=/tmp
O=gpurun_out/$1
shift
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --profile --no-overlap --steps 3 --warmup 1 "$@" > $O/stats1.log 2>&1
f=$(find $O -name '*_kernel_stats.csv' | head -1)
cp $f $O/kernel_stats.csv
rm -rf $O/stats1
head -40 $O/kernel_stats.csv | cut -c1-180
