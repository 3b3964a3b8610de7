#!/bin/bash
# rocprofv3 kernel stats of ONE module's forward + backward (scripts/bench_module.py): scripts/prof_module.sh <module> <samples> -> gpurun_out/pm_<module>/kernel_stats.csv
set -e
export TMPDIR=/tmp
O=gpurun_out/pm_$1
rm -rf $O && mkdir -p $O
SV_B=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 scripts/bench_module.py $1 > $O/log.txt 2>&1
cp $(find $O -name '*_kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/st
tail -1 $O/log.txt
