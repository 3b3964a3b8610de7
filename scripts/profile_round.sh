#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   kernel stats (three streams and --no-overlap) and the three PMC passes, each in its own run.
set -e
export TMPDIR=/tmp
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --profile --steps 3 --warmup 1 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --profile --no-overlap --steps 3 --warmup 1 > $O/stats1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --profile --steps 1 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --profile --steps 1 --warmup 1 > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 bench.py --profile --steps 1 --warmup 1 > $O/mfma.log 2>&1
find $O -name '*.csv' | sort
