#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (run on the GPU box via gpurun); copies the stats CSV into gpurun_out/
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
shift
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 50 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-200
tail -1 $OUT/bench.log | cut -c1-400
