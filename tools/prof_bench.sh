#!/bin/bash
# rocprofv3 kernel-trace summary of a bench command (run on the GPU box via gpurun):
#   tools/prof_bench.sh NAME [bench.py flags]   ->  gpurun_out/prof_NAME/{kernel_stats.csv, bench.log, bench_line.json}
set -e
cd /tmp && export TMPDIR=/tmp
NAME=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$NAME
shift
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py "$@" > $OUT/bench.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
grep '^{"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
head -8 $OUT/kernel_stats.csv | cut -c1-160
