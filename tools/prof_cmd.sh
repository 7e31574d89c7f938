#!/bin/bash
# rocprofv3 kernel-trace summary of any python command: tools/prof_cmd.sh NAME script.py [args]  (env passes through)
set -e
cd /tmp && export TMPDIR=/tmp
NAME=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$NAME
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$@" > $OUT/run.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python3 - $OUT/kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('  %-60s calls %6s avg %9.1f us  min %8.1f max %8.1f  %5.1f%%'%(r['Name'].replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, float(r['Percentage'])))
PY
