#!/bin/bash
# rocprofv3 PMC pass (counters only, no trace domains) of the bench command; one JSON summary per call:
#   tools/pmc_bench.sh NAME "COUNTER COUNTER ..." [bench.py flags]
set -e
cd /tmp && export TMPDIR=/tmp
NAME=$1; shift
PMC=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $PMC --output-format csv -d $OUT -- python3 bench.py --steps ${PMC_STEPS:-10} --warmup ${PMC_WARMUP:-10} --repeats 1 --no-cpu-baseline --no-mean-line "$@" > $OUT/bench.log 2>&1
F=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        import re
        m = re.search(r'(k_[a-z_]+(<\d+(, *\d+)?>)?)', row['Kernel_Name'])
        k = m.group(1) if m else row['Kernel_Name'][:40]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
import json
out = {k: {c: {'launches': len(v), 'mean': sum(v) / len(v), 'mean_last10': sum(v[-10:]) / len(v[-10:])} for c, v in d.items()} for k, d in agg.items() if k.startswith('k_td') or k.startswith('k_apply')}
print(json.dumps(out, indent=1))
PY
