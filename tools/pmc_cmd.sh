#!/bin/bash
# rocprofv3 PMC pass of any python command: tools/pmc_cmd.sh NAME "COUNTERS" script.py [args]; prints per-kernel means
set -e
cd /tmp && export TMPDIR=/tmp
NAME=$1; shift
PMC=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$NAME
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 ${PMC_TIMEOUT:-200} rocprofv3 --pmc $PMC --output-format csv -d $OUT -- python3 "$@" > $OUT/run.log 2>&1
F=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        m = re.search(r'(k_[a-z_]+(<[^>]*>)?)', row['Kernel_Name'])
        k = m.group(1) if m else row['Kernel_Name'][:40]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in agg.items():
    if k.startswith('k_td_play'):
        print(k, {c: '%.4g' % (sum(v[-20:]) / len(v[-20:])) for c, v in d.items()})
PY
