#!/bin/bash
# L1 -> L2 read requests and L1 tag look-ups of k_td_play per launch under the two lane-order keys (G2048_SORT_VALUES = 0: positions
# of the big tiles, round 2's key; 1: positions and values, shipped) for the fresh agent of the bench and for the same lanes after 3 000
# training steps under the mean rule.  Counters only (no trace domains), one pass per configuration.
cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/gpurun_out/pmc_sort_key
rm -rf $P && mkdir -p $P
for v in 0 1; do
  for leg in fresh trained; do
    if [ $leg = fresh ]; then FL="--steps 10 --warmup 10 --repeats 1 --no-cpu-baseline --no-mean-line --trained-steps 0"; else FL="--steps 10 --warmup 10 --repeats 1 --no-cpu-baseline --no-mean-line --trained-steps 3000"; fi
    d=$P/v${v}_$leg; mkdir -p $d
    ( cd /tmp && export TMPDIR=/tmp && export G2048_SORT_VALUES=$v && timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py $FL > $d/bench.log 2>&1 )
    echo "values=$v $leg rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, os, collections
P = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'pmc_sort_key')
for d in sorted(glob.glob(P + '/v*')):
    rows = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_td_play' in r['Kernel_Name']:
                rows[r['Counter_Name']].append(float(r['Counter_Value']))
    out = {k: sum(v[-10:]) / len(v[-10:]) for k, v in rows.items() if v}
    print(os.path.basename(d), {k: round(v / 1e6, 2) for k, v in out.items()}, 'M per launch (mean of the last 10 launches)')
PY
