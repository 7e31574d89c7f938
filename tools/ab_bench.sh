#!/bin/bash
# A/B of environment knobs on the whole bench: tools/ab_bench.sh "VAR=a" "VAR=b" ...  (prints ms_per_step)
cd $GRAFT_REPO_ROOT
for setting in "$@"; do
  echo -n "$setting : "; env $setting python3 bench.py --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.4f  play %.3f update %.3f' % (d['ms_per_step'], d['roofline']['ms_k_td_play'], d['roofline']['ms_k_td_update']))"
done
