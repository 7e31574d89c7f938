#!/bin/bash
# A/B of environment knobs on the steady state of the bench workload: tools/ab_env.sh "VAR=a" "VAR=b" ...
cd $GRAFT_REPO_ROOT
for setting in "$@"; do
  echo -n "$setting : "; env $setting python3 tools/plan_experiment.py | tail -1
done
