#!/bin/bash
# planner parameter sweep (environment knobs of build_slices) on the steady state of the bench workload
cd $GRAFT_REPO_ROOT
for ac in 0.75 1.5 3.0; do for fr in 0.125 0.25 0.5; do
  echo -n "ADDCOST=$ac FIXEDRATIO=$fr : "; G2048_PLAN_ADDCOST=$ac G2048_PLAN_FIXEDRATIO=$fr python3 tools/plan_experiment.py | tail -1
done; done
for thr in 0.003 0.03 0.1; do
  echo -n "THR=$thr : "; G2048_PLAN_THR=$thr python3 tools/plan_experiment.py | tail -1
done
