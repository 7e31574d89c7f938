#!/bin/bash
# several PMC passes of the bench command, two counters each (TA / TCP have few slots); summaries under gpurun_out/$1/
D=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $D
i=0
for pair in "$@"; do
  i=$((i+1))
  bash $GRAFT_REPO_ROOT/tools/pmc_bench.sh p$i "$pair" > $D/pmc_$i.json 2> $D/pmc_$i.err || echo "pass $i ($pair) failed" >> $D/failed.txt
  echo "pass $i done: $pair"
done
