#!/bin/bash
# Round-4 evidence run (GPU box, via gpurun): rocprofv3 kernel-trace stats of the driver's and the default bench command and of
# configs 2, 3 and 5's per-GPU workload; the --pmc passes (counters only, no trace domains, one small counter set per pass);
# the bench lines.  Raw output under gpurun_out/r04_profiles/; the files meant for the tracked profiles/ directory are
# assembled under gpurun_out/r04_profiles/profiles_out/ BY THIS SCRIPT (tools/pmc_traffic.py stamps traffic.json with the
# library's hash), and `python tools/collect_profiles.py` copies them into profiles/ back in the build container.
#   PART=1 bash tools/r04_profiles.sh [quick]      traces + PMC passes + traffic.json   (one gpurun call of <= 20 minutes)
#   PART=2 bash tools/r04_profiles.sh              the bench lines with the stamped traffic.json in profiles/, the other workloads, the look-ahead trials
cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/gpurun_out/r04_profiles
O=$P/profiles_out
PART=${PART:-1}
if [ "$PART" = 1 ]; then rm -rf $P; fi
mkdir -p $O
QUICK=$1

trace() {   # NAME bench-flags...: kernel-trace stats of one bench command
    local name=$1; shift
    ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_$name -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $P/trace_$name.log 2>&1 )
    find $P/trace_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_${name}_kernel_stats.csv
    grep -h '^{' $P/trace_$name.log | tail -1 > $O/r04_${name}_line_under_rocprof.json
    echo "trace $name done"
}

pmc() {     # CONFIG PASSNAME "COUNTERS" n batch bench-flags...
    local cfg=$1 pass=$2 ctrs=$3 n=$4 batch=$5; shift 5
    local d=$P/pmc/${cfg}__${pass}
    mkdir -p $d
    echo "{\"key\": \"$cfg\", \"n\": $n, \"batch\": $batch, \"counters\": \"$ctrs\", \"flags\": \"$*\"}" > $d/meta.json
    ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --pmc $ctrs --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $d/bench.log 2>&1 )
    echo "pmc $cfg $pass rc=$?"
}

if [ "$PART" = 1 ]; then
FAST="--steps 10 --warmup 10 --repeats 1 --no-cpu-baseline --no-mean-line --trained-steps 0"
trace bench_driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
# the same command without its two extra legs (mean rule on the same boards; 3 000 more training steps + a timing on the trained
# agent's boards): rocprofv3's per-kernel AVERAGE over the full command mixes three input distributions, this one is the
# sum-rule path the metric is quoted on (conditioning + warm-up + timed regions + the 20 event-timed steps)
trace bench_driver_timed_path --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-mean-line --trained-steps 0
trace bench_default --no-cpu-baseline
trace bench_n6 --n-tuple 6 --steps 50 --warmup 20 --no-cpu-baseline --no-mean-line --trained-steps 0
trace config2_env --workload env --steps 200
trace config3_eval --workload eval --steps 200
trace lookahead_values --workload lookahead --steps 64

for spec in "n5_b1048576 5" "n6_b1048576 6"; do
    set -- $spec; cfg=$1; n=$2
    pmc $cfg fetch "FETCH_SIZE" $n 1048576 --n-tuple $n $FAST
    pmc $cfg write "WRITE_SIZE" $n 1048576 --n-tuple $n $FAST
    pmc $cfg tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" $n 1048576 --n-tuple $n $FAST
    pmc $cfg tcc "TCC_HIT_sum TCC_MISS_sum" $n 1048576 --n-tuple $n $FAST
    pmc $cfg sq "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" $n 1048576 --n-tuple $n $FAST
    pmc $cfg grbm "GRBM_GUI_ACTIVE" $n 1048576 --n-tuple $n $FAST
    if [ -z "$QUICK" ]; then
        pmc $cfg ta "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" $n 1048576 --n-tuple $n $FAST
        pmc $cfg tcpstall "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" $n 1048576 --n-tuple $n $FAST
    fi
done
pmc eval_n3_b262144 fetch "FETCH_SIZE" 3 262144 --workload eval --steps 50 --warmup 10
pmc eval_n3_b262144 write "WRITE_SIZE" 3 262144 --workload eval --steps 50 --warmup 10
pmc env_b65536 fetch "FETCH_SIZE" 0 65536 --workload env --steps 50 --warmup 10
pmc env_b65536 write "WRITE_SIZE" 0 65536 --workload env --steps 50 --warmup 10
python3 tools/pmc_traffic.py $P/pmc $O r04 > $P/pmc_traffic.txt 2>&1
tail -30 $P/pmc_traffic.txt
mkdir -p profiles && cp $O/traffic.json profiles/traffic.json
sha256sum 2048_amd/lib2048_hip.so > $O/r04_lib_sha256.txt
fi
if [ "$PART" = 2 ]; then
# with the stamped traffic file in place: the bench lines the round quotes
python3 bench.py > $O/r04_bench_line.json 2> $P/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_driver_line.json 2>> $P/bench.err
# the device-resident look-ahead (csrc/lookahead.hip): an n = 5 agent trained in the run, then QAgent.trial greedy and with
# expectimax(3, 4, 6) on 100 games — the reference's best published configuration (README.md:131-146)
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_lookahead -- python3 $GRAFT_REPO_ROOT/tools/lookahead_trial.py 5 262144 10000000 100 3 4 6 > $O/r04_lookahead_trial_n5_100games_under_rocprof.txt 2>&1 )
find $P/trace_lookahead -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_lookahead_trial_kernel_stats.csv
echo "trace lookahead done"
python3 tools/lookahead_trial.py 5 262144 10000000 1000 3 4 6 > $O/r04_lookahead_trial_n5_1000games.txt 2>> $P/bench.err
if [ -z "$QUICK" ]; then
( python3 bench.py --rule mean --no-cpu-baseline --trained-steps 0
  python3 bench.py --n-tuple 4 --no-cpu-baseline --trained-steps 0
  python3 bench.py --n-tuple 6 --no-cpu-baseline --steps 100 --trained-steps 0
  python3 bench.py --n-tuple 3 --no-cpu-baseline --trained-steps 0
  python3 bench.py --n-tuple 2 --no-cpu-baseline --trained-steps 0
  python3 bench.py --workload env --steps 200
  python3 bench.py --workload eval --steps 200
  python3 bench.py --workload lookahead --steps 64
  python3 bench.py --sync-at-one --no-cpu-baseline --no-mean-line
  python3 bench.py --steps 2000 --warmup 500 --repeats 1 --no-cpu-baseline --trained-steps 0 ) > $O/r04_other_workloads.jsonl 2>> $P/bench.err
fi
ls -la $O
fi
