#!/bin/bash
# Round-3 evidence run (GPU box, via gpurun): rocprofv3 kernel-trace stats of the driver's and the default bench command and of
# configs 2, 3 and 5's per-GPU workload; the --pmc passes (counters only, no trace domains, one small counter set per pass);
# the bench lines.  Raw output under gpurun_out/r03_profiles/; the files meant for the tracked profiles/ directory are
# assembled under gpurun_out/r03_profiles/profiles_out/ BY THIS SCRIPT (tools/pmc_traffic.py stamps traffic.json with the
# library's hash), and `python tools/collect_profiles.py` copies them into profiles/ back in the build container.
#   bash tools/r03_profiles.sh [quick]
cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/gpurun_out/r03_profiles
O=$P/profiles_out
rm -rf $P && mkdir -p $O
QUICK=$1

trace() {   # NAME bench-flags...: kernel-trace stats of one bench command
    local name=$1; shift
    ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_$name -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $P/trace_$name.log 2>&1 )
    find $P/trace_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03_${name}_kernel_stats.csv
    grep -h '^{' $P/trace_$name.log | tail -1 > $O/r03_${name}_line_under_rocprof.json
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
python3 tools/pmc_traffic.py $P/pmc $O r03 > $P/pmc_traffic.txt 2>&1
tail -30 $P/pmc_traffic.txt
# with the stamped traffic file in place: the bench lines the round quotes
mkdir -p profiles && cp $O/traffic.json profiles/traffic.json
python3 bench.py > $O/r03_bench_line.json 2> $P/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_driver_line.json 2>> $P/bench.err
if [ -z "$QUICK" ]; then
( python3 bench.py --rule mean --no-cpu-baseline --trained-steps 0
  python3 bench.py --n-tuple 4 --no-cpu-baseline --trained-steps 0
  python3 bench.py --n-tuple 6 --no-cpu-baseline --steps 100 --trained-steps 0
  python3 bench.py --n-tuple 3 --no-cpu-baseline --trained-steps 0
  python3 bench.py --n-tuple 2 --no-cpu-baseline --trained-steps 0
  python3 bench.py --workload env --steps 200
  python3 bench.py --workload eval --steps 200
  python3 bench.py --sync-at-one --no-cpu-baseline --no-mean-line
  python3 bench.py --steps 2000 --warmup 500 --repeats 1 --no-cpu-baseline --trained-steps 0 ) > $O/r03_other_workloads.jsonl 2>> $P/bench.err
fi
sha256sum 2048_amd/lib2048_hip.so > $O/r03_lib_sha256.txt
ls -la $O
