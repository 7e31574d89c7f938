#!/bin/bash
# Round-2 evidence run (GPU box): kernel-trace stats of the driver's and the default bench command, PMC passes for the
# HBM traffic and for the L1 / address path of k_td_play, and the other workloads' bench lines.  Everything lands under
# gpurun_out/r02_profiles/; the summaries are then copied into profiles/ by hand.
cd $GRAFT_REPO_ROOT
P=gpurun_out/r02_profiles
mkdir -p $P
bash tools/prof_bench.sh r02_driver --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $P/prof_driver.txt 2>&1
cp gpurun_out/prof_r02_driver/kernel_stats.csv $P/r02_bench_driver_kernel_stats.csv; cp gpurun_out/prof_r02_driver/bench_line.json $P/r02_bench_driver_line_under_rocprof.json
echo "driver profile done"
bash tools/prof_bench.sh r02_default --no-cpu-baseline > $P/prof_default.txt 2>&1
cp gpurun_out/prof_r02_default/kernel_stats.csv $P/r02_bench_default_kernel_stats.csv; cp gpurun_out/prof_r02_default/bench_line.json $P/r02_bench_default_line_under_rocprof.json
echo "default profile done"
bash tools/prof_bench.sh r02_n6 --no-cpu-baseline --no-mean-line --n-tuple 6 --steps 50 --warmup 20 > $P/prof_n6.txt 2>&1
cp gpurun_out/prof_r02_n6/kernel_stats.csv $P/r02_bench_n6_kernel_stats.csv
echo "n6 profile done"
PMC_STEPS=20 PMC_WARMUP=20 bash tools/pmc_many.sh r02_profiles/pmc "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"
G2048_SORT_EVERY=0 PMC_STEPS=20 PMC_WARMUP=20 bash tools/pmc_many.sh r02_profiles/pmc_nosort "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
echo "pmc done"
python3 bench.py > $P/r02_bench_line.json 2> $P/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $P/r02_bench_driver_line.json 2>> $P/bench.err
( python3 bench.py --rule mean --no-cpu-baseline
  python3 bench.py --n-tuple 4 --no-cpu-baseline
  python3 bench.py --n-tuple 6 --no-cpu-baseline --steps 100
  python3 bench.py --n-tuple 3 --no-cpu-baseline
  python3 bench.py --n-tuple 2 --no-cpu-baseline
  python3 bench.py --workload env --steps 200
  python3 bench.py --workload eval --steps 200
  python3 bench.py --sync-at-one --no-cpu-baseline --no-mean-line
  python3 bench.py --steps 2000 --warmup 500 --repeats 1 --no-cpu-baseline ) > $P/r02_other_workloads.jsonl 2>> $P/bench.err
echo "benches done"
ls $P
