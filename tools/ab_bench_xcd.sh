for i in 1 2 3; do
  for v in 0 1; do
    G2048_PLAN_XCD_BALANCE=$v python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-mean-line --trained-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcd_balance=$v driver', round(d['value']/1e9,3), d['ms_per_step'], {k:round(x,4) for k,x in d['roofline']['ms_kernels'].items()})"
    G2048_PLAN_XCD_BALANCE=$v python bench.py --no-cpu-baseline --no-mean-line --trained-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcd_balance=$v default', round(d['value']/1e9,3), d['ms_per_step'], {k:round(x,4) for k,x in d['roofline']['ms_kernels'].items()})"
  done
done
