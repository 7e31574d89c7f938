cd /root/repo
for cfg in "$@"; do
  name=${cfg%%:*}; kv=${cfg#*:}
  env $kv python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_cmp_$name.json 2> gpurun_out/r06_cmp_$name.err
  python -c "
import json
d=json.load(open('gpurun_out/r06_cmp_$name.json')); print('$name', round(d['value']/1e9,3), round(d['roofline']['ms_kernels'][[k for k in d['roofline']['ms_kernels'] if 'owner' in k][0]],4), round(d['mean_rule']['value']/1e9,3) if 'mean_rule' in d else None, round(d['trained_agent']['value']/1e9,3), {k: round(v,4) for k,v in d['trained_agent']['ms_kernels'].items()})
"
done
