for v in 1 0; do
  echo "== ARX_TEXT_BWD=$v"
  ARX_TEXT_BWD=$v ARX_SEED_STATS=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end --steps 3 --warmup 1 > gpurun_out/ab_bwd_$v.json 2> gpurun_out/ab_bwd_$v.err
  grep "seed stats" gpurun_out/ab_bwd_$v.err | sort | uniq -c | sort -k1nr | head -12
  python3 -c "
import json
d=json.loads(open('gpurun_out/ab_bwd_$v.json').read().strip().splitlines()[-1]); a=d['kernel_ms_per_step_alone']
print(round(d['value']), d['ms_per_step'], {k:a.get(k) for k in ('seed_fwd','seed_bwd','seed_bwd_wave','locate','seed_gather')}, d.get('index'), d['setup_s'])"
done
