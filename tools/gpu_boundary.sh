for v in "" 1; do
ARX_BENCH_SYNC_FETCH=$v timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 --boundary-steps 12 > gpurun_out/bnd$v.json 2> gpurun_out/bnd$v.err; echo "sync_fetch='$v' rc=$?"
python3 - <<PY
import json
d=json.loads(open('gpurun_out/bnd$v.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], 'boundary', round(d['boundary']['value']), d['boundary']['ms_per_step'], d['boundary']['matches_resident'])
print(d['host_phase_ms_per_batch']['boundary'])
PY
done
