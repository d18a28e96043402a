timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/bnd.json 2> gpurun_out/bnd.err; echo rc=$?
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bnd.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['parity_ok'], 'boundary', round(d['boundary']['value']), d['boundary']['ms_per_step'], d['boundary']['matches_resident'])
print(d['host_phase_ms_per_batch'])
PY
tail -2 gpurun_out/bnd.err
