for i in 1 2; do
for a in "" "--no-stagger --streams 3"; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end --steps 20 --warmup 3 $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a :', round(d['value']), round(d['ms_per_step'],1), d['host_phase_ms_per_batch']['resident'])"
done; done
