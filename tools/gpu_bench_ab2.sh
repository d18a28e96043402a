run() { echo -n "$* : "; env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --boundary-steps 0 --no-end-to-end 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['kernel_ms_per_step_alone']
print(round(d['value']), 'pairs/s', round(d['ms_per_step'],1), 'ms/step;', {k: a.get(k) for k in ('reg2aln','reg2aln_nw','compact','ext_init','ext_step','rescue_step','seed_gather','seed_merge','dedup','mapq')})"; }
for s in "$@"; do run $s || exit 1; done
