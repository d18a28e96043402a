#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
ARX_E2E_TRACE=1 timeout -k 10 400 python3 bench.py --no-cpu-baseline --boundary-steps 0 --steps 3 > $O/a.json 2> $O/a.err; grep "e2e" $O/a.json | head -40; python3 -c "
import json; d=json.loads(open('$O/a.json').read().strip().splitlines()[-1]); print(d['value'], d['end_to_end']['value'], d['end_to_end']['worker_seconds'])"
ARX_E2E_TRACE=1 ARX_KMER_K=12 timeout -k 10 400 python3 bench.py --no-cpu-baseline --boundary-steps 0 --steps 3 > $O/b.json 2> $O/b.err; python3 -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('K12', d['value'], d['end_to_end']['value'], d['end_to_end']['worker_seconds'])"
