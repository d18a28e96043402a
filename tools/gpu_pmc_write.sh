#!/bin/bash
# memory-side WRITE traffic of one GRCh38 batch per kernel (run under gpurun; its own --pmc pass)
export TMPDIR=/tmp ROUNDS_BRIEF=1
OUT=$PWD/gpurun_out/pmc_write
rm -rf $OUT; mkdir -p $OUT
python3 tools/gpu_rounds.py 4333 grch38 > $OUT/plain.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 tools/gpu_rounds.py 4333 grch38 > $OUT/w.txt 2>&1
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_write/w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"k_items<arx::(\w+)>|k_block_items<arx::(\w+)>", n) or re.search(r"arx::(k_\w+(?:<\d+>)?)", n)
        k = next((g for g in m.groups() if g), n[:40]) if m else n[:40]
        acc[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:14]:
    print(k, "dispatches", len(disp[k]), "WRITE_SIZE %.1f MB per dispatch (KB units x 1024)" % (v * 1024 / len(disp[k]) / 1e6))
PY
