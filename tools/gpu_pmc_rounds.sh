#!/bin/bash
# SQ counters of one 350k-pair device batch (tools/gpu_rounds.py), summed per kernel (run under gpurun)
export TMPDIR=/tmp ROUNDS_BRIEF=1
OUT=$PWD/gpurun_out/pmc_rounds
rm -rf $OUT; mkdir -p $OUT
python3 tools/gpu_rounds.py 4333 grch38 > $OUT/plain.txt 2>&1   # warms the index cache
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- python3 tools/gpu_rounds.py 4333 grch38 > $OUT/sq.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_INSTS_FLAT --output-format csv -d $OUT/sq2 -- python3 tools/gpu_rounds.py 4333 grch38 > $OUT/sq2.txt 2>&1
python3 - <<'PY'
import csv, glob, collections, re, os
out = os.environ.get("OUT", "gpurun_out/pmc_rounds")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for d in ("sq", "sq2"):
    for f in glob.glob(os.path.join("gpurun_out/pmc_rounds", d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            m = re.search(r"k_items<arx::(\w+)>|k_block_items<arx::(\w+)>", n) or re.search(r"arx::(k_\w+(?:<\d+>)?)", n)
            k = next((g for g in m.groups() if g), n[:40]) if m else n[:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, d)].add(r["Dispatch_Id"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    print(k, "dispatches", len(disp[(k, "sq")]), {a: ("%.3g" % b) for a, b in sorted(v.items())})
PY
