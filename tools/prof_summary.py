#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/gpu_profile.sh into the files kept under profiles/: per-kernel stats of the
kernel trace, counters summed per kernel, and the FETCH_SIZE calibration factor for scattered 64-byte block reads."""
import csv, glob, json, os, re, sys, collections

out = sys.argv[1]


def short(name):
    for pat in (r"k_items<arx::(\w+)>", r"k_block_items<arx::(\w+)>", r"arx::(k_\w+(?:<\d+>)?)", r"(k_calib_blocks)", r"rocprim.*?::(\w+_kernel)"):
        m = re.search(pat, name)
        if m:
            return m.group(1)
    return name[:60]


def find(d, pat):
    return sorted(glob.glob(os.path.join(out, d, "**", pat), recursive=True))


res = {}
for f in find("trace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [dict(kernel=short(r["Name"]), calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6,
                                avg_ms=float(r["AverageNs"]) / 1e6, pct=float(r["Percentage"])) for r in rows]
    import shutil
    shutil.copy(f, os.path.join(out, "kernel_stats.csv"))


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in find(d, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return {k: dict(v, dispatches=len(disp[k])) for k, v in acc.items()}


pmc = {}
for d in ("pmc_fetch", "pmc_sq", "pmc_sq2"):
    for k, v in counters(d).items():
        pmc.setdefault(k, {}).update({kk: vv for kk, vv in v.items() if kk != "dispatches"})
        pmc[k]["dispatches_" + d] = v["dispatches"]
res["pmc_by_kernel"] = pmc
cal = counters("pmc_calib").get("k_calib_blocks")
if cal:
    algo = cal["dispatches"] * (1 << 24) * 64
    raw = cal["FETCH_SIZE"] * 1024.0            # FETCH_SIZE is reported in KB
    res["fetch_calibration"] = dict(pattern="one random distinct 64-byte block per lane, four 16-byte loads (load_block)", algorithmic_bytes=algo,
                                    FETCH_SIZE_raw_bytes=raw, bytes_per_raw_byte=algo / raw if raw else None)
for nm in ("bench_plain", "bench_trace"):
    try:
        res[nm] = json.loads(open(os.path.join(out, nm + ".json")).read().strip().splitlines()[-1])
    except Exception as e:
        res[nm] = repr(e)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
ks = {r["kernel"]: r for r in res.get("kernel_stats", [])}
print("kernel trace (avg ms per launch):", {k: round(v["avg_ms"], 3) for k, v in list(ks.items())[:12]})
bt = res.get("bench_plain", {})
if isinstance(bt, dict):
    print("bench plain:", round(bt.get("value", 0)), "pairs/s; HIP-event avg seed launch", bt.get("roofline", {}).get("avg_launch_ms"))
print("fetch calibration:", res.get("fetch_calibration"))
for k in ("k_seed_bwd", "k_seed_bwd_wave", "k_seed_fwd1", "k_seed_fwd2", "k_strat_dyn", "k_locate_dyn", "k_sw_u8_g16<10>", "k_extend_g16<4>", "k_reg2aln_nw_g16"):
    if k in pmc:
        print(k, {a: b for a, b in pmc[k].items()})
