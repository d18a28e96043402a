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
# memory-side bytes per read of the seeding kernels (bench.py's roofline.traffic reads this file from profiles/r02/)
try:
    bp = res.get("bench_plain", {})
    cal = (res.get("fetch_calibration") or {}).get("bytes_per_raw_byte") or 1.0
    reads_per_launch = bp["roofline"]["reads_per_launch"]
    def per_read(names, launches_per_batch):
        raw = sum(pmc[k]["FETCH_SIZE"] for k in names if k in pmc and "FETCH_SIZE" in pmc[k])
        disp = max(pmc[k]["dispatches_pmc_fetch"] for k in names if k in pmc)
        return raw * 1024.0 * cal / (disp / launches_per_batch) / reads_per_launch
    traffic = dict(source="rocprofv3 --pmc FETCH_SIZE over `python3 " + "bench.py --no-cpu-baseline --boundary-steps 0" + "` (its own pass, tools/gpu_profile.sh); FETCH_SIZE x 1024 x calibration factor, per seeding-stage run of %d reads" % reads_per_launch,
                   calibration=res.get("fetch_calibration"), workload=bp["config"]["workload"],
                   bwd_fabric_bytes_per_read=per_read(["k_seed_bwd_g", "k_seed_bwd_g<16>", "k_seed_bwd_g<32>", "k_seed_bwd_g<64>", "k_seed_bwd", "k_seed_bwd2", "k_seed_bwd_wave", "KSeedBwdTail"], 2.0), fwd_fabric_bytes_per_read=per_read(["k_seed_fwd1", "k_seed_fwd2"], 1.0),
                   strat_fabric_bytes_per_read=per_read(["k_strat_dyn"], 1.0), locate_fabric_bytes_per_read=per_read(["k_locate_dyn", "KLocate"], 1.0),
                   algorithmic=dict(bwd=bp["roofline"]["algorithmic_bytes_per_read"], fwd=bp["roofline_fwd"]["algorithmic_bytes_per_read"],
                                    strat=bp["roofline_strat"]["algorithmic_bytes_per_read"], locate=bp["roofline_locate"]["algorithmic_bytes_per_read"]))
    json.dump(traffic, open(os.path.join(out, "seed_traffic.json"), "w"), indent=1)
    print("seed traffic per read:", {k: round(v) for k, v in traffic.items() if k.endswith("per_read")}, "algorithmic:", traffic["algorithmic"])
except Exception as e:
    print("seed traffic not derived:", repr(e))
# VALU / LDS figures of the DP kernels and of the roofline kernel (bench.py's roofline_sw and roofline.bound read this file from profiles/r03/)
try:
    bp = res.get("bench_plain", {})
    ks_ = {}
    for r in res.get("kernel_stats", []):      # template instances share a short name: sum them
        e = ks_.setdefault(r["kernel"], dict(total_ms=0.0, calls=0))
        e["total_ms"] += r["total_ms"]; e["calls"] += r["calls"]
    passes = bp.get("whole_path_passes", bp["steps"] + bp["warmup"] + 1)   # whole-path passes over the read set in the profiled command (priming + warm-up + timed + the "alone" pass)
    reads = 2.0 * bp["config"]["pairs_per_step_per_gpu"]
    groups = {"extend": (["k_extend_g16<4>", "k_extend_g16<7>", "k_extend_g16<10>", "k_extend_g16<16>", "k_extend_classes", "k_extend_b16", "k_extend_classes_b"], "cells_extend"),
              "sw_u8": (["k_sw_u8_g16<10>", "k_sw_u8_g16<16>"], "cells_u8"), "reg2aln_nw": (["k_reg2aln_nw_g16"], "cells_global"),
              "seed_bwd": (["k_seed_bwd_g"], None), "seed_fwd": (["k_seed_fwd1", "k_seed_fwd2"], None), "seed_strat": (["k_strat_dyn"], None), "locate": (["k_locate_dyn", "KLocate"], None)}
    swc = {}
    for key, (names, cells_key) in groups.items():
        have = [k for k in names if k in pmc and "SQ_INSTS_VALU" in pmc[k]]
        if not have:
            continue
        valu = sum(pmc[k]["SQ_INSTS_VALU"] for k in have); act = sum(pmc[k].get("SQ_ACTIVE_INST_VALU", 0.0) for k in have)
        lds = sum(pmc[k].get("SQ_INSTS_LDS", 0.0) for k in have); salu = sum(pmc[k].get("SQ_INSTS_SALU", 0.0) for k in have)
        wait = sum(pmc[k].get("SQ_WAIT_ANY", 0.0) for k in have); wcyc = sum(pmc[k].get("SQ_WAVE_CYCLES", 0.0) for k in have)
        total_ns = sum(ks_[k]["total_ms"] for k in have if k in ks_) * 1e6
        e = dict(kernels=have, SQ_INSTS_VALU=valu, SQ_INSTS_LDS=lds, SQ_INSTS_SALU=salu, trace_total_ms=total_ns / 1e6,
                 valu_busy=(act * 4.0 / (1024.0 * total_ns * 2.4)) if total_ns else None,     # rocprofv3's VALUBusy: SQ_ACTIVE_INST_VALU x 4 / SIMDs / cycles
                 lds_insts_per_valu=lds / valu if valu else None, wait_share=wait / wcyc if wcyc else None)
        if cells_key:
            cells = bp["work_per_read"][cells_key] * reads * passes
            e["lane_ops_per_cell"] = valu * 64.0 / cells if cells else None
            e["reference_cells_in_profiled_run"] = cells
        swc[key] = e
    json.dump(swc, open(os.path.join(out, "sw_counters.json"), "w"), indent=1)
    print("sw counters:", {k: {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("valu_busy", "lane_ops_per_cell", "lds_insts_per_valu", "wait_share")} for k, v in swc.items()})
except Exception as e:
    print("sw counters not derived:", repr(e))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
ks = {r["kernel"]: r for r in res.get("kernel_stats", [])}
print("kernel trace (avg ms per launch):", {k: round(v["avg_ms"], 3) for k, v in list(ks.items())[:12]})
bt = res.get("bench_plain", {})
if isinstance(bt, dict):
    print("bench plain:", round(bt.get("value", 0)), "pairs/s; HIP-event avg seed launch", bt.get("roofline", {}).get("avg_launch_ms"))
print("fetch calibration:", res.get("fetch_calibration"))
for k in ("k_seed_bwd_g", "k_rescue_heavy", "k_chain_heavy", "k_dedup_heavy", "k_seed_bwd", "k_seed_bwd_wave", "k_seed_fwd1", "k_seed_fwd2", "k_strat_dyn", "k_locate_dyn", "KLocate", "KSeedBwdTail", "k_sw_u8_g16<10>", "k_extend_g16<4>", "k_reg2aln_nw_g16"):
    if k in pmc:
        print(k, {a: b for a, b in pmc[k].items()})
