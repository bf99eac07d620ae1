#!/usr/bin/env python3
"""Condense rocprofv3 output (tools/profile_bench.sh) into profiles/<tag>_kernel_stats.csv and
profiles/<tag>_sweep_pmc.json (per-launch means for the sweep kernel, with the gfx950 FETCH_SIZE x2
correction of MI355X_MICROARCH.md applied in `hbm_bytes_per_launch`)."""
import csv, glob, json, os, shutil, sys, collections
out, tag = sys.argv[1], sys.argv[2]
def newest(pattern):
    """rocprofv3 names its files by pid; gpurun merges every call's files into the same directory here, so take
    the newest per directory"""
    by_dir = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


st = newest(os.path.join(out, "trace", "*", "*_kernel_stats.csv"))
if st:
    shutil.copy(st[0], "profiles/%s_bench_kernel_stats.csv" % tag)
import hashlib
def _is_sweep(name):
    return "sweep2_kernel" in name or "sweep_mfma" in name
h = hashlib.sha256()
for f in ("ibo_amd/csrc/sweep2.hip", "ibo_amd/csrc/sweep.hip", "ibo_amd/csrc/ibo_common.h"):     # the same list as bench.py:pmc_numbers
    h.update(open(f, "rb").read())
res = {"kernel": None, "source": "rocprofv3 --pmc, python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras", "counters": {},
       "source_sha": h.hexdigest()[:16]}
for f in newest(os.path.join(out, "pmc_*", "*", "*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if _is_sweep(r["Kernel_Name"]):
            res["kernel"] = r["Kernel_Name"].split("(")[0]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            res["vgpr"] = r.get("VGPR_Count"); res["lds"] = r.get("LDS_Block_Size"); res["wg"] = r.get("Workgroup_Size")
    for k, v in acc.items():
        res["counters"][k] = sum(v) / len(v)
c = res["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half the bytes of wide coalesced reads
    res["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    res["hbm_bytes_per_launch_uncorrected"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
    res["mfma_util_pct"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c:
    res["mfma_f64_flops_per_launch"] = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
tr = newest(os.path.join(out, "trace", "*", "*_kernel_trace.csv"))
if tr:
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0])) if _is_sweep(r["Kernel_Name"])]
    res["kernel_trace_mean_ms"] = sum(d) / len(d) / 1e6
    res["kernel_trace_launches"] = len(d)
for l in open(os.path.join(out, "bench_trace.log")):
    if l.startswith("{"):
        b = json.loads(l)
        res["bench_under_trace"] = {"value": b["value"], "kernel_ms_hip_events": b["roofline"]["kernel_ms"], "frac": b["roofline"]["frac"]}
json.dump(res, open("profiles/%s_sweep_pmc.json" % tag, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
