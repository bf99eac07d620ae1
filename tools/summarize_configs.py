#!/usr/bin/env python3
"""Condense tools/profile_configs.sh's rocprofv3 output into profiles/<tag>_<name>_kernel_stats.csv and
profiles/<tag>_<name>_pmc.json.  Counters are summed over the launches of the named kernels within one run of the workload
and divided by the units that run processes (theta-points, fits, sweeps); FETCH_SIZE gets the gfx950 x2 correction of
MI355X_MICROARCH.md.  `source_sha` = hash of the kernel sources (the list bench.py hashes for the same block)."""
import csv, glob, hashlib, json, os, shutil, sys, collections

out, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SOURCES = json.load(open(os.path.join(ROOT, "tools", "pmc_sources.json")))


def sha_of(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def newest(pattern):
    by_dir = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


# name -> (kernel-name substrings, units per run of the workload, what a unit is)
SPEC = {
    "c3": (("sweep2_kernel<2, 3",), -(1 << 22), "EI evaluation (bench.py --config c3: launches of 2^22 candidates, N=2048, D=8, Matern-5/2)"),
    "c5": (("chol_update3_kernel",), 4 * 64, "theta-point (tools/c5_only.py: 4 grids of 64)"),
    "fit4096": (("chol_pipe", "chol_step"), 8, "fit (tools/time_fit.py: the constructor's fit + 7)"),
    "fit2048": (("chol_pipe", "chol_step"), 8, "fit"),           # chol_pipe8_kernel since round 4 (chol_pipe_kernel / chol_step*_kernel before)
    "fit1024": (("chol_pipe", "chol_step"), 8, "fit"),
    "learn4096": (("chol_pipe", "chol_step", "wtw_kernel", "chol_update3", "syrk3", "nlml_grad"), 8, "NLML + gradient evaluation (tools/learn_only.py 4096 16 8): factorisation, W^T W, contraction"),
    "learn1024": (("chol_pipe", "chol_step", "wtw_kernel", "nlml_grad"), 8, "NLML + gradient evaluation (tools/learn_only.py 1024 16 8)"),
}
for name in ("c3", "c5", "fit4096", "fit2048", "fit1024", "learn4096", "learn1024", "gallery", "c4"):
    st = newest(os.path.join(out, name, "trace", "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, name)))
    if name not in SPEC:
        continue
    subs, units, unit_name = SPEC[name]
    res = {"workload": name, "kernels": list(subs), "unit": unit_name, "counters_per_unit": {}, "source_files": SOURCES[name],
           "source_sha": sha_of(SOURCES[name]), "source": "rocprofv3 --pmc (own runs) over the command in tools/profile_configs.sh"}
    launches = 0
    for f in newest(os.path.join(out, name, "pmc*", "*", "*_counter_collection.csv")):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if any(s in r["Kernel_Name"] for s in subs):
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k, v in acc.items():
            u = units if units and units > 0 else (n[k] * -units if units else n[k])      # negative: that many units per launch
            res["counters_per_unit"][k] = v / u
            launches = max(launches, n[k])
    res["launches_per_run"] = launches
    res["units_per_run"] = units if units and units > 0 else (launches * -units if units else launches)
    c = res["counters_per_unit"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        res["hbm_bytes_per_unit"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        res["hbm_read_bytes_per_unit"] = 2.0 * c["FETCH_SIZE"] * 1024.0
        res["hbm_write_bytes_per_unit"] = c["WRITE_SIZE"] * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        res["mfma_util_pct"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c:
        res["mfma_f64_flops_per_unit"] = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"] > 0:
        res["valu_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
    tr = newest(os.path.join(out, name, "trace", "*", "*_kernel_trace.csv"))
    if tr:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr[0])) if any(s in r["Kernel_Name"] for s in subs)]
        if d:
            res["kernel_ms_per_unit"] = sum(d) / 1e6 / (units if units and units > 0 else (len(d) * -units if units else len(d)))
            res["kernel_trace_launches"] = len(d)
    json.dump(res, open(os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (tag, name)), "w"), indent=1, sort_keys=True)
    print(name, json.dumps({k: res[k] for k in res if k not in ("counters_per_unit",)}, sort_keys=True))
