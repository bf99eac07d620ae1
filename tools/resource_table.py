#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage over every kernel source -> profiles/<tag>_kernel_resource_usage.tsv
(registers, scratch, spills, LDS per kernel instantiation).  `python3 tools/resource_table.py r02`"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
rows = []
units = [("sweep2_fam", ["-DS2_FAM=FAM_%s" % fam, "-DS2_PIECE=%d" % pc]) for fam in ("SE", "M3", "M5") for pc in (0, 1)]
units += [(f, []) for f in ("sweep2", "sweep", "small2", "update3", "linalg", "assemble", "legacy", "comm", "abi_core", "abi_fit", "abi_sweep", "abi_nlml", "abi_legacy")]
for f, defs in units:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage"] + defs +
                         ["-c", os.path.join(root, "ibo_amd", "csrc", f + ".hip"), "-o", "/dev/null"], capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark: .*?(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"file": ("sweep2_kernels.h" if f == "sweep2_fam" else f + ".hip"), "kernel": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
cols = ["file", "kernel", "VGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
with open(os.path.join(root, "profiles", "%s_kernel_resource_usage.tsv" % tag), "w") as o:
    o.write("# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage, every kernel of libibo_hip.so\n")
    o.write("# default large-batch path: sweep2_kernel / acq_finish_kernel (+ sweep2_rank1_kernel in gallery rounds); sweep_mfma_kernel<..., true> is the\n")
    o.write("# small-batch SPLIT form (DIRECT's batches), sweep_mfma_kernel<..., false> the fallback when the dot form is not admissible\n")
    o.write("\t".join(cols) + "\n")
    for r in rows:
        o.write("\t".join(r.get(c, "") for c in cols) + "\n")
print(len(rows), "kernels")
