#!/usr/bin/env python3
"""Registers, spills, scratch and occupancy of every kernel of a .hip file, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage; hipcc cross-compiles gfx950 without a GPU).
usage: scripts/kernel_resources.py opal_amd/csrc/particles.hip [name-filter] [-- extra hipcc flags]"""
import re
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--")
    args, extra = args[:k], args[k + 1:]
src = args[0]
flt = args[1] if len(args) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print(f"{'kernel':100s} VGPR AGPR SGPR scratch occ sgpr-spill vgpr-spill LDS")
for b, n in zip(blocks, dem):
    if flt and flt not in n:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    v = [g(k) for k in ("VGPRs", "AGPRs", "SGPRs", r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", "SGPRs Spill",
                        "VGPRs Spill", r"LDS Size \[bytes/block\]")]
    print(f"{n[:100]:100s} {v[0]:4d} {v[1]:4d} {v[2]:4d} {v[3]:7d} {v[4]:3d} {v[5]:10d} {v[6]:10d} {v[7]:6d}")
