#!/usr/bin/env python3
"""Per-kernel resources of the SHIPPED build (hipcc -Rpass-analysis=kernel-resource-usage of the same sources and
flags as `make`), condensed into profiles/<tag>_kernel_resources.txt; and where the spill code of the two count
kernels sits (the main loops must be free of v_readlane / v_writelane / scratch traffic).

  python tools/kernel_resources.py r03      (runs `make asm` in the package first; no GPU needed)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "parallel-sequence-alignment_amd")
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
subprocess.run(["make", "-C", PKG, "asm"], check=True, capture_output=True)
usage = open(os.path.join(PKG, "build", "asm", "resource_usage.txt")).read()
asm = open(os.path.join(PKG, "build", "asm", "psa_device-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
out = [f"# kernel resources of the build of csrc sha256 {bench.csrc_sha256()} (tools/kernel_resources.py {tag})",
       "# kernel | VGPRs | AGPRs | SGPRs | SGPR spills | VGPR spills | scratch B/lane | occupancy waves/SIMD | static LDS B"]
for blk in re.split(r"remark: [^\n]*Function Name: ", usage)[1:]:
    name = blk.split(" ")[0]
    f = lambda key: re.search(key + r": (\d+)", blk).group(1)
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = demangled.split("(")[0].replace("void ", "").replace("psa::", "")
    out.append(" | ".join([short, f("VGPRs"), f("AGPRs"), f("TotalSGPRs"), f("SGPRs Spill"), f("VGPRs Spill"),
                           f(r"ScratchSize \[bytes/lane\]"), f(r"Occupancy \[waves/SIMD\]"), f(r"LDS Size \[bytes/block\]")]))
out.append("")
out.append("# main loops (basic blocks with > 400 v_bitop3_b32): VALU instructions, and spill / scratch code inside them")
COUNT = "_ZN3psa16psa_count_kernelILi16ELb%dELb%dEEEvNS_9CountArgsE"        # <16, RAW, CHUNKED>
NAMES = {COUNT % (0, 0): "psa_count_kernel<16> (one chunk per tile)", COUNT % (0, 1): "psa_count_kernel<16> (chunked)",
         COUNT % (1, 0): "psa_count_kernel<16, raw> (one chunk per tile)", COUNT % (1, 1): "psa_count_kernel<16, raw> (chunked)",
         "_ZN3psa20psa_count_duo_kernelENS_7DuoArgsE": "psa_count_duo_kernel"}
for kern in NAMES:
    i = asm.index(kern + ":")
    body = asm[i:asm.index(".Lfunc_end", i)]
    blocks, cur = [], ("entry", [])
    for ln in body.split("\n"):
        mt = re.match(r"^(\.LBB\d+_\d+):", ln)
        if mt:
            blocks.append(cur)
            cur = (mt.group(1), [])
        else:
            cur[1].append(ln)
    blocks.append(cur)
    total_lane = sum("v_readlane" in l or "v_writelane" in l for _, b in blocks for l in b)
    total_scratch = sum("scratch_" in l for _, b in blocks for l in b)
    short = NAMES[kern]
    for name, b in blocks:
        nb = sum("v_bitop3" in l for l in b)
        if nb > 400:
            valu = sum(l.strip().startswith("v_") for l in b)
            lanes = sum("v_readlane" in l or "v_writelane" in l for l in b)
            scratch = sum("scratch_" in l for l in b)
            out.append(f"{short} {name}: {valu} VALU ({nb} v_bitop3_b32), v_readlane/v_writelane {lanes}, scratch ops {scratch}")
    out.append(f"{short} whole kernel (prologue, epilogue, tail): v_readlane/v_writelane {total_lane}, scratch ops {total_scratch}")
path = os.path.join(ROOT, "profiles", f"{tag}_kernel_resources.txt")
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out))
