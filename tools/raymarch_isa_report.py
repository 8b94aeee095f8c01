#!/usr/bin/env python3
"""Regenerates the listings and per-iteration instruction counts of profiles/r02/raymarch_loop_isa.md from the current
sources: compiles raymarch_kernels.hip to gfx950 assembly (hipcc -S, no GPU needed) and extracts (a) the hand-written
march loop = the ;;#ASMSTART..ASMEND block of raymarch_kernel<2, true, 2, true, false, ASM=true> (cubic-box variant) and
(b) hipcc's loop of the same kernel with ASM=false (the C++ march_fast).  Prints a JSON summary; --listing writes
the two listings to profiles/r02/raymarch_loop_{hand,hipcc}.s"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sdf-viewer_amd", "csrc", "raymarch_kernels.hip")


def kind(line):
    line = line.strip()
    if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
        return None
    op = line.split()[0]
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("global_") or op.startswith("buffer_"):
        return "VMEM"
    if op.startswith("v_"):
        return "VALU"
    return None


def count(lines):
    c = {}
    for l in lines:
        k = kind(l)
        if k:
            c[k] = c.get(k, 0) + 1
    c["total"] = sum(c.values())
    return c


def kernel(lines, name):
    start = [i for i, l in enumerate(lines) if l.startswith(name + ":")][0]
    end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]  # (early exits have s_endpgm too)
    return lines[start:end]


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rm.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                               "-fvisibility=hidden", "-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    pre = "_ZN4sdfv12_GLOBAL__N_115raymarch_kernelILi2ELb1ELi2ELb1ELb0ELb"
    asmk = kernel(lines, pre + "1ELb0EEEvNS_12RaymarchArgsE")
    ck = kernel(lines, pre + "0ELb0EEEvNS_12RaymarchArgsE")
    starts = [i for i, l in enumerate(asmk) if "ASMSTART" in l]
    ends = [i for i, l in enumerate(asmk) if "ASMEND" in l]
    blocks = [[l.strip() for l in asmk[a + 1:b] if l.strip()] for a, b in zip(starts, ends)]
    # the cubic-box variant is the block whose out-of-bounds test starts with v_max3_f32 of absolute values
    hand = [b for b in blocks if any(l.startswith("v_max3_f32") and "|" in l for l in b)][0]
    # layout: .Lloop ... common path ... back-branch, then (out of line) .Lfetch ... interior ... .Lborder ... , .Ldone
    i_loop = [i for i, l in enumerate(hand) if l.startswith(".Lloop")][0]
    i_back = [i for i, l in enumerate(hand) if l.startswith("s_cbranch_scc1") and ".Lloop" in l][0]
    i_fetch = [i for i, l in enumerate(hand) if l.startswith(".Lfetch")][0]
    i_border = [i for i, l in enumerate(hand) if l.startswith(".Lborder")][0]
    i_test = [i for i, l in enumerate(hand) if l.startswith("s_cbranch_scc1") and ".Lborder" in l][0]
    i_done = [i for i, l in enumerate(hand) if l.startswith(".Ldone")][0]
    hdr = [i for i, l in enumerate(ck) if "Inner Loop Header" in l][0]
    lo = max(i for i in range(hdr) if ck[i].strip().startswith("s_branch"))
    label = ck[hdr].split(":")[0]
    hi = [i for i in range(hdr, len(ck)) if "s_cbranch_execz" in ck[i] and ck[i].split()[-1] == label][0]
    comp = ck[lo + 1:hi + 1]
    # hipcc's cell fetch: from the block that converts the floors to integers to the first label after the last gather
    def is_block_start(l):
        return l.startswith(".LBB") or l.startswith("; %bb.")
    first_cvt = [i for i, l in enumerate(comp) if "v_cvt_i32_f32" in l][0]
    i14 = max(i for i in range(first_cvt) if is_block_start(comp[i]))
    last_load = max(i for i, l in enumerate(comp) if "global_load" in l)
    i18 = [i for i in range(last_load, len(comp)) if is_block_start(comp[i])][0]
    res = {"hand_common_path": count(hand[i_loop:i_back + 1]),
           "hand_fetch_block_interior_cell": count(hand[i_fetch:i_border]),
           "hand_fetch_block_border_cell": count(hand[i_fetch:i_test + 1] + hand[i_border:i_done]),
           "hipcc_common_path": count(comp[:i14] + comp[i18:]), "hipcc_fetch_block_listed": count(comp[i14:i18])}
    for name in (pre + "1ELb0EEEvNS_12RaymarchArgsE", pre + "0ELb0EEEvNS_12RaymarchArgsE"):
        i = [k for k, l in enumerate(lines) if ".name:" in l and name in l][0]
        meta = "\n".join(lines[i:i + 12])
        res["vgpr_sgpr_" + ("asm" if name.endswith("Lb1ELb0EEEvNS_12RaymarchArgsE") else "hipcc")] = \
            [int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)), int(re.search(r"\.sgpr_count:\s+(\d+)", meta).group(1))]
    print(json.dumps(res, indent=1))
    if "--listing" in sys.argv:
        open(os.path.join(ROOT, "profiles", "r02/raymarch_loop_hand.s"), "w").write("\n".join(hand) + "\n")
        open(os.path.join(ROOT, "profiles", "r02/raymarch_loop_hipcc.s"), "w").write("\n".join(l.rstrip() for l in comp) + "\n")


if __name__ == "__main__":
    main()
