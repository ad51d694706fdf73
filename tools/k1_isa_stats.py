#!/usr/bin/env python3
"""Static instruction mix of path_kernel variants (device assembly of csrc/mcr_hip.hip, no GPU needed).

    python tools/k1_isa_stats.py [--kernel SUBSTR] [--flags "..."] [--keep out.s]

Compiles the translation unit to gfx950 assembly with the flags of csrc/build.py and prints, per kernel whose mangled
name contains SUBSTR (default: the headline count-only variant), the register budget and the number of instructions by
class.  Used for A/B comparisons of kernel edits before spending GPU time on them.
"""

from __future__ import annotations

import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

HEADLINE = "path_kernelILi0ELi0ELi3ELb0ELb0ELi0ELb0ELb0E"   # <MODE 0, Philox, both assets taxed, no annual tax, no injection, whole path, unsplit, no extended streams>


def classify(op: str) -> str:
    if op.startswith(("v_add_f64", "v_mul_f64", "v_fma_f64", "v_fmac_f64", "v_max_f64", "v_min_f64")):
        return "fp64_arith"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "fp64_trans"
    if op.startswith(("v_frexp", "v_ldexp", "v_cvt", "v_rndne", "v_fract", "v_trunc")):
        return "fp64_conv"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "v_mov"
    if op.startswith(("v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32")):
        return "v_intmul"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "v_lane"
    if op.startswith("v_"):
        return "v_int_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "s_wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "s_branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    from monte_carlo_retirement_amd.csrc import build as B

    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=HEADLINE)
    ap.add_argument("--flags", default="")
    ap.add_argument("--keep", default=None)
    args = ap.parse_args()
    src = os.path.join(B.HERE, "mcr_hip.hip")
    out = args.keep or os.path.join(tempfile.mkdtemp(prefix="k1isa_"), "mcr_hip.s")
    flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")] + B.PER_SOURCE_FLAGS.get("mcr_hip.hip", [])
    cmd = [B.hipcc(), *flags, *args.flags.split(), "--cuda-device-only", "-S", "-o", out, src]
    subprocess.check_call(cmd)
    text = open(out).read()
    # kernels: "<name>:" ... ".end_amdhsa_kernel"/".Lfunc_end"
    for m in re.finditer(r"^(_ZN3mcr\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if args.kernel not in name:
            continue
        mix = collections.Counter()
        for line in body.splitlines():
            line = line.split(";")[0].strip()
            if not line or line.endswith(":") or line.startswith("."):
                continue
            mix[classify(line.split()[0])] += 1
        meta = {}
        km = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
        if km:
            for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size"):
                mm = re.search(r"\.amdhsa_" + key + r"\s+(\S+)", km.group(1))
                if mm:
                    meta[key] = mm.group(1)
        valu = sum(v for k, v in mix.items() if k.startswith(("fp64", "v_")))
        print(name)
        print("  ", meta)
        print("   VALU total", valu, "| all", sum(mix.values()))
        for k, v in sorted(mix.items(), key=lambda kv: -kv[1]):
            print(f"   {k:12s} {v}")


if __name__ == "__main__":
    main()
