#!/usr/bin/env python3
"""Fingerprints of the device code of libmcr_hip.so's translation units (no GPU needed): every kernel's instruction stream
(device assembly with the flags of csrc/build.py, comments / labels' numbering / debug lines stripped) hashed, with its
register budget, so that a refactoring that must not change the generated code can be checked kernel by kernel.

    python tools/isa_fingerprint.py --out before.json            (then edit)
    python tools/isa_fingerprint.py --compare before.json [--flags "..."]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def fingerprints(extra_flags=()):
    from monte_carlo_retirement_amd.csrc import build as B

    res = {}
    tmp = tempfile.mkdtemp(prefix="isafp_")
    for src in B.SOURCES:
        out = os.path.join(tmp, src + ".s")
        flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")] + B.PER_SOURCE_FLAGS.get(src, [])
        subprocess.check_call([B.hipcc(), *flags, *extra_flags, "--cuda-device-only", "-S", "-o", out, os.path.join(B.HERE, src)],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
            name, body = m.group(1), m.group(2)
            lines = []
            for line in body.splitlines():
                line = line.split(";")[0].strip()
                if not line or line.startswith((".loc", ".file", ".cfi", ".p2align")):
                    continue
                lines.append(re.sub(r"\.LBB\d+_", ".LBB_", line))       # block labels are numbered per function
            km = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
            meta = {}
            if km:
                for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size"):
                    mm = re.search(r"\.amdhsa_" + key + r"\s+(\S+)", km.group(1))
                    if mm:
                        meta[key] = mm.group(1)
            res[name] = {"sha": hashlib.sha256("\n".join(lines).encode()).hexdigest()[:20], "insts": sum(1 for x in lines if not x.endswith(":")), **meta}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--compare")
    ap.add_argument("--flags", default="")
    args = ap.parse_args()
    fp = fingerprints(args.flags.split())
    if args.out:
        json.dump(fp, open(args.out, "w"), indent=1)
        print(f"{len(fp)} kernels -> {args.out}")
    if args.compare:
        old = json.load(open(args.compare))
        same = [k for k in fp if k in old and fp[k] == old[k]]
        changed = [k for k in fp if k in old and fp[k] != old[k]]
        print(f"{len(same)} kernels identical, {len(changed)} changed, {len(set(fp) - set(old))} new, {len(set(old) - set(fp))} gone")
        for k in changed:
            print("  changed:", k[:150], old[k], "->", fp[k])
        for k in sorted(set(fp) - set(old)):
            print("  new:", k[:150])
        for k in sorted(set(old) - set(fp)):
            print("  gone:", k[:150])
        sys.exit(1 if changed else 0)


if __name__ == "__main__":
    main()
