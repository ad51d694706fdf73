"""Builds csrc/libmcr_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m monte_carlo_retirement_amd.csrc.build [--force] [--verbose]
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["mcr_hip.hip", "mcr_aggregate.hip"]
HEADERS = ["mcr_device.h", "mcr_math.h", "mcr_tables.h", "mcr_numpy_rng.h", "mcr_numpy_tables.h", "mcr_host.h", os.path.join("..", "..", "include", "mcr.h")]
TARGET = os.path.join(HERE, "libmcr_hip.so")
ARCH = "gfx950"
# -ffp-contract=off: every a*b+c rounds twice, like the reference's Python floats.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", f"--offload-arch={ARCH}",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm)")


def is_stale() -> bool:
    if not os.path.exists(TARGET):
        return True
    t = os.path.getmtime(TARGET)
    deps = [os.path.join(HERE, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


# per-source flags.  The DEVICE code of the path TU never sees a NaN as an operand (scenario blocks are range-checked on
# the host, NaN outputs are stored as bit patterns): -fno-honor-nans there drops the v_max(x, x) canonicalisations in
# front of every min / max (10 instructions of the headline kernel).  Host code keeps NaN semantics (validate_params).
PER_SOURCE_FLAGS = {"mcr_hip.hip": ["-Xarch_device", "-fno-honor-nans"]}


def build(force: bool = False, verbose: bool = False, extra_flags=(), variant: str = "") -> str:
    """`variant` (with `extra_flags`, e.g. ["-DMCR_RQ_UNROLL=8"]): a second library csrc/libmcr_hip_<variant>.so for A/B
    measurements (selected at run time with MCR_HIP_LIBRARY); the default build is csrc/libmcr_hip.so."""
    if variant:
        return _build_to(os.path.join(HERE, f"libmcr_hip_{variant}.so"), f"_{variant}", verbose, extra_flags)
    if not force and not is_stale():
        return TARGET
    return _build_to(TARGET, "", verbose, extra_flags)


def _build_to(target: str, obj_suffix: str, verbose: bool, extra_flags) -> str:
    compile_flags = [f for f in FLAGS if f != "-shared"]
    objs = []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", f"{obj_suffix}.o"))
        cmd = [hipcc(), *compile_flags, *PER_SOURCE_FLAGS.get(src, []), *extra_flags, "-c", "-o", obj, os.path.join(HERE, src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", target, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra_flags=["-Rpass-analysis=kernel-resource-usage"] if "--verbose" in sys.argv else [])
    print(TARGET)
