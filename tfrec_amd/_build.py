"""In-tree build of the native libraries (hipcc cross-compiles gfx950 without a GPU)."""
from __future__ import annotations

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_SO = os.path.join(_PKG, "libtfrec_amd.so")
HOST_SO = os.path.join(_PKG, "libtfrec_host.so")

HIP_SOURCES = ["frontend.hip", "chains.hip", "chains2.hip", "capi.hip"]
# -ffp-contract=off: the demodulator biquads must round after every multiply and add (bit-exact parity);
# no fast-math anywhere.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
               "-Wall", "-Wno-unused-function"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc() -> str:
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def build_device_lib(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into tfrec_amd/libtfrec_amd.so."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ("tfrec_dev.h", "dsp_dev.h", "decoder_dev.h", "fm_resolve.h", "fm_resolve_tables.h", "whb_chain_asm.h")] + [ os.path.join(ROOT, "include", "tfrec_amd.h"), __file__]
    if force or _stale(LIB_SO, deps):
        objs = []
        for s in srcs:
            o = s[:-4] + ".o"
            if force or _stale(o, deps):
                cmd = [hipcc()] + HIPCC_FLAGS + ["-c", s, "-o", o]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
            objs.append(o)
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_SO] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_SO


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_device_lib(force, verbose)
    host = os.path.join(_PKG, "host", "Makefile")
    if os.path.exists(host):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(host)] + (["-B"] if force else []))
    from . import synth

    synth.build(force)


if __name__ == "__main__":
    build_all(verbose=True)
