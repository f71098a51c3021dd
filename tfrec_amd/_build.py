"""In-tree build of the native libraries (hipcc cross-compiles gfx950 without a GPU).

Two builds of the same HIP sources (tfrec_amd/csrc/knobs.h):
  libtfrec_amd.so      the product: no environment knobs, no what-if branches, no test hooks in the binary;
  libtfrec_amd_exp.so  -DTFREC_AMD_EXPERIMENTS: the knobs read from the environment (tests that drive a hook, A/B sessions).

Staleness is decided by CONTENT, not by time stamps: every object and library has a side file `<target>.stamp` holding the
SHA-256 of everything that went into it (sources, headers, flags, this file).  A tree that arrives with prebuilt objects of
other sources -- or with fresh time stamps on old ones, as a snapshot copy makes them -- is rebuilt; `build_all(force=True)`
rebuilds regardless.  `last_actions()` lists what the last call compiled or found current (the driver's build() prints it).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_SO = os.path.join(_PKG, "libtfrec_amd.so")
LIB_EXP_SO = os.path.join(_PKG, "libtfrec_amd_exp.so")
HOST_SO = os.path.join(_PKG, "libtfrec_host.so")

HIP_SOURCES = ["frontend.hip", "chains.hip", "chains2.hip", "capi.hip"]
# every header of csrc/ goes into every object's content hash (the stage headers chains2.hip is cut into included)
HIP_HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))
# -ffp-contract=off: the demodulator biquads must round after every multiply and add (bit-exact parity);
# no fast-math anywhere.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
               "-Wall", "-Wno-unused-function"]
EXP_FLAGS = ["-DTFREC_AMD_EXPERIMENTS"]

_actions: list[str] = []


def last_actions() -> list[str]:
    return list(_actions)


def _digest(paths: list[str], extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def _current(target: str, digest: str) -> bool:
    try:
        with open(target + ".stamp") as f:
            return os.path.exists(target) and f.read().strip() == digest
    except OSError:
        return False


def _stamp(target: str, digest: str) -> None:
    with open(target + ".stamp", "w") as f:
        f.write(digest + "\n")


def hipcc() -> str:
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def _build_variant(lib_so: str, suffix: str, extra_flags: list[str], force: bool, verbose: bool) -> str:
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = [os.path.join(CSRC, h) for h in HIP_HEADERS] + [os.path.join(ROOT, "include", "tfrec_amd.h")]
    flags = HIPCC_FLAGS + extra_flags
    todo = []
    objs = []
    for s in srcs:
        o = s[:-4] + suffix + ".o"
        objs.append(o)
        d = _digest([s] + deps, " ".join(flags))
        if force or not _current(o, d):
            todo.append((s, o, d))
        else:
            _actions.append("current  " + os.path.relpath(o, ROOT))

    def compile_one(job):
        s, o, d = job
        cmd = [hipcc()] + flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        _stamp(o, d)
        return "compiled " + os.path.relpath(o, ROOT)

    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1)) as ex:
            _actions.extend(ex.map(compile_one, todo))
    dl = _digest(objs, "link " + " ".join(flags))
    if force or todo or not _current(lib_so, dl):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_so] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        _stamp(lib_so, _digest(objs, "link " + " ".join(flags)))
        _actions.append("linked   " + os.path.relpath(lib_so, ROOT))
    else:
        _actions.append("current  " + os.path.relpath(lib_so, ROOT))
    return lib_so


def build_device_lib(force: bool = False, verbose: bool = False, experiments: bool = False) -> str:
    """Compile every HIP source for gfx950 into tfrec_amd/libtfrec_amd.so (or, experiments=True, libtfrec_amd_exp.so)."""
    if experiments:
        return _build_variant(LIB_EXP_SO, ".exp", EXP_FLAGS, force, verbose)
    return _build_variant(LIB_SO, "", [], force, verbose)


def build_all(force: bool = False, verbose: bool = False) -> None:
    del _actions[:]
    with ThreadPoolExecutor(max_workers=2) as ex:  # the two variants beside each other (hipcc is one thread per file)
        a = ex.submit(build_device_lib, force, verbose, False)
        b = ex.submit(build_device_lib, force, verbose, True)
        a.result()
        b.result()
    host = os.path.join(_PKG, "host", "Makefile")
    if os.path.exists(host):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(host)] + (["-B"] if force else []))
    from . import synth

    synth.build(force)


if __name__ == "__main__":
    import sys

    build_all(force="--force" in sys.argv, verbose=True)
    print("\n".join(last_actions()))
