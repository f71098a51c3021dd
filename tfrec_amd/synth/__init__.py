"""Deterministic synthetic 8-bit IQ streams (ctypes binding of iqgen.c).

Test/bench signal source only: produces the raw u8 interleaved IQ format the reference records
with ``-S`` and replays with ``-L`` (sdr.cpp:233-234, engine.cpp:67-81).  Recipes: SURVEY.md App. C.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libiqgen.so")
_SRC = os.path.join(_HERE, "iqgen.c")

PROTO_NAMES = ("TFA_1", "TFA_2", "TFA_3", "TX22", "WHB")
BLOCK_BYTES = 65536


class Truth(C.Structure):
    _fields_ = [
        ("proto", C.c_int32),
        ("nbytes", C.c_int32),
        ("start", C.c_int64),
        ("length", C.c_int64),
        ("amp_q4", C.c_int32),
        ("f0_hz", C.c_int32),
        ("frame", C.c_uint8 * 64),
    ]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(
            ["gcc", "-O2", "-std=c11", "-fopenmp", "-fPIC", "-shared", "-Wall", "-o", _SO, _SRC]
        )
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(_SO)
        lib.iqgen_stream.restype = C.c_int
        lib.iqgen_stream.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.POINTER(Truth), C.c_int]
        lib.iqgen_stream_rate.restype = C.c_int
        lib.iqgen_stream_rate.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.POINTER(Truth), C.c_int, C.c_int]
        lib.iqgen_stream_ex.restype = C.c_int
        lib.iqgen_stream_ex.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.POINTER(Truth), C.c_int, C.c_int, C.c_int]
        lib.iqgen_batch.restype = C.c_int
        lib.iqgen_batch.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib = lib
    return _lib


def gen_stream(seed: int, stream: int, n_blocks: int, proto_mask: int = 0x1F, noise_q8: int = 256,
               with_truth: bool = False, rate_mult: int = 1, corrupt_every: int = 0):
    """One stream of ``n_blocks`` blocks of 65536*rate_mult bytes -> uint8 array (and the planted bursts).
    rate_mult 10 = the 15.36 MS/s input of BASELINE config 5.  corrupt_every = c > 0: every c-th burst carries a planted
    fault (wrong checksum / right checksum but a field the decoder's sanity test rejects / frame cut short, in turn)."""
    lib = _load()
    out = np.empty(n_blocks * BLOCK_BYTES * rate_mult, dtype=np.uint8)
    cap = 4096
    truth = (Truth * cap)()
    n = lib.iqgen_stream_ex(seed, stream, n_blocks, proto_mask, noise_q8, out.ctypes.data, truth, cap, rate_mult,
                            corrupt_every)
    if not with_truth:
        return out
    recs = []
    for k in range(min(n, cap)):
        t = truth[k]
        recs.append(dict(proto=t.proto, start=t.start, length=t.length, amp=t.amp_q4 / 16.0, f0_hz=t.f0_hz,
                         frame=bytes(t.frame[: t.nbytes])))
    return out, recs


def gen_batch(seed: int, first_stream: int, n_streams: int, n_blocks: int, proto_mask: int = 0x1F,
              noise_q8: int = 256, out: np.ndarray | None = None) -> np.ndarray:
    """``n_streams`` streams, shape [n_streams, n_blocks*65536] uint8 (OpenMP over streams)."""
    lib = _load()
    if out is None:
        out = np.empty((n_streams, n_blocks * BLOCK_BYTES), dtype=np.uint8)
    assert out.dtype == np.uint8 and out.flags.c_contiguous and out.size == n_streams * n_blocks * BLOCK_BYTES
    lib.iqgen_batch(seed, first_stream, n_streams, n_blocks, proto_mask, noise_q8, out.ctypes.data)
    return out
