"""ctypes binding of oracle/tfrec_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
REF_DRIVER = os.path.join(_HERE, "_ref", "ref_driver")
FM_BOUNDARY = os.path.join(_HERE, "_build", "fm_boundary")            # near-boundary discriminator input search
FM_RESOLVE_CHECK = os.path.join(_HERE, "_build", "fm_resolve_check")  # host build of the product's exact fm_dev slow path
REFERENCE_DIR = "/root/reference"

SLOT_NAMES = ("TFA_1", "TFA_2", "TFA_3", "TX22", "WHB")


class Event(C.Structure):
    _fields_ = [
        ("slot", C.c_int16),
        ("status", C.c_int16),
        ("byte_cnt", C.c_int32),
        ("rssi_db", C.c_int32),
        ("offset", C.c_int32),
        ("end_sample", C.c_int64),
        ("rssi_raw", C.c_int64),
        ("rdata", C.c_uint8 * 64),
    ]


class Data(C.Structure):
    _fields_ = [
        ("slot", C.c_int32),
        ("type", C.c_int32),
        ("id", C.c_uint64),
        ("temp", C.c_double),
        ("humidity", C.c_double),
        ("sequence", C.c_int32),
        ("alarm", C.c_int32),
        ("rssi", C.c_int32),
        ("flags", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the C restatement (and, where /root/reference exists, the real reference harness)."""
    src = [os.path.join(_HERE, f) for f in ("tfrec_oracle.c", "tfrec_oracle.h", "Makefile", "fm_boundary.c", "fm_resolve_check.cpp")]
    src += [os.path.join(os.path.dirname(_HERE), "tfrec_amd", "csrc", f) for f in ("fm_resolve.h", "fm_resolve_tables.h")]
    outs = [_SO, FM_BOUNDARY, FM_RESOLVE_CHECK]
    stale = any(not os.path.exists(o) for o in outs) or any(os.path.getmtime(s) > os.path.getmtime(o) for s in src for o in outs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if os.path.isdir(REFERENCE_DIR):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        for f in ("orc_set_log_bits", "orc_set_keep_dec", "orc_set_quiet"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.orc_process.restype = C.c_long
        L.orc_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_process_s16.restype = C.c_long
        L.orc_process_s16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_decim10.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_time_many.restype = C.c_double
        L.orc_time_many.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int]
        L.orc_hex.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_process_many.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                       C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_process_parts.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_num_events.restype = C.c_size_t
        L.orc_num_events.argtypes = [C.c_void_p]
        L.orc_events.restype = C.POINTER(Event)
        L.orc_events.argtypes = [C.c_void_p]
        L.orc_num_data.restype = C.c_size_t
        L.orc_num_data.argtypes = [C.c_void_p]
        L.orc_data.restype = C.POINTER(Data)
        L.orc_data.argtypes = [C.c_void_p]
        L.orc_text.restype = C.c_char_p
        L.orc_text.argtypes = [C.c_void_p]
        L.orc_bits_text.restype = C.c_char_p
        L.orc_bits_text.argtypes = [C.c_void_p]
        L.orc_num_dec.restype = C.c_size_t
        L.orc_num_dec.argtypes = [C.c_void_p]
        L.orc_dec.restype = C.POINTER(C.c_int16)
        L.orc_dec.argtypes = [C.c_void_p]
        L.orc_thresh.restype = C.c_int
        L.orc_thresh.argtypes = [C.c_void_p]
        L.orc_atan_uncertain.restype = C.c_uint64
        L.orc_atan_uncertain.argtypes = [C.c_void_p]
        L.orc_clear_logs.argtypes = [C.c_void_p]
        L.orc_fm_dev.restype = C.c_int
        L.orc_fm_dev.argtypes = [C.c_int] * 4
        L.orc_fm_dev_nrzs.restype = C.c_int
        L.orc_fm_dev_nrzs.argtypes = [C.c_int] * 4
        L.orc_crc8.restype = C.c_uint8
        L.orc_crc8.argtypes = [C.c_void_p, C.c_int]
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.orc_iir_coeffs.argtypes = [C.c_double, C.POINTER(C.c_double)]
        L.orc_iir_run.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_decimate.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def event_tuple(e) -> tuple:
    """Canonical comparable form of a flush event."""
    return (int(e.slot), int(e.end_sample), int(e.byte_cnt), int(e.rssi_db), int(e.offset), bytes(e.rdata))


def data_tuple(d) -> tuple:
    return (int(d.slot), int(d.type), int(d.id), float(d.temp), float(d.humidity), int(d.sequence), int(d.alarm),
            int(d.rssi), int(d.flags))


class Oracle:
    """One reference-equivalent receiver (one stream)."""

    def __init__(self, types_mask: int = 0x2F, thresh: int = 500, wide: int = 0, log_bits: bool = False,
                 keep_dec: bool = False, quiet: bool = False):
        self.L = lib()
        self.h = self.L.orc_create(types_mask, thresh, wide)
        self.L.orc_set_log_bits(self.h, int(log_bits))
        self.L.orc_set_keep_dec(self.h, int(keep_dec))
        self.L.orc_set_quiet(self.h, int(quiet))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process(self, iq: np.ndarray) -> int:
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        return int(self.L.orc_process(self.h, iq.ctypes.data, iq.size))

    def process_s16(self, x: np.ndarray) -> int:
        """Blocks of 65536 int16 values (already (I,Q) int16 at 1.536 MS/s: BASELINE config 5)."""
        a = np.ascontiguousarray(x, dtype=np.int16)
        return int(self.L.orc_process_s16(self.h, a.ctypes.data, a.size))

    def hex(self, data: bytes):
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        self.L.orc_hex(self.h, buf, len(data))

    def events(self):
        n = self.L.orc_num_events(self.h)
        p = self.L.orc_events(self.h)
        return [event_tuple(p[i]) for i in range(n)]

    def events_raw(self):
        n = self.L.orc_num_events(self.h)
        p = self.L.orc_events(self.h)
        return [(event_tuple(p[i]), int(p[i].rssi_raw)) for i in range(n)]

    def events_full(self):
        """event_tuple + the raw RSSI accumulator (tfa1.cpp:161, tfa2.cpp:373, whb.cpp:678) + what flush() made of the
        bytes (0 too short, 1 telegram, 2 rejected by CRC / sanity: tfa1.cpp:63-73, tfa2.cpp:93/237, whb.cpp:506-510)."""
        n = self.L.orc_num_events(self.h)
        p = self.L.orc_events(self.h)
        return [event_tuple(p[i]) + (int(p[i].rssi_raw), int(p[i].status)) for i in range(n)]

    def data(self):
        n = self.L.orc_num_data(self.h)
        p = self.L.orc_data(self.h)
        return [data_tuple(p[i]) for i in range(n)]

    def text(self) -> str:
        return self.L.orc_text(self.h).decode()

    def bits_text(self) -> str:
        return self.L.orc_bits_text(self.h).decode()

    def dec(self) -> np.ndarray:
        n = self.L.orc_num_dec(self.h)
        return np.ctypeslib.as_array(self.L.orc_dec(self.h), shape=(n,)).copy()

    def thresh(self) -> int:
        return int(self.L.orc_thresh(self.h))

    def clear(self):
        self.L.orc_clear_logs(self.h)


ORC_EVENT_DTYPE = np.dtype([("slot", "<i2"), ("status", "<i2"), ("byte_cnt", "<i4"), ("rssi_db", "<i4"), ("offset", "<i4"),
                            ("end_sample", "<i8"), ("rssi_raw", "<i8"), ("rdata", "u1", (64,))])
assert ORC_EVENT_DTYPE.itemsize == 96


def usable_threads() -> int:
    """Host threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
    except Exception:
        pass
    return n


def process_many(iq: np.ndarray, types_mask: int = 0x2F, thresh: int = 500, wide: int = 0, cap: int = 1024,
                 threads: int | None = None):
    """Fresh receivers over a whole batch iq[n_streams, n_bytes] (OpenMP, one stream per thread at a time).
    Returns a list with one ORC_EVENT_DTYPE array per stream."""
    a = np.ascontiguousarray(iq, dtype=np.uint8)
    n = a.shape[0]
    out = np.zeros((n, cap), dtype=ORC_EVENT_DTYPE)
    counts = np.zeros(n, dtype=np.int64)
    lib().orc_process_many(types_mask, thresh, wide, a.ctypes.data, a.strides[0], a.shape[1], n,
                           threads or usable_threads(), out.ctypes.data, cap, counts.ctypes.data)
    assert counts.max(initial=0) <= cap, "orc_process_many: raise cap"
    return [out[s, : counts[s]] for s in range(n)]


def process_parts(parts, types_mask: int = 0x2F, thresh: int = 500, wide: int = 0, reps=None, keep_from: int = 0,
                  cap: int = 4096, threads: int | None = None):
    """Receivers that CONTINUE over several batches: parts = [iq_0[n_streams, n_bytes_0], iq_1, ...]; stream s runs
    parts[p][s] reps[p] times in a row (default 1), parts in order, on one carried state -- the bytes of one long dump.
    Returns one ORC_EVENT_DTYPE array per stream with the events from part `keep_from` on (end_sample counted from the
    stream's first sample)."""
    # (a part may be a column slice of a longer array: rows need not be adjacent, bytes of a row must be)
    arrs = [p if (isinstance(p, np.ndarray) and p.dtype == np.uint8 and p.ndim == 2 and p.strides[1] == 1)
            else np.ascontiguousarray(p, dtype=np.uint8) for p in parts]
    n = arrs[0].shape[0]
    assert all(a.ndim == 2 and a.shape[0] == n for a in arrs)
    reps = list(reps) if reps is not None else [1] * len(arrs)
    bases = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    strides = (C.c_size_t * len(arrs))(*[a.strides[0] for a in arrs])
    nbytes = (C.c_size_t * len(arrs))(*[a.shape[1] for a in arrs])
    creps = (C.c_int * len(arrs))(*reps)
    out = np.zeros((n, cap), dtype=ORC_EVENT_DTYPE)
    counts = np.zeros(n, dtype=np.int64)
    lib().orc_process_parts(types_mask, thresh, wide, len(arrs), bases, strides, nbytes, creps, keep_from, n,
                            threads or usable_threads(), out.ctypes.data, cap, counts.ctypes.data)
    assert counts.max(initial=0) <= cap, "orc_process_parts: raise cap"
    return [out[s, : counts[s]] for s in range(n)]


def canon(ev: np.ndarray) -> np.ndarray:
    """Comparable matrix [n, 5 + 64 + 2] of ORC_EVENT_DTYPE events: slot, end_sample, byte_cnt, rssi_db, offset, rdata,
    rssi_raw, status."""
    m = np.empty((len(ev), 71), dtype=np.int64)
    for k, f in enumerate(("slot", "end_sample", "byte_cnt", "rssi_db", "offset")):
        m[:, k] = ev[f]
    m[:, 5:69] = ev["rdata"]
    m[:, 69] = ev["rssi_raw"]
    m[:, 70] = ev["status"]
    return m


# ------------------------------------------------------------------ real-reference helpers (this container only)

def have_reference() -> bool:
    return os.path.isdir(REFERENCE_DIR) and os.path.exists(REF_DRIVER)


def parse_ref_events(path: str):
    """Parse the event log written by ref_driver into (events, data, bits_text)."""
    ev, data, bits = [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "F":
                ev.append((int(p[1]), int(p[2]), int(p[3]), int(p[4]), int(p[5]), bytes.fromhex(p[6])))
            elif p[0] == "D":
                data.append((int(p[1]), int(p[2]), int(p[3], 16), float(p[4]), float(p[5]), int(p[6]), int(p[7]),
                             int(p[8]), int(p[9])))
            elif p[0] == "W":
                bits.append(line)
    return ev, data, "".join(bits)


def decim10(iq: np.ndarray) -> np.ndarray:
    """BASELINE config 5 front end (defined by the oracle, no reference counterpart): u8 IQ at 15.36 MS/s ->
    interleaved int16 IQ at 1.536 MS/s, from zero history."""
    a = np.ascontiguousarray(iq, dtype=np.uint8)
    n_in = a.size // 2
    assert n_in % 10 == 0
    out = np.empty(2 * (n_in // 10), dtype=np.int16)
    lib().orc_decim10(a.ctypes.data, n_in, out.ctypes.data)
    return out


def run_reference(iq_path: str, types_mask: int, thresh: int, wide: int, workdir: str, bits: bool = False,
                  in16: bool = False):
    """Run the real reference on an IQ file (u8, or raw int16 with in16); returns dict(text, events, data, bits, dec)."""
    evp = os.path.join(workdir, "ref.ev")
    decp = os.path.join(workdir, "ref.dec")
    out = subprocess.run([REF_DRIVER, "run16" if in16 else "run", "%x" % types_mask, str(thresh), str(wide), iq_path, evp, decp,
                          "1" if bits else "0"], capture_output=True, text=True, check=True)
    text = out.stdout.split("---\n", 1)[1]
    ev, data, bt = parse_ref_events(evp)
    dec = np.fromfile(decp, dtype=np.int16)
    return dict(text=text, events=ev, data=data, bits=bt, dec=dec)
