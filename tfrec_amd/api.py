"""ctypes binding of the C ABI in include/tfrec_amd.h.

The product is the HIP library ``libtfrec_amd.so``; this module only loads it, marshals arguments and
exposes the events as numpy records.  There is NO CPU fallback: if the library is missing or no GPU is
present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

BLOCK_BYTES = 65536
BLOCK_DEC = 8192
NSLOTS = 5
FIFO_DEPTH = 4  # TFREC_AMD_FIFO_DEPTH
SLOT_NAMES = ("TFA_1", "TFA_2", "TFA_3", "TX22", "WHB")

F_ALL_FLUSHES = 1
F_TIMING = 2
F_SERIAL_CHAINS = 4
F_INPUT_10X = 8
F_BITS = 16
STATUS_BITS = 0x80

E_OK, E_INVAL, E_NOMEM, E_HIP, E_OVERFLOW, E_STATE = 0, -1, -2, -3, -4, -5


class TfrecAmdError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("tfrec_amd error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("n_streams", C.c_int32),
        ("types_mask", C.c_int32),
        ("thresh", C.c_int32),
        ("filter_type", C.c_int32),
        ("device", C.c_int32),
        ("max_blocks", C.c_int32),
        ("max_events", C.c_int32),
        ("flags", C.c_uint32),
    ]


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("frontend_ms", "chains_ms", "total_ms", "windows_ms", "spec_biquad_ms",
                                         "repair_biquad_ms", "fix_biquad_ms", "slicer_ms", "coop_slicer_ms", "decode_ms",
                                         "commit_ms", "whb_biquad_ms", "whb_demod_ms", "whb_decode_ms", "whb_commit_ms",
                                         "tfa1_slicer_ms", "tfa1_coop_slicer_ms", "tfa1_decode_commit_ms", "fmdev_ms",
                                         "whb_verify_ms")]


class Stats(C.Structure):
    _fields_ = [("biquad_segments", C.c_uint64), ("biquad_unconverged", C.c_uint64), ("biquad_serial", C.c_uint64),
                ("tfa2_resliced", C.c_uint64), ("tfa1_recomputed", C.c_uint64), ("biquad_repair_slots", C.c_uint64),
                ("whb_respeculated", C.c_uint64), ("tfa1_scalar_groups", C.c_uint64), ("tfa2_scalar_groups", C.c_uint64),
                ("tfa1_vector_groups", C.c_uint64), ("tfa2_vector_groups", C.c_uint64)]


class FmStats(C.Structure):
    _fields_ = [("resolved", C.c_uint64), ("host_verified", C.c_uint64), ("host_mismatch", C.c_uint64),
                ("undecidable", C.c_uint64), ("reserved", C.c_uint64 * 4)]


EVENT_DTYPE = np.dtype(
    [
        ("stream", "<u4"),
        ("slot", "u1"),
        ("status", "u1"),
        ("byte_cnt", "<u2"),
        ("offset", "<i4"),
        ("seq", "<u4"),
        ("end_sample", "<i8"),
        ("rssi_raw", "<i8"),
        ("rdata", "u1", (64,)),
    ]
)
assert EVENT_DTYPE.itemsize == 96

# every symbol include/tfrec_amd.h declares
EXPORTS = (
    "tfrec_amd_version", "tfrec_amd_strerror", "tfrec_amd_last_error", "tfrec_amd_create", "tfrec_amd_destroy",
    "tfrec_amd_submit_device", "tfrec_amd_submit_host", "tfrec_amd_sync", "tfrec_amd_drain_events",
    "tfrec_amd_pending_events", "tfrec_amd_rssi_db", "tfrec_amd_read_decimated", "tfrec_amd_atan_uncertain",
    "tfrec_amd_get_timings", "tfrec_amd_read_thresh", "tfrec_amd_get_stats", "tfrec_amd_get_layout", "tfrec_amd_host_alloc",
    "tfrec_amd_host_free", "tfrec_amd_read_stage0", "tfrec_amd_get_fm_stats", "tfrec_amd_fm_dev_probe",
    "tfrec_amd_fifo_depth", "tfrec_amd_get_memory", "tfrec_amd_iir_probe",
)

_libs = {}


def library_path(experiments: bool = False) -> str:
    return _build.LIB_EXP_SO if experiments else _build.LIB_SO


def load_library(build: bool = True, experiments: bool = False):
    """Load libtfrec_amd.so -- or, experiments=True, libtfrec_amd_exp.so: the same sources with the environment knobs and
    test hooks compiled in (csrc/knobs.h) -- building it in-tree first when hipcc is available.  Raises if absent."""
    key = bool(experiments)
    if key in _libs:
        return _libs[key]
    if build:
        try:
            _build.build_device_lib(experiments=key)
        except (OSError, FileNotFoundError):
            pass  # no hipcc on this box: use the prebuilt library that travelled with the tree
    lib_so = library_path(key)
    if key:
        lib_so = os.environ.get("TFREC_AMD_LIB", lib_so)  # (A/B sessions: an alternative experiments build)
    if not os.path.exists(lib_so):
        raise RuntimeError("HIP extension %s is missing: run __graft_entry__.build()" % lib_so)
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7; importing torch first makes our
    # library bind to that same copy (same SONAME) instead of loading /opt/rocm's next to it, which would
    # leave whichever runtime comes second without a GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(lib_so)
    L.tfrec_amd_version.restype = C.c_char_p
    L.tfrec_amd_strerror.restype = C.c_char_p
    L.tfrec_amd_strerror.argtypes = [C.c_int]
    L.tfrec_amd_last_error.restype = C.c_char_p
    L.tfrec_amd_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.tfrec_amd_destroy.argtypes = [C.c_void_p]
    L.tfrec_amd_submit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.tfrec_amd_submit_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.tfrec_amd_sync.argtypes = [C.c_void_p]
    L.tfrec_amd_drain_events.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.tfrec_amd_pending_events.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.tfrec_amd_rssi_db.argtypes = [C.c_int, C.c_int64]
    L.tfrec_amd_rssi_db.restype = C.c_int
    L.tfrec_amd_read_decimated.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.tfrec_amd_read_stage0.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.tfrec_amd_atan_uncertain.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.tfrec_amd_get_timings.argtypes = [C.c_void_p, C.POINTER(Timings)]
    L.tfrec_amd_read_thresh.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.tfrec_amd_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.tfrec_amd_get_layout.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.tfrec_amd_get_memory.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.tfrec_amd_get_fm_stats.argtypes = [C.c_void_p, C.POINTER(FmStats)]
    L.tfrec_amd_fm_dev_probe.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(FmStats)]
    L.tfrec_amd_iir_probe.argtypes = [C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    L.tfrec_amd_fifo_depth.restype = C.c_int
    if L.tfrec_amd_fifo_depth() != FIFO_DEPTH:
        raise RuntimeError("libtfrec_amd.so was built with FIFO depth %d, this binding expects %d" % (
            L.tfrec_amd_fifo_depth(), FIFO_DEPTH))
    _libs[key] = L
    return L


def _check(L, rc: int, ok=(E_OK,)):
    if rc not in ok:
        detail = L.tfrec_amd_last_error().decode() if rc == E_HIP or rc == E_INVAL or rc == E_NOMEM else ""
        raise TfrecAmdError(rc, L.tfrec_amd_strerror(rc).decode() + (" (" + detail + ")" if detail else ""))
    return rc


def rssi_db(slot: int, rssi_raw: int) -> int:
    return int(load_library().tfrec_amd_rssi_db(int(slot), int(rssi_raw)))


class Receiver:
    """A batch of ``n_streams`` independent receivers on one GPU (one C-ABI context).

    ``submit`` replaces, for every stream, the reference's per-block
    ``process_iq`` + ``fsk_demod::process`` (engine.cpp:85-86); ``drain`` returns the decoder flush events.
    """

    def __init__(self, n_streams: int, types_mask: int = 0x2F, thresh: int = 500, filter_type: int = 0,
                 device: int = 0, max_blocks: int = 48, max_events: int | None = None, all_flushes: bool = False,
                 timing: bool = False, serial_chains: bool = False, input_10x: bool = False, bits: bool = False,
                 experiments: bool = False):
        # experiments=True: the build that reads the TFREC_AMD_* knobs / test hooks from the environment (csrc/knobs.h);
        # the default is the product library, which has none
        self.L = load_library(experiments=experiments)
        if max_events is None:
            max_events = max(4096, n_streams * max_blocks * 4 * (8 if all_flushes else 2))
        flags = ((F_ALL_FLUSHES if all_flushes else 0) | (F_TIMING if timing else 0)
                 | (F_SERIAL_CHAINS if serial_chains else 0) | (F_INPUT_10X if input_10x else 0) | (F_BITS if bits else 0))
        self.block_bytes = BLOCK_BYTES * (10 if input_10x else 1)
        self.cfg = Config(n_streams, types_mask, thresh, filter_type, device, max_blocks, max_events, flags)
        self.h = C.c_void_p()
        _check(self.L, self.L.tfrec_amd_create(C.byref(self.cfg), C.byref(self.h)))
        self.n_streams = n_streams
        self.max_events = max_events
        self._keep = ()

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.tfrec_amd_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def submit(self, iq, n_blocks: int | None = None, stream=None):
        """iq: torch uint8 CUDA tensor [n_streams, n_bytes] (resident in HBM) or a numpy/host array."""
        if isinstance(iq, np.ndarray):
            a = np.ascontiguousarray(iq, dtype=np.uint8).reshape(self.n_streams, -1)
            nb = a.shape[1] // self.block_bytes if n_blocks is None else n_blocks
            _check(self.L, self.L.tfrec_amd_submit_host(self.h, a.ctypes.data, a.strides[0], nb))
            return nb
        import torch

        assert iq.is_cuda and iq.dtype == torch.uint8 and iq.dim() == 2 and iq.shape[0] == self.n_streams
        assert iq.stride(1) == 1
        nb = iq.shape[1] // self.block_bytes if n_blocks is None else n_blocks
        st = torch.cuda.current_stream(iq.device) if stream is None else stream
        self._keep = (getattr(self, "_keep", ()) + (iq,))[-FIFO_DEPTH:]  # inputs stay alive while their submit may be in flight
        _check(self.L, self.L.tfrec_amd_submit_device(self.h, C.c_void_p(iq.data_ptr()), iq.stride(0), nb,
                                                      C.c_void_p(st.cuda_stream)))
        return nb

    def sync(self):
        _check(self.L, self.L.tfrec_amd_sync(self.h))

    def drain(self, allow_overflow: bool = False) -> np.ndarray:
        out = np.empty(self.max_events, dtype=EVENT_DTYPE)
        n = C.c_int(0)
        rc = self.L.tfrec_amd_drain_events(self.h, out.ctypes.data, self.max_events, C.byref(n))
        _check(self.L, rc, ok=(E_OK, E_OVERFLOW) if allow_overflow else (E_OK,))
        return out[: n.value]

    def stage0(self, stream: int, n_pairs: int) -> np.ndarray:
        """input_10x: the 1.536 MS/s int16 IQ the 10:1 stage produced for the last submit."""
        out = np.empty(2 * n_pairs, dtype=np.int16)
        _check(self.L, self.L.tfrec_amd_read_stage0(self.h, stream, out.ctypes.data, n_pairs))
        return out

    def decimated(self, stream: int, n_pairs: int) -> np.ndarray:
        out = np.empty(2 * n_pairs, dtype=np.int16)
        _check(self.L, self.L.tfrec_amd_read_decimated(self.h, stream, out.ctypes.data, n_pairs))
        return out

    def atan_uncertain(self) -> int:
        n = C.c_uint64(0)
        _check(self.L, self.L.tfrec_amd_atan_uncertain(self.h, C.byref(n)))
        return int(n.value)

    def fm_stats(self) -> dict:
        """fm_dev samples decided by the exact slow path / checked against the host's libm at drain / differing from it /
        too close to an atan2 rounding midpoint for glibc's error bound (tfrec_amd_get_fm_stats)."""
        st = FmStats()
        _check(self.L, self.L.tfrec_amd_get_fm_stats(self.h, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in FmStats._fields_[:4]}

    def thresh(self, stream: int) -> int:
        v = C.c_int(0)
        _check(self.L, self.L.tfrec_amd_read_thresh(self.h, stream, C.byref(v)))
        return int(v.value)

    def timings(self) -> dict:
        t = Timings()
        _check(self.L, self.L.tfrec_amd_get_timings(self.h, C.byref(t)))
        return {n: float(getattr(t, n)) for n, _ in Timings._fields_}

    def layout(self) -> int:
        """Internal HIP streams of the pipeline: 6 deep, 4 shallow, 2 serial cross-check (tfrec_amd_get_layout)."""
        n = C.c_int(0)
        _check(self.L, self.L.tfrec_amd_get_layout(self.h, C.byref(n)))
        return int(n.value)

    def memory(self) -> dict:
        """Bytes of device memory / page-locked host memory the context holds (the caller's input batches not counted)."""
        d, h = C.c_uint64(), C.c_uint64()
        _check(self.L, self.L.tfrec_amd_get_memory(self.h, C.byref(d), C.byref(h)))
        return {"device_bytes": int(d.value), "pinned_host_bytes": int(h.value)}

    def stats(self) -> dict:
        """Counters of the speculative stages (how the work was done; results never depend on them)."""
        st = Stats()
        _check(self.L, self.L.tfrec_amd_get_stats(self.h, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in Stats._fields_}


def fm_dev_nrzs_probe(records: np.ndarray, device: int = 0) -> np.ndarray:
    """The device's fm_dev_nrzs (dsp_stuff.cpp:269-279) on int32 quadruples [n, 4] = (ar, aj, br, bj) -> int32[n]."""
    L = load_library()
    q = np.ascontiguousarray(records, dtype=np.int32).reshape(-1, 4)
    out = np.empty(len(q), dtype=np.int32)
    _check(L, L.tfrec_amd_fm_dev_probe(device, 2, q.ctypes.data, len(q), out.ctypes.data, None))
    return out


def iir_probe(cutoff: float, x: np.ndarray, form: int = 1, device: int = 0) -> np.ndarray:
    """The device's iir2 (dsp_stuff.cpp:28-56) with set(cutoff) over the doubles x from the zero state -> float64[n].
    form 0: iir_step (the reference's association), form 1: iir_step_t (the form the kernels run)."""
    L = load_library()
    xin = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(len(xin), dtype=np.float64)
    _check(L, L.tfrec_amd_iir_probe(device, float(cutoff), int(form), xin.ctypes.data, len(xin), out.ctypes.data))
    return out


def fm_dev_probe(records: np.ndarray, device: int = 0, cross: bool = False):
    """The device's fm_dev (dsp_stuff.cpp:284-292) on int32 quadruples [n, 4] = (ar, aj, br, bj), or (cross=True) on
    int64 cross terms [n, 2] = (cr, cj) -> (int32[n], stats)."""
    L = load_library()
    q = (np.ascontiguousarray(records, dtype=np.int64).reshape(-1, 2) if cross
         else np.ascontiguousarray(records, dtype=np.int32).reshape(-1, 4))
    out = np.empty(len(q), dtype=np.int32)
    st = FmStats()
    _check(L, L.tfrec_amd_fm_dev_probe(device, 1 if cross else 0, q.ctypes.data, len(q), out.ctypes.data, C.byref(st)))
    return out, {n: int(getattr(st, n)) for n, _ in FmStats._fields_[:4]}


def events_canon(events: np.ndarray):
    """Vector form of event_tuples_full for whole batches: (stream[n], int64 matrix [n, 5 + 64 + 2] with the columns slot,
    end_sample, byte_cnt, rssi_db, offset, rdata, rssi_raw, status) -- the layout of oracle.canon()."""
    L = load_library()
    events = events[events["status"] != STATUS_BITS]
    m = np.empty((len(events), 71), dtype=np.int64)
    m[:, 0] = events["slot"]
    m[:, 1] = events["end_sample"]
    m[:, 2] = events["byte_cnt"]
    slots = events["slot"].tolist()
    raws = events["rssi_raw"].tolist()
    m[:, 3] = [L.tfrec_amd_rssi_db(sl, rw) for sl, rw in zip(slots, raws)]
    m[:, 4] = events["offset"]
    m[:, 5:69] = events["rdata"]
    m[:, 69] = events["rssi_raw"]
    m[:, 70] = events["status"]
    return events["stream"].astype(np.int64), m


def bits_by_flush(events: np.ndarray, stream: int):
    """TFREC_AMD_F_BITS: {(slot, seq): "0110..."} -- the bits handed to decoder::store_bit before flush number seq of the
    slot, from the BITS chunks of `events` (drain order; concatenate the drains of consecutive submits first)."""
    out = {}
    for e in events:
        if int(e["stream"]) != stream or int(e["status"]) != STATUS_BITS:
            continue
        n = int(e["byte_cnt"])
        bits = np.unpackbits(e["rdata"], bitorder="little")[:n]
        key = (int(e["slot"]), int(e["seq"]))
        out[key] = out.get(key, "") + "".join("01"[b] for b in bits)
    return out


def event_tuples_full(events: np.ndarray, stream: int | None = None):
    """event_tuples + (rssi_raw, status): the raw RSSI accumulator itself (tfa1.cpp:161, tfa2.cpp:373, whb.cpp:678) and the
    verdict of the decoder's acceptance tests computed on the GPU -- what oracle.Oracle.events_full() returns."""
    L = load_library()
    out = []
    for e in events:
        if (stream is not None and int(e["stream"]) != stream) or int(e["status"]) == STATUS_BITS:
            continue
        out.append((int(e["slot"]), int(e["end_sample"]), int(e["byte_cnt"]),
                    int(L.tfrec_amd_rssi_db(int(e["slot"]), int(e["rssi_raw"]))), int(e["offset"]),
                    bytes(e["rdata"]), int(e["rssi_raw"]), int(e["status"])))
    return out


def event_tuples(events: np.ndarray, stream: int | None = None):
    """Canonical comparable form (slot, end_sample, byte_cnt, rssi_db, offset, rdata) of flush events,
    in per-(stream, slot) order -- the same tuple the oracle and the reference harness produce."""
    L = load_library()
    out = []
    for e in events:
        if (stream is not None and int(e["stream"]) != stream) or int(e["status"]) == STATUS_BITS:
            continue
        out.append((int(e["slot"]), int(e["end_sample"]), int(e["byte_cnt"]),
                    int(L.tfrec_amd_rssi_db(int(e["slot"]), int(e["rssi_raw"]))), int(e["offset"]),
                    bytes(e["rdata"])))
    return out
