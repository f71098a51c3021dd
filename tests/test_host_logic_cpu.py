"""Host-side arithmetic the C ABI glue relies on, restated and checked on the CPU."""


def test_tfa2_numbits_multiplier_reproduces_the_fp64_expression():
    """tfa2.cpp:397 numbits = (int)(((tdiff / 2) + spb / 2) / spb); the kernels evaluate it as
    (h * A + 2^39) >> 40 with A chosen and checked exhaustively in tfrec_amd_create (capi.hip).  The same search must
    succeed for the three samples-per-bit values the reference registers (main.cpp:186-217)."""
    for baud in (17240, 9600, 8842):
        spb = (1536000 / 4.0) / baud
        hmax = int(16 * spb) + 2
        a0 = int(float(1 << 40) / spb)
        found = None
        for a in (a0, a0 + 1, a0 - 1):
            if all(((h * a + (1 << 39)) >> 40) == int((float(h) + spb / 2) / spb) for h in range(hmax + 1)):
                found = a
                break
        assert found is not None, baud


def test_python_constants_match_the_c_header():
    """The ctypes binding restates a few constants of include/tfrec_amd.h; they must not drift."""
    import os
    import re

    from tfrec_amd import api

    hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "tfrec_amd.h")).read()

    def define(name):
        m = re.search(r"#define\s+%s\s+([0-9xa-fA-Fu]+)" % name, hdr)
        assert m, name
        return int(m.group(1).rstrip("u"), 0)

    assert define("TFREC_AMD_FIFO_DEPTH") == api.FIFO_DEPTH
    assert define("TFREC_AMD_BLOCK_BYTES") == api.BLOCK_BYTES
    assert define("TFREC_AMD_BLOCK_DEC") == api.BLOCK_DEC
    assert define("TFREC_AMD_NSLOTS") == api.NSLOTS
    for flag in ("ALL_FLUSHES", "TIMING", "SERIAL_CHAINS", "INPUT_10X"):
        assert define("TFREC_AMD_F_" + flag) == getattr(api, "F_" + flag)
