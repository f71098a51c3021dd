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
