#!/usr/bin/env python3
"""CPU model of the lane-per-step TFA_1 slicer (chains2.hip: coop_tfa1, round 5) against the sample-by-sample rules of
tfa1.cpp:164-177 + decoder.cpp:118-122 on random candidate strings: emitted bits and the last_bit_idx left behind."""
import random, sys

KB = 8192
SPAN = 2 * KB


def reference(cand, og, lbi0, blk0):
    """cand[i] = candidate at sample og + i; lbi0 relative to block blk0.  Returns (bits, lbi relative to the last sample's block)."""
    bits = []
    lbi, cur = lbi0, blk0
    for i, cbit in enumerate(cand):
        g = og + i
        b = g >> 13
        while cur < b:  # demodulator::start at every block
            if lbi:
                lbi -= SPAN
            cur += 1
        if not cbit:
            continue
        index = 2 * (g & (KB - 1))
        if lbi:
            if index - lbi > 4:
                n = 22
                while n <= index - lbi:
                    bits.append(1)
                    n += 20
                bits.append(0)
        if index - lbi > 2:
            lbi = index
    last = og + len(cand) - 1
    while cur < (last >> 13):
        if lbi:
            lbi -= SPAN
        cur += 1
    return bits, lbi


def ctz(x):
    return (x & -x).bit_length() - 1


def model(cand, og, lbi0, blk0, stats):
    n = len(cand)
    last = og + n - 1
    nsteps = ((last - og) >> 6) + 1
    bits = []
    lbi, cur = lbi0, blk0
    for sb in range(0, nsteps, 64):
        ng = min(64, nsteps - sb)
        words = []
        for l in range(64):
            w = 0
            for k in range(64):
                i = 64 * (sb + l) + k
                if i < n and cand[i]:
                    w |= 1 << k
            words.append(w)
        have = lbi != 0
        Labs = lbi + SPAN * cur if have else SPAN * ((og + 64 * sb) >> 13)
        hz = False
        for l in range(64):
            gb = og + 64 * (sb + l)
            rel = gb & (KB - 1)
            d0 = (KB - rel) & (KB - 1)
            d1 = (KB + 1 - rel) & (KB - 1)
            w = words[l]
            if not have and d1 < 64 and (w >> d1) & 1:
                hz = True
        ok = not hz
        if ok:
            g0 = 2 * (og + 64 * sb) - Labs
            cont0 = have and g0 <= 4 and (words[0] & 1)
            pm = fm = 0
            for l in range(64):
                nw = ~words[l] & (2**64 - 1)
                q = 64 - nw.bit_length()
                if q < 64 and (q & 1):
                    pm |= 1 << l
                if q == 64:
                    fm |= 1 << l
            cin = (((pm | fm) + pm + (1 if (cont0 and g0 == 2) else 0)) & (2**64 - 1)) ^ fm
            Lc = [0] * 64
            acc = [[] for _ in range(64)]
            defer = [False] * 64
            I0f = [0] * 64
            longs = [0] * 64
            bad = False
            for l in range(64):
                w = words[l]
                Ibase = 2 * (og + 64 * (sb + l))
                cont_l = cont0 if l == 0 else bool((w & 1) and (words[l - 1] >> 63))
                L = Ibase - (2 if (cin >> l) & 1 else 4)
                first = True
                while w:
                    k0 = ctz(w)
                    inv = ~(w >> k0) & (2**64 - 1)
                    ln = ctz(inv) if inv else 64 - k0
                    if k0 + ln < 64:
                        ln = ctz(~(w >> k0))
                    w = 0 if k0 + ln >= 64 else w & (~0 << (k0 + ln))
                    I0 = Ibase + 2 * k0
                    if first and not cont_l:
                        defer[l] = True
                        I0f[l] = I0
                        L = I0
                    else:
                        gap = I0 - L
                        if gap > 4 and (L & (SPAN - 1)) != 0:
                            ones = (gap - 22) // 20 + 1 if gap >= 22 else 0
                            if ones >= 32 or len(acc[l]) + ones + 1 > 64:
                                bad = True
                            else:
                                acc[l] += [1] * ones + [0]
                        if gap > 2:
                            L = I0
                    first = False
                    if ln > 1:
                        d = I0 - L
                        t1 = 1 if d >= 2 else 2
                        if t1 <= ln - 1:
                            L = I0 + 2 * t1 + 4 * ((ln - 1 - t1) >> 1)
                Lc[l] = L
            for l in range(64):
                below = [m for m in range(l) if words[m]]
                Lprev = Lc[below[-1]] if below else Labs
                if defer[l] and (Lprev & (SPAN - 1)) != 0:
                    gap = I0f[l] - Lprev
                    if gap <= 2:
                        bad = True
                    if gap > 4:
                        ones = (gap - 22) // 20 + 1 if gap >= 22 else 0
                        if ones >= 32:
                            longs[l] = ones
                        elif len(acc[l]) + ones + 1 > 64:
                            bad = True
                        else:
                            acc[l] = [1] * ones + [0] + acc[l]
            if not bad:
                for l in range(64):
                    if longs[l]:
                        bits += [1] * longs[l] + [0]
                    bits += acc[l]
                gend = min(og + 64 * (sb + ng) - 1, last)
                nb = gend >> 13
                ne = [l for l in range(64) if words[l]]
                Lnew = Lc[ne[-1]] if ne else Labs
                lbi = Lnew - SPAN * nb if (Lnew & (SPAN - 1)) else 0
                cur = nb
                stats[0] += 1
                continue
        stats[1] += 1
        # scalar walk of the group (the reference's rules sample by sample, block-relative)
        lo, hi = 64 * sb, min(n, 64 * (sb + ng))
        for i in range(lo, hi):
            g = og + i
            b = g >> 13
            if not cand[i]:
                continue
            if b != cur:
                if lbi:
                    lbi -= SPAN * (b - cur)
                cur = b
            index = 2 * (g & (KB - 1))
            if lbi and index - lbi > 4:
                nn = 22
                while nn <= index - lbi:
                    bits.append(1)
                    nn += 20
                bits.append(0)
            if index - lbi > 2:
                lbi = index
    bl = last >> 13
    if bl != cur:
        if lbi:
            lbi -= SPAN * (bl - cur)
    return bits, lbi


def main():
    rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    stats = [0, 0]
    for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
        n = rnd.randrange(4096, 40000)
        og = rnd.randrange(0, 3 * KB)
        if rnd.random() < 0.3:
            og = rnd.choice([0, KB - 1, KB, KB + 1, KB - 64, KB - 63])
        style = rnd.randrange(5)
        cand = []
        while len(cand) < n:
            if style == 0:
                p = rnd.choice([0.02, 0.1, 0.3, 0.5, 0.8])
                cand += [1 if rnd.random() < p else 0 for _ in range(rnd.randrange(1, 400))]
            elif style == 1:  # bursts: a zero pulse every ~10 samples, runs of 1-3
                cand += [0] * rnd.choice([8, 9, 10, 18, 19, 28, 40]) + [1] * rnd.choice([1, 1, 2, 3])
            elif style == 2:  # long runs
                cand += [1] * rnd.randrange(1, 300) + [0] * rnd.randrange(1, 5)
            elif style == 3:  # long silences
                cand += [0] * rnd.randrange(1, 1500) + [1] * rnd.randrange(1, 4)
            else:
                cand += [rnd.randrange(2) for _ in range(rnd.randrange(1, 100))] + [0] * rnd.randrange(0, 200)
        cand = cand[:n]
        if rnd.random() < 0.5:
            lbi0, blk0 = 0, og >> 13
        else:  # a continued window: last_bit_idx from the submit before, relative to block -1 -> rebased to the window's block
            lbi0 = rnd.choice([SPAN - 2, SPAN - 4, SPAN - 6, SPAN - 40, SPAN - 600, 2, 4])
            og = rnd.choice([0, 0, 0, 1, 2])
            lbi0 = lbi0 - SPAN * ((og >> 13) + 1)
            blk0 = og >> 13
        a = reference(cand, og, lbi0, blk0)
        b = model(cand, og, lbi0, blk0, stats)
        if a != b:
            print("MISMATCH trial", trial, "n", n, "og", og, "lbi0", lbi0, "style", style, len(a[0]), len(b[0]), a[1], b[1])
            for i, (x, y) in enumerate(zip(a[0], b[0])):
                if x != y:
                    print("first differing bit", i)
                    break
            return 1
    print("ok: groups vector %d, scalar %d" % tuple(stats))
    return 0


sys.exit(main())
