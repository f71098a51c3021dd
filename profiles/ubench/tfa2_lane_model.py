#!/usr/bin/env python3
"""CPU model of the lane-per-step TFA_2-family slicer with frozen thresholds (chains2.hip: coop_tfa2, round 5): every lane walks
its step's candidates from a SPECULATED start state (last_bit = polarity of the nearest sample above / below the thresholds
before it, last_bit_idx = the nearest alternation before it), then the start states are compared with what the lane before
really left behind and the lanes that were wrong walk again -- against the sample-by-sample rules of tfa2.cpp:383-411 +
decoder.cpp:118-122."""
import random, sys

KB = 8192
SPAN = 2 * KB
SPB = 22.27
TD_LO = int(SPB / 4) + 1
TD_HI = int(32 * SPB + 0.999999) - 1


def numbits(d):
    return int(((d // 2) + SPB / 2) / SPB)


def reference(cls, og, last_bit, lbi, blk):
    """cls[i] in (0 neutral, 1 above hi, 2 below lo) for sample og + i.  Returns (bits, last_bit, lbi rel. to the last sample's block, bitcnt)."""
    bits, bitcnt = [], 0
    cur = blk
    for i, c in enumerate(cls):
        g = og + i
        while cur < (g >> 13):
            if lbi:
                lbi -= SPAN
            cur += 1
        if c == 0:
            continue
        bit = 1 if c == 1 else 0
        if bit == last_bit:
            continue
        index = 2 * (g & (KB - 1))
        if index > lbi + 8:
            bitcnt += 1
            td = index - lbi
            if TD_LO <= td <= TD_HI:
                nb = numbits(td)
                if nb < 32:
                    bits += [last_bit] * (nb - 1)
                bits.append(bit)
                last_bit = bit
        if index - lbi > 2:
            lbi = index
    last = og + len(cls) - 1
    while cur < (last >> 13):
        if lbi:
            lbi -= SPAN
        cur += 1
    return bits, last_bit, lbi, bitcnt


def ctz(x):
    return (x & -x).bit_length() - 1


M64 = 2**64 - 1


def walk(m1, m0, Ibase, last_bit, lbi):
    """One lane: the walk of walk_one_block in absolute index units.  Returns (bits, last_bit, lbi, bitcnt, bad)."""
    bits, bitcnt, bad = [], 0, False
    todo = M64
    acc_lo = max(TD_LO, 9)
    while True:
        m = (m0 if last_bit else m1) & todo
        if not m:
            break
        k = ctz(m)
        todo = (~1 << k) & M64
        index = Ibase + 2 * k
        d = index - lbi
        if d > 2:
            lbi = index
        if d > 8:
            bitcnt += 1
        if acc_lo <= d <= TD_HI:
            nb = numbits(d)
            run = nb - 1 if (nb < 32 and nb > 1) else 0
            bits += [last_bit] * run + [last_bit ^ 1]
            if len(bits) > 64:
                bad = True
            last_bit ^= 1
            continue
        rest = (m >> 1) >> k
        R = ctz(~rest)
        if R > 0:
            e = index + 2 - lbi
            t_set = 1 if e > 2 else ((2 - e) >> 1) + 2
            if t_set <= R:
                lbi = index + 2 * (t_set + 2 * ((R - t_set) >> 1))
            todo = (~1 << (k + R)) & M64
    return bits, last_bit, lbi, bitcnt, bad


def model(cls, og, last_bit, lbi, blk, stats):
    n = len(cls)
    last = og + n - 1
    bits, bitcnt = [], 0
    cur = blk
    pos = 0
    while pos < n:
        gs = og + pos
        # bring lbi to the block of the group's first sample
        b0 = gs >> 13
        if b0 != cur:
            if lbi:
                lbi -= SPAN * (b0 - cur)
            cur = b0
        ng = min(64, (n - pos + 63) // 64)
        done = False
        if lbi != 0:
            Labs = lbi + SPAN * cur
            H, Lw = [], []
            for l in range(64):
                h = w = 0
                for k in range(64):
                    i = pos + 64 * l + k
                    if i < n:
                        if cls[i] == 1:
                            h |= 1 << k
                        elif cls[i] == 2:
                            w |= 1 << k
                H.append(h)
                Lw.append(w)
            # speculated start states
            start = []
            sb, sl = last_bit, Labs
            for l in range(64):
                start.append((sb, sl))
                anym = H[l] | Lw[l]
                # alternation edges of the lane under "every one accepted"
                c = sb
                for k in range(64):
                    if (anym >> k) & 1:
                        p = (H[l] >> k) & 1
                        if p != c:
                            sl = 2 * (gs + 64 * l + k)
                            c = p
                sb = c
            res = [None] * 64
            dirty = [True] * 64
            it = 0
            bad = False
            while True:
                it += 1
                for l in range(64):
                    if dirty[l]:
                        res[l] = walk(H[l], Lw[l], 2 * (gs + 64 * l), start[l][0], start[l][1])
                dirty = [False] * 64
                prev = (last_bit, Labs)
                mism = False
                for l in range(64):
                    if start[l] != prev:
                        start[l] = prev
                        dirty[l] = True
                        mism = True
                    prev = (res[l][1], res[l][2])
                if not mism:
                    break
                if it > 8:
                    bad = True
                    break
            stats[2] += it
            if not bad and not any(r[4] for r in res):
                for l in range(64):
                    bits += res[l][0]
                    bitcnt += res[l][3]
                last_bit, Labs = res[63][1], res[63][2]
                gend = min(gs + 64 * ng - 1, last)
                cur = gend >> 13
                lbi = Labs - SPAN * cur
                done = True
                stats[0] += 1
        if not done:
            stats[1] += 1
            seg = cls[pos:pos + 64 * ng]
            b, last_bit, lbi, bc = reference(seg, gs, last_bit, lbi, cur)
            cur = (gs + len(seg) - 1) >> 13
            bits += b
            bitcnt += bc
        pos += 64 * ng
    return bits, last_bit, lbi, bitcnt


def main():
    rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    stats = [0, 0, 0]
    for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 100):
        n = rnd.randrange(4096, 30000)
        og = rnd.randrange(0, 3 * KB)
        style = rnd.randrange(4)
        cls = []
        while len(cls) < n:
            if style == 0:  # clean NRZ: runs of whole bits, a neutral sample at some edges
                nb = rnd.choice([1, 1, 1, 2, 2, 3, 5, 8, 33, 40])
                pol = 1 if (not cls or cls[-1] != 1) else 2
                cls += [0] * rnd.choice([0, 0, 1, 2]) + [pol] * max(1, int(nb * SPB) - rnd.randrange(0, 3))
            elif style == 1:  # the same with glitches
                nb = rnd.choice([1, 1, 2, 3])
                pol = 1 if (not cls or cls[-1] != 1) else 2
                seg = [pol] * int(nb * SPB)
                for _ in range(rnd.choice([0, 0, 1, 2])):
                    k = rnd.randrange(len(seg))
                    for q in range(k, min(len(seg), k + rnd.choice([1, 1, 2, 3, 5]))):
                        seg[q] = rnd.choice([0, 3 - pol])
                cls += seg
            elif style == 2:  # noise
                p = rnd.choice([0.1, 0.5, 0.9])
                cls += [rnd.choice([1, 2]) if rnd.random() < p else 0 for _ in range(rnd.randrange(1, 300))]
            else:  # bursts and silences
                if rnd.random() < 0.5:
                    cls += [0] * rnd.randrange(1, 2000)
                else:
                    for _ in range(rnd.randrange(1, 60)):
                        pol = 1 if (not cls or cls[-1] != 1) else 2
                        cls += [pol] * rnd.choice([20, 21, 22, 23, 44, 45, 67])
        cls = cls[:n]
        last_bit = rnd.randrange(2)
        blk = og >> 13
        lbi = rnd.choice([0, 2 * (og & (KB - 1)) - rnd.choice([2, 4, 10, 30, 100, 1000, 20000]), 2 * (og & (KB - 1))])
        a = reference(cls, og, last_bit, lbi, blk)
        b = model(cls, og, last_bit, lbi, blk, stats)
        if a != b:
            print("MISMATCH trial", trial, "style", style, "n", n, "og", og, "lbi", lbi, len(a[0]), len(b[0]), a[1:], b[1:])
            return 1
    print("ok: groups vector %d, scalar %d, walks per vector group %.2f" % (stats[0], stats[1], stats[2] / max(1, stats[0] + stats[1])))
    return 0


sys.exit(main())
