"""What a two-literal table entry (missing 4 / next-round 1a) could save on the bench corpus: the share of codes
that are literals followed by another literal whose two code lengths fit the 10-bit direct table, i.e. the loop trips a packed
entry would remove (CPU only: a plain bit-serial inflate of the corpus's dynamic blocks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swcompression_amd import corpus

LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DEXT = [0, 0, 0, 0] + [e for e in range(1, 14) for _ in (0, 1)]


def canon(lens):
    maxl = max(lens) if lens else 0
    cnt = [0] * (maxl + 2)
    for l in lens:
        if l:
            cnt[l] += 1
    code, nxt = 0, [0] * (maxl + 2)
    for l in range(1, maxl + 1):
        code = (code + cnt[l - 1]) << 1
        nxt[l] = code
    t = {}
    for s, l in enumerate(lens):
        if l:
            t[(l, nxt[l])] = s
            nxt[l] += 1
    return t


def stats(z):
    pos = 0

    def bits(n):
        nonlocal pos
        v = 0
        for i in range(n):
            v |= ((z[pos >> 3] >> (pos & 7)) & 1) << i
            pos += 1
        return v

    def dec(t):
        code = l = 0
        while True:
            code = (code << 1) | bits(1)
            l += 1
            if (l, code) in t:
                return t[(l, code)], l
    codes = pairs10 = pairs11 = lits = 0
    while True:
        last, bt = bits(1), bits(2)
        assert bt == 2
        hlit, hdist, hclen = bits(5) + 257, bits(5) + 1, bits(4) + 4
        cl = [0] * 19
        for i in range(hclen):
            cl[[16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15][i]] = bits(3)
        t = canon(cl)
        lens = []
        while len(lens) < hlit + hdist:
            s, _ = dec(t)
            if s < 16:
                lens.append(s)
            elif s == 16:
                lens += [lens[-1]] * (3 + bits(2))
            elif s == 17:
                lens += [0] * (3 + bits(3))
            else:
                lens += [0] * (11 + bits(7))
        tl, td = canon(lens[:hlit]), canon(lens[hlit:hlit + hdist])
        prev_lit_len = 0   # code length of the previous symbol if it was an UNPAIRED literal
        while True:
            s, l = dec(tl)
            codes += 1
            if s < 256:
                lits += 1
                if prev_lit_len and prev_lit_len + l <= 10:
                    pairs10 += 1; prev_lit_len = 0
                elif prev_lit_len and prev_lit_len + l <= 11:
                    pairs11 += 1; prev_lit_len = 0
                else:
                    prev_lit_len = l
                continue
            prev_lit_len = 0
            if s == 256:
                break
            bits(LEXT[s - 257])
            d, _ = dec(td)
            codes += 1
            bits(DEXT[d])
        if last:
            break
    return codes, lits, pairs10, pairs11


n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
units, _ = corpus.build_units("deflate", n, 65536, seed=2)
tot = [0, 0, 0, 0]
for z in units:
    for i, v in enumerate(stats(z)):
        tot[i] += v
print("%d streams: %.0f codes per stream, %.1f %% of them literals; literal pairs that fit a 10-bit index: %.2f %% of the codes "
      "(an 11-bit table: %.2f %% more)" % (n, tot[0] / n, 100 * tot[1] / tot[0], 100 * tot[2] / tot[0], 100 * tot[3] / tot[0]))
