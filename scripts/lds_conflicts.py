#!/usr/bin/env python
"""ds_read_b128 bank-conflict count of the conv kernels' pixel-fragment reads (MI355X_MICROARCH.md LDS table:
a wave64 ds_read_b128 is served in 4 groups of 16 lanes; within a group every distinct 16-byte slot
(address/16 mod 16) costs one cycle unless the addresses are identical)."""
import sys
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]

def cycles(PSB, HC, TW, S):
    tot = 0
    for g in GROUPS:
        slots = {}
        for li in g:
            row, col = li // TW, li % TW
            a = ((row * S) * HC + col * S) * PSB
            slots.setdefault((a // 16) % 16, set()).add(a)
        tot += max(len(v) for v in slots.values())
    return tot          # 2 = conflict-free (per half-wave; the lh=1 half behaves identically)

if __name__ == '__main__':
    for NP in (2, 3):
        for CK in (16, 32):
            for S in (1, 2):
                for TW in (8, 16, 32):
                    HC = (TW - 1) * S + 3
                    base = NP * CK * 2
                    res = [(pad, cycles(base + pad, HC, TW, S)) for pad in range(0, 129, 16)]
                    best = min(res, key=lambda r: (r[1], r[0]))
                    print('NP %d CK %2d S %d TW %2d HC %2d: base %3d  ' % (NP, CK, S, TW, HC, base) +
                          ' '.join('%d:%d' % r for r in res) + '   best pad %d' % best[0])


def cycles2(PSB, ROWB, TW, S):
    tot = 0
    for g in GROUPS:
        slots = {}
        for li in g:
            row, col = li // TW, li % TW
            a = (row * S) * ROWB + col * S * PSB
            slots.setdefault((a // 16) % 16, set()).add(a)
        tot += max(len(v) for v in slots.values())
    return tot


def search():
    for NP in (2, 3):
        for CK in (16, 32):
            for S in (1, 2):
                for TW in (8, 16, 32):
                    HC = (TW - 1) * S + 3
                    base = NP * CK * 2
                    best = None
                    for pad in range(0, 65, 16):
                        for rpad in range(0, 257, 16):
                            c = cycles2(base + pad, HC * (base + pad) + rpad, TW, S)
                            key = (c, pad * HC + rpad)
                            if best is None or key < best[0]:
                                best = (key, pad, rpad)
                    print('NP %d CK %2d S %d TW %2d: cycles %d pixel pad %d row pad %d' % (NP, CK, S, TW, best[0][0], best[1], best[2]))

if __name__ == '__main__' and len(sys.argv) > 1:
    search()
