#!/usr/bin/env python
"""CPU emulation of Winograd F(2x2,3x3) with the f16x2 three-product multiply, over the whole ROMP HRNet-32 network
(VERDICT r04 "next round" item 2, step A: the precision gate BEFORE any kernel is written).

Every 3x3 stride-1 conv with >= 32 input channels (the BasicBlock convs of the HRNet branches, model.py:54-83: 73.7 of the
85.7 GFLOP of a forward) is computed as the kernel would:

  x (H2: x * 2^4 = h1 + h2, 22 bits)  --f32-->  V = B^T d B per 4x4 tile (f32 adds, what the VALU does on the LDS halo tile)
  V * 2^vs split again into two fp16 pieces (in-kernel re-split; vs = per-tensor power of two, or the fixed scale given)
  U = G g G^T in f64 on the host, * 2^ws, split into two fp16 pieces (pack time)
  M_p = sum_cin U_p V_p for the 16 positions p with the products h1h1 + h1h2 + h2h1, f32 accumulation (MFMA)
  Y = A^T M A (f32 adds in the epilogue), / 2^(vs+ws)

every other conv stays on the product path's f16x2 three-product direct arithmetic.  Reported: maps max-abs against the
plain-f32 oracle (what the parity gate compares with; gate 1e-4, wanted <= 2e-5) and against an f64 run.

    python scripts/winograd_emul.py [B] [mode ...]      modes: direct wino_f32 wino_f16x2 wino_f16x2@<fixed input scale>
"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from oracle import romp_oracle as O  # noqa: E402
from precision_emul import split, pow2_scale, MODES  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def h2_round(x, shift=4):
    """What an H2 tensor holds of x: fp16(x*2^s) + fp16(rest), back in f32 (exact sum)."""
    s = 2.0 ** shift
    p = split(x * s, torch.float16, 2)
    return (p[0] + p[1]) / s


def input_transform(x):
    """x (B,C,H,W) f32, H and W even -> V (16, C, B*T) with f32 adds in the order a kernel would do them (rows, then columns)."""
    Bn, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                 # B, C, H/2, W/2, 4, 4
    d = t.permute(4, 5, 1, 0, 2, 3).reshape(4, 4, C, -1)   # r, c, C, B*T
    # rows: B^T d
    r0, r1, r2, r3 = d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]
    rows = torch.stack([r0, r1, r2, r3])                   # 4(r'), 4(c), C, N
    c0, c1, c2, c3 = rows[:, 0] - rows[:, 2], rows[:, 1] + rows[:, 2], rows[:, 2] - rows[:, 1], rows[:, 1] - rows[:, 3]
    V = torch.stack([c0, c1, c2, c3], 1)                   # 4, 4, C, N
    return V.reshape(16, C, -1)


def output_transform(M, Bn, H, W):
    """M (16, Cout, B*T) -> y (B, Cout, H, W): A^T M A with f32 adds."""
    Co = M.shape[1]
    m = M.reshape(4, 4, Co, -1)
    r0 = m[0] + m[1] + m[2]
    r1 = m[1] - m[2] - m[3]
    rows = torch.stack([r0, r1])                           # 2, 4, Co, N
    y0 = rows[:, 0] + rows[:, 1] + rows[:, 2]
    y1 = rows[:, 1] - rows[:, 2] - rows[:, 3]
    y = torch.stack([y0, y1], 1)                           # 2(r), 2(c), Co, N
    y = y.reshape(2, 2, Co, Bn, H // 2, W // 2).permute(3, 2, 4, 0, 5, 1).reshape(Bn, Co, H, W)
    return y


def make_conv(mode, fixed=None, stats=None):
    direct_prods = MODES['f16x2_3'][2]

    def direct(x, w, stride):
        sw = pow2_scale(w, 256.0)
        sx = 16.0
        xp, wp = split(x * sx, torch.float16, 2), split(w * sw, torch.float16, 2)
        y = None
        for (i, j) in sorted(direct_prods, key=lambda p: -(p[0] + p[1])):
            t = F.conv2d(xp[i], wp[j], None, stride=stride, padding=w.shape[-1] // 2)
            y = t if y is None else y + t
        return y / (sx * sw)

    def conv(x, sd, name, stride=1):
        w = sd[name + '.weight']
        b = sd.get(name + '.bias')
        wino = mode != 'direct' and w.shape[-1] == 3 and stride == 1 and w.shape[1] >= 32 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
        fl = float(w.numel()) * x.shape[2] * x.shape[3] / (stride * stride)
        stats['flops'] += fl
        if not wino:
            y = direct(x, w, stride)
        else:
            stats['hit'] += 1
            stats['flops_hit'] += fl
            Bn, C, H, W = x.shape
            xr = h2_round(x)                                # the H2 tensor the kernel reads
            V = input_transform(xr)
            U = torch.einsum('ij,ocjk,lk->iloc', G, w.double(), G).reshape(16, w.shape[0], C)      # f64 on the host
            stats['max_V'] = max(stats['max_V'], float(V.abs().max()))
            stats['max_x'] = max(stats['max_x'], float(x.abs().max()))
            if mode == 'wino_f32':
                M = torch.bmm(U.float(), V)
            else:
                sv = fixed if fixed else pow2_scale(V, 1024.0)
                su = pow2_scale(U, 256.0)
                Vp = split(V * sv, torch.float16, 2)
                Up = split((U * su).float(), torch.float16, 2)
                M = None
                for (i, j) in sorted(direct_prods, key=lambda p: -(p[0] + p[1])):
                    t = torch.bmm(Up[j], Vp[i])
                    M = t if M is None else M + t
                M = M / (sv * su)
            y = output_transform(M, Bn, H, W)
        if b is not None:
            y = y + b.view(1, -1, 1, 1)
        return y
    return conv


def main():
    args = sys.argv[1:]
    B = int(args[0]) if args and args[0].isdigit() else 1
    modes = [a for a in args if not a.isdigit()] or ['direct', 'wino_f32', 'wino_f16x2', 'wino_f16x2@4']
    torch.set_num_threads(os.cpu_count())
    for seed in (0, 1):
        sd = O.make_romp_state_dict(seed)
        img = O.make_images(B, seed=1 + seed)
        cm32, pm32 = O.romp_net_forward(sd, img)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        orig_conv, orig_coord = O._conv, O.coord_maps
        O.coord_maps = lambda size=128: orig_coord(size).double()
        cm64, pm64 = O.romp_net_forward(sd64, img.double())
        O.coord_maps = orig_coord
        print('weights seed %d: f32 oracle vs f64: center %.3e params %.3e' % (seed, float((cm32 - cm64).abs().max()), float((pm32 - pm64).abs().max())))
        for mode in modes:
            fixed = None
            m = mode
            if '@' in mode:
                m, fixed = mode.split('@')
                fixed = float(fixed)
            stats = dict(flops=0.0, flops_hit=0.0, hit=0, max_V=0.0, max_x=0.0)
            O._conv = make_conv(m, fixed, stats)
            cm, pm = O.romp_net_forward(sd, img)
            O._conv = orig_conv
            print('  %-14s vs f32 oracle: center %.3e params %.3e | vs f64: center %.3e params %.3e  [%d Winograd convs, %.0f %% of the MACs; max|x| %.1f max|V| %.1f]' % (
                mode, float((cm - cm32).abs().max()), float((pm - pm32).abs().max()), float((cm - cm64).abs().max()), float((pm - pm64).abs().max()),
                stats['hit'], 100.0 * stats['flops_hit'] / max(stats['flops'], 1.0), stats['max_x'], stats['max_V']), flush=True)


if __name__ == '__main__':
    main()
