#!/usr/bin/env python
"""CPU emulation of split-precision conv arithmetic over the whole ROMP HRNet-32 network.

Every conv of the oracle network is replaced by a sum of f32 convolutions over low-precision PIECES of the
operands (products of two <=11-bit pieces are exact in f32, the accumulation is f32 like the MFMA's), and the
resulting maps are compared with the plain-f32 oracle (what the parity gate compares with) and with an f64 run.

    python scripts/precision_emul.py [B]
"""
import sys
import os
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import romp_oracle as O  # noqa: E402


def split(x, dt, n):
    ps = []
    r = x
    for _ in range(n):
        p = r.to(dt).to(torch.float32)
        ps.append(p)
        r = r - p
    return ps


def pow2_scale(t, target):
    """power of two s so that max|t|*s ~ target"""
    m = float(t.abs().max())
    if m == 0:
        return 1.0
    import math
    return 2.0 ** math.floor(math.log2(target / m))


MODES = {
    # name: (dtype, n_pieces, [(xi, wi) products])
    'bf16x3_6': (torch.bfloat16, 3, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
    'bf16x3_5': (torch.bfloat16, 3, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0)]),
    'bf16x3_3': (torch.bfloat16, 2, [(0, 0), (0, 1), (1, 0)]),
    'f16x2_3': (torch.float16, 2, [(0, 0), (0, 1), (1, 0)]),
    'f16x2_4': (torch.float16, 2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
    'f16x1_1': (torch.float16, 1, [(0, 0)]),
    'f16_x2w1_2': (torch.float16, 2, [(0, 0), (1, 0)]),       # activations 2 pieces, weights 1 piece
    'f16_x1w2_2': (torch.float16, 2, [(0, 0), (0, 1)]),       # activations 1 piece, weights 2 pieces
}
# "<mode>%<substring>": the mode only for convs whose state_dict name contains the substring, f16x2_3 (the product path) elsewhere


def make_conv(mode, act_scale_target=None, only=None):
    dt, n, prods0 = MODES[mode]
    stats = {'max_act': 0.0, 'hit': 0, 'flops_hit': 0.0, 'flops': 0.0}

    def conv(x, sd, name, stride=1):
        w = sd[name + '.weight']
        fl = float(w.numel()) * x.shape[2] * x.shape[3] / (stride * stride)
        stats['flops'] += fl
        prods = prods0
        if only is not None:
            if only in name and w.shape[-1] == 3:
                stats['hit'] += 1
                stats['flops_hit'] += fl
            else:
                prods = MODES['f16x2_3'][2]
        sx = sw = 1.0
        if dt == torch.float16:
            # per-tensor power-of-two scales (exact): keep the low pieces out of the fp16 subnormal range
            sw = pow2_scale(w, 256.0)
            sx = pow2_scale(x, 1024.0) if act_scale_target is None else act_scale_target
            stats['max_act'] = max(stats['max_act'], float(x.abs().max()))
        xp = split(x * sx, dt, n)
        wp = split(w * sw, dt, n)
        y = None
        for (i, j) in sorted(prods, key=lambda p: -(p[0] + p[1])):        # smallest terms first
            t = F.conv2d(xp[i], wp[j], None, stride=stride, padding=w.shape[-1] // 2)
            y = t if y is None else y + t
        y = y * (1.0 / (sx * sw))
        b = sd.get(name + '.bias')
        if b is not None:
            y = y + b.view(1, -1, 1, 1)
        return y
    return conv, stats


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    torch.set_num_threads(os.cpu_count())
    sd = O.make_romp_state_dict(0)
    img = O.make_images(B, seed=1)
    cm32, pm32 = O.romp_net_forward(sd, img)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    orig_conv, orig_coord = O._conv, O.coord_maps
    O.coord_maps = lambda size=128: orig_coord(size).double()
    cm64, pm64 = O.romp_net_forward(sd64, img.double())
    O.coord_maps = orig_coord
    print('f32 oracle vs f64: center %.3e params %.3e' % (float((cm32 - cm64).abs().max()), float((pm32 - pm64).abs().max())))
    for mode in sys.argv[2:] or MODES:
        fixed = only = None
        if '%' in mode:
            mode, only = mode.split('%')
        if '@' in mode:
            mode, fixed = mode.split('@')
            fixed = float(fixed)
        O._conv, stats = make_conv(mode, fixed, only)
        cm, pm = O.romp_net_forward(sd, img)
        O._conv = orig_conv
        print('%-12s vs f32 oracle: center %.3e params %.3e | vs f64: center %.3e params %.3e  (max act %.1f)' % (
            mode + ('@%g' % fixed if fixed else ''), float((cm - cm32).abs().max()), float((pm - pm32).abs().max()),
            float((cm - cm64).abs().max()), float((pm - pm64).abs().max()), stats['max_act']) +
              ('  [%s: %d convs, %.0f %% of the MACs]' % (only, stats['hit'], 100.0 * stats['flops_hit'] / max(stats['flops'], 1.0)) if only else ''), flush=True)


if __name__ == '__main__':
    main()
