"""Time EVERY valid kernel variant of a few conv layers (GPU), H2 tensors, back-to-back launches on one stream.
usage: [SWEEP_B=32] [SWEEP_FILTER=h2r,h2_,h2d] [SWEEP_CASES=all|s1|s2|k1] [SWEEP_CHECK=1] python scripts/conv_sweep.py
SWEEP_CHECK=1 also compares each variant's output with the float32-MFMA kernel family's (max-abs), so a sweep doubles as a
numerics smoke test of new variants at the benchmarked batch size."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

CASES = {
    's1': [(32, 32, 3, 1, 128, True), (64, 64, 3, 1, 64, True), (128, 128, 3, 1, 32, True), (256, 256, 3, 1, 16, True)],
    's2': [(32, 64, 3, 2, 128, False), (64, 128, 3, 2, 64, False), (128, 256, 3, 2, 32, False), (64, 64, 3, 2, 256, False),
           (32, 128, 3, 2, 128, False), (32, 96, 3, 2, 128, False), (64, 192, 3, 2, 64, False), (256, 64, 3, 2, 128, False), (48, 192, 3, 2, 128, False)],
    'k1': [(64, 256, 1, 1, 128, False), (256, 64, 1, 1, 128, True), (64, 64, 1, 1, 128, False)],
    # ResNet-50 layers 2-4 (resnet_50.py:64-78): conv1 4C -> C, conv3 C -> 4C + residual, the strided downsample
    'rn': [(256, 128, 1, 1, 128, False), (128, 512, 1, 1, 64, True), (512, 128, 1, 1, 64, False), (256, 512, 1, 2, 128, False),
           (256, 1024, 1, 1, 32, True), (1024, 256, 1, 1, 32, False), (512, 1024, 1, 2, 64, False),
           (512, 2048, 1, 1, 16, True), (2048, 512, 1, 1, 16, False), (1024, 2048, 1, 2, 32, False)],
}


def run_case(case, B, filt, check):
    from romp_amd import lib as L
    from romp_amd.plan import Program, Act, set_conv_math, encode_h2, decode_h2, ACT_SHIFT
    cin, cout, k, s, H, use_res = case
    dev = torch.device('cuda:0')
    lib = L.load()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    Ho = H // s
    res = torch.randn(B, Ho, Ho, cout, generator=g) if use_res else None
    P = Program(dev)
    set_conv_math(P, 'all')
    P.buf_floats += [cin * H * H, cout * Ho * Ho]
    P.conv('t', Act(0, cin, H, H, cin), [w], [torch.rand(cout, generator=g) + 0.5], [torch.randn(cout, generator=g) * 0.1], k, s, True,
           res=Act(1, cout, Ho, Ho, cout) if use_res else None)
    op = P.ops[0]
    op.in_fmt = op.out_fmt = 1
    op.res_fmt = 1 if use_res else 0
    op.act_shift = ACT_SHIFT
    xd = encode_h2(x).to(dev)
    rd = encode_h2(res).to(dev) if use_res else None
    out = torch.empty(B, Ho, Ho, cout, device=dev)
    buf = C.create_string_buffer(128)
    flops = 2.0 * B * Ho * Ho * cout * cin * k * k
    # algorithmic bytes: every input pixel the conv uses once, the output (and residual) once, the weights once
    abytes = 4.0 * (B * (Ho * Ho * cin * (1 if k == 1 else s * s) + Ho * Ho * cout * (2 if use_res else 1)) + cin * cout * k * k)
    st = torch.cuda.current_stream().cuda_stream
    rows, ref = [], None
    for v in range(lib.romp_conv_num_variants()):
        if lib.romp_conv_describe(C.byref(op), B, v, buf, 128) != 0:
            continue
        name = buf.value.decode()
        if filt and not any(t in name for t in filt):
            continue
        try:
            for _ in range(3):
                L.check(lib.romp_conv_forward(C.byref(op), L.ptr(xd), L.ptr(rd), L.ptr(out), B, 0, v, st))
            torch.cuda.synchronize()
        except Exception as e:                       # noqa: BLE001 -- report and go on with the sweep
            rows.append((name, v, float('inf'), 0.0, str(e)[:60]))
            continue
        err = ''
        if check:
            o = decode_h2(out.cpu())
            if ref is None:
                ref = o
            err = '%.2e' % (o - ref).abs().max().item()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        n = 30
        e0.record()
        for _ in range(n):
            L.check(lib.romp_conv_forward(C.byref(op), L.ptr(xd), L.ptr(rd), L.ptr(out), B, 0, v, st))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        rows.append((name, v, ms * 1e3, flops / ms / 1e9, '%6.0f GB/s  %s' % (abytes / ms / 1e6, err)))
    return rows


if __name__ == '__main__':
    B = int(os.environ.get('SWEEP_B', '32'))
    filt = [t for t in os.environ.get('SWEEP_FILTER', '').split(',') if t]
    which = os.environ.get('SWEEP_CASES', 's1')
    check = os.environ.get('SWEEP_CHECK', '0') == '1'
    cases = sum(CASES.values(), []) if which == 'all' else sum((CASES[k] for k in which.split(',')), [])
    for case in cases:
        print('case', case, 'B=%d' % B, 'dbg=%s' % os.environ.get('ROMP_CONV_DEBUG', '0'), flush=True)
        for name, v, us, tf, err in sorted(run_case(case, B, filt, check), key=lambda r: r[2]):
            print('  %-40s v%-3d %8.1f us %7.1f TF  %s' % (name, v, us, tf, err), flush=True)
