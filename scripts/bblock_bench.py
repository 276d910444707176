"""Time the fused 32-channel BasicBlock kernel (csrc/conv_h2b.hip) against the same block as two conv launches.
usage: [BB_C=32|64] [BB_B=32] [BB_H=128|64] python scripts/bblock_bench.py      (GPU; ROMP_FUSE_BLOCKS=0 gives the unfused lowering)
The phase knock-outs (ROMP_CONV_DEBUG = 1 2 4 7 8 for the fused kernels) exist in a developer build only since round 6:
    python -c "from romp_amd import build as b; b.build(extra_flags=['-DROMP_BBLOCK_KNOCKOUTS'], lib='romp_amd/libromp_hip_ko.so', objdir='romp_amd/build_ko')"
    ROMP_HIP_LIB=romp_amd/libromp_hip_ko.so ROMP_CONV_DEBUG=8 python scripts/bblock_bench.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from romp_amd import lib as L
from romp_amd.plan import Program, Act, set_conv_math


def build(dev, H, fuse, Cc=32):
    os.environ['ROMP_FUSE_BLOCKS'] = 'all' if fuse else '0'
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(Cc, Cc, 3, 3, generator=g) / (Cc * 9) ** 0.5 for _ in range(3)]
    sc = [torch.rand(Cc, generator=g) + 0.5 for _ in range(3)]
    sh = [torch.randn(Cc, generator=g) * 0.2 for _ in range(3)]
    P = Program(dev)
    set_conv_math(P, 'f16x2')
    a0 = P.conv('c0', Act(L.BUF_IMAGE, Cc, H, H, Cc), [ws[0]], [sc[0]], [sh[0]], 3, 1, True)
    a1 = P.conv('c1', a0, [ws[1]], [sc[1]], [sh[1]], 3, 1, True)
    a2 = P.conv('c2', a1, [ws[2]], [sc[2]], [sh[2]], 3, 1, True, res=a0)
    ops = P.op_array()
    return P, ops


NAMES = {1: 'entry', 2: 'tables + halo DMA issued', 3: 'weights landed', 4: 'set-up', 11: 'conv1 units', 13: 'hand-over tail', 12: 'barrier A', 17: 'conv2 units', 14: 'finish tail', 15: 'barrier B'}


def trace_report(lib):
    """per-wave s_memtime stamps of the last fused launch (100 MHz ticks): where a tile's time goes"""
    import numpy as np
    words = 4096 * 64
    host = (C.c_uint64 * words)()
    assert lib.romp_conv_trace_read(host, words) > 0
    a = np.frombuffer(host, dtype=np.uint64).reshape(4096, 64)
    cnt = a[:, 0].astype(np.int64)
    live = np.nonzero(cnt > 1)[0]
    t = (a[:, 1:] >> np.uint64(8)).astype(np.int64)
    code = (a[:, 1:] & np.uint64(255)).astype(np.int64)
    agg = {}
    for wv in live:
        for i in range(1, cnt[wv]):
            agg.setdefault((int(code[wv, i - 1]), int(code[wv, i])), []).append(int(t[wv, i] - t[wv, i - 1]))
    span = np.array([t[wv, cnt[wv] - 1] - t[wv, 0] for wv in live])
    print('  trace: %d waves, entry -> last stamp mean %.0f max %.0f ticks (10 ns)' % (len(live), span.mean(), span.max()))
    for key, vals in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        vals = np.array(vals)
        print('   %-30s n/wave %5.1f  mean %7.1f  p50 %7.1f  max %7.0f  share %5.1f%%' % (
            '%s -> %s' % (NAMES.get(key[0], key[0]), NAMES.get(key[1], key[1])), len(vals) / len(live), vals.mean(), np.median(vals), vals.max(),
            100.0 * vals.sum() / span.sum()))
    wv = live[len(live) // 2]
    print('   wave %d:' % wv, ' '.join('%d@%d' % (code[wv, i], t[wv, i] - t[wv, 0]) for i in range(min(cnt[wv], 20))))


if __name__ == '__main__':
    Cc = int(os.environ.get('BB_C', '32'))
    B, H = int(os.environ.get('BB_B', '32')), int(os.environ.get('BB_H', '128' if Cc == 32 else '64'))
    dev = torch.device('cuda:0')
    lib = L.load()
    x = torch.randn(B, H, H, Cc, device=dev)
    dummy = torch.empty(16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for fuse in ((1,) if os.environ.get('BB_FUSED_ONLY') == '1' else (1, 0, 1, 0)):
        P, ops = build(dev, H, fuse, Cc)
        h = C.c_void_p()
        sizes = (C.c_int64 * len(P.buf_floats))(*P.buf_floats)
        L.check(lib.romp_net_create(C.byref(h), ops, len(P.ops), sizes, len(P.buf_floats), B))
        L.check(lib.romp_net_autotune(h, B, 3, st))
        ms = (C.c_float * len(P.ops))()
        L.check(lib.romp_net_profile(h, L.ptr(x), B, L.ptr(dummy), L.ptr(dummy), st, ms, 15))
        names = []
        buf = C.create_string_buffer(128)
        for i, op in enumerate(P.ops):
            L.check(lib.romp_conv_describe(C.byref(op), B, lib.romp_net_tuned_variant(h, B, i), buf, 128))
            names.append(buf.value.decode())
        print('fuse=%d  ' % fuse + '  '.join('%s %.1f us' % (n, t * 1e3) for n, t in zip(names, ms)) + '   block total %.1f us' % (sum(ms[1:]) * 1e3), flush=True)
        if fuse and os.environ.get('ROMP_CONV_TRACE') == '1':
            trace_report(lib)
        lib.romp_net_destroy(h)
