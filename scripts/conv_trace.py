"""Phase timeline of one conv layer (GPU): the split-precision kernels stamp s_memtime at their phase boundaries
(ROMP_TRACE in csrc/conv_split.h); this prints, per kernel variant, where a wave's time goes.
usage: ROMP_CONV_TRACE=1 python scripts/conv_trace.py [B] [variant-substring ...]
Events: 1 entry, 2 first loads issued, 3 first stage staged, 4 first barrier, 10 stage start (next loads issued),
11 MFMA block done, 16 DMA landed, 12 barrier after MFMA, 13 next stage written, 14 epilogue issued, 15 barrier."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('ROMP_CONV_TRACE', '1')
import numpy as np  # noqa: E402
import torch  # noqa: E402
from romp_amd import lib as L  # noqa: E402
from romp_amd.plan import Program, Act, set_conv_math  # noqa: E402

NAMES = {1: 'entry', 2: 'loads issued', 3: 'stage0 staged', 4: 'barrier0', 10: 'next loads issued', 11: 'mfma done', 16: 'dma landed',
         12: 'barrier A', 13: 'lds written', 14: 'epilogue', 15: 'barrier B'}


def run(case, B, tags):
    cin, cout, k, s, H, use_res = case
    dev = torch.device('cuda:0')
    lib = L.load()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, H, cin, generator=g).to(dev)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    Ho = H // s
    res = torch.randn(B, Ho, Ho, cout, generator=g).to(dev) if use_res else None
    P = Program(dev)
    set_conv_math(P, 'all')
    P.buf_floats += [cin * H * H, cout * Ho * Ho]
    P.conv('t', Act(0, cin, H, H, cin), [w], [torch.ones(cout)], [torch.zeros(cout)], k, s, True,
           res=Act(1, cout, Ho, Ho, cout) if use_res else None)
    op = P.ops[0]
    from romp_amd.plan import ACT_SHIFT
    op.in_fmt = op.out_fmt = 1                   # H2 tensors (timing only: the input bytes are whatever randn left there)
    op.res_fmt = 1 if use_res else 0
    op.act_shift = ACT_SHIFT
    out = torch.empty(B, Ho, Ho, cout, device=dev)
    buf = C.create_string_buffer(128)
    st = torch.cuda.current_stream().cuda_stream
    words = 4096 * 64
    host = (C.c_uint64 * words)()
    for v in range(lib.romp_conv_num_variants()):
        if lib.romp_conv_describe(C.byref(op), B, v, buf, 128) != 0:
            continue
        name = buf.value.decode()
        if not any(t in name for t in tags):
            continue
        for _ in range(3):
            L.check(lib.romp_conv_forward(C.byref(op), L.ptr(x), L.ptr(res), L.ptr(out), B, 0, v, st))
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        L.check(lib.romp_conv_forward(C.byref(op), L.ptr(x), L.ptr(res), L.ptr(out), B, 0, v, st))
        e1.record()
        torch.cuda.synchronize()
        n = lib.romp_conv_trace_read(host, words)
        assert n > 0, lib.romp_last_error()
        a = np.frombuffer(host, dtype=np.uint64).reshape(4096, 64)
        cnt = a[:, 0].astype(np.int64)
        live = np.nonzero(cnt > 0)[0]
        if len(live) == 0:
            print(name, ': no stamps (not a split-precision kernel?)')
            continue
        t = (a[:, 1:] >> np.uint64(8)).astype(np.int64)
        code = (a[:, 1:] & np.uint64(255)).astype(np.int64)
        t0 = min(t[wv, 0] for wv in live)
        tend = max(t[wv, cnt[wv] - 1] for wv in live)
        print('\n%s  case %s B=%d: event-timed %.1f us; %d waves traced; first entry -> last stamp %d ticks' % (
            name, case, B, e0.elapsed_time(e1) * 1e3, len(live), tend - t0))
        # per-interval statistics: (prev code -> code)
        agg = {}
        for wv in live:
            c = cnt[wv]
            for i in range(1, c):
                key = (int(code[wv, i - 1]), int(code[wv, i]))
                agg.setdefault(key, []).append(int(t[wv, i] - t[wv, i - 1]))
        entry = np.array([t[wv, 0] - t0 for wv in live])
        exit_ = np.array([t[wv, cnt[wv] - 1] - t0 for wv in live])
        print('  wave entry  (ticks after the first): mean %.0f  p50 %.0f  max %.0f' % (entry.mean(), np.median(entry), entry.max()))
        print('  wave last stamp:                     mean %.0f  p50 %.0f  min %.0f  max %.0f' % (exit_.mean(), np.median(exit_), exit_.min(), exit_.max()))
        tot = sum(sum(v) for v in agg.values()) / len(live)
        print('  %-34s %8s %9s %9s %9s %7s' % ('interval', 'n/wave', 'mean', 'p50', 'max', 'share'))
        for key, vals in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            vals = np.array(vals)
            print('  %-34s %8.1f %9.0f %9.0f %9.0f %6.1f%%' % ('%s -> %s' % (NAMES.get(key[0], key[0]), NAMES.get(key[1], key[1])), len(vals) / len(live),
                                                             vals.mean(), np.median(vals), vals.max(), 100.0 * vals.sum() / len(live) / tot))
        # one wave's raw timeline
        wv = live[len(live) // 2]
        print('  timeline of wave %d:' % wv, ' '.join('%s@%d' % (code[wv, i], t[wv, i] - t0) for i in range(cnt[wv])))


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    tags = sys.argv[2:]
    if os.environ.get('TRACE_CASES'):            # "cin,cout,k,s,H,res;..." e.g. TRACE_CASES="32,128,3,2,128,0;64,64,3,2,256,0"
        for c in os.environ['TRACE_CASES'].split(';'):
            v = [int(t) for t in c.split(',')]
            run((v[0], v[1], v[2], v[3], v[4], bool(v[5])), B, tags or ['h2s'])
        sys.exit(0)
    cases = [((32, 32, 3, 1, 128, True), tags or ['h2_k3s1_mt1_nt1_tw16', 'h2d_k3s1_mt2_nt1_tw32']),
             ((64, 64, 3, 1, 64, True), tags or ['h2_k3s1_mt1_nt2_tw16', 'h2d_k3s1_mt2_nt2_tw16_ck16']),
             ((128, 128, 3, 1, 32, True), tags or ['h2_k3s1_mt1_nt2_tw16']),
             ((256, 256, 3, 1, 16, True), tags or ['h2_k3s1_mt1_nt2_tw16', 'h2_k3s1_mt1_nt1_tw16'])]
    for case, tg in cases:
        run(case, B, tg)
