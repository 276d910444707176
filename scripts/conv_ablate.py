"""Ablation timing of one conv layer (GPU): which part of the kernel costs what.
usage: [ABLATE_KIND=mfma|bx3|bxd|h2|h2d] [ABLATE_B=32] [ABLATE_DBG=0,4,...] python scripts/conv_ablate.py
(prints a table; outputs are wrong under dbg flags -- bit meanings: ConvParams::dbg in csrc/conv_mfma.hip)"""
import ctypes as C, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run_case(case, B, variants, dbg):
    import torch
    from romp_amd import lib as L
    from romp_amd.plan import Program, Act, set_conv_math
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
    if os.environ.get('ABLATE_FMT', 'h2') == 'h2':
        from romp_amd.plan import encode_h2, ACT_SHIFT
        op.in_fmt = op.out_fmt = 1
        op.res_fmt = 1 if use_res else 0
        op.act_shift = ACT_SHIFT
    out = torch.empty(B, Ho, Ho, cout, device=dev)
    buf = C.create_string_buffer(128)
    flops = 2.0 * B * Ho * Ho * cout * cin * k * k
    rows = []
    for v in range(lib.romp_conv_num_variants()):
        if lib.romp_conv_describe(C.byref(op), B, v, buf, 128) != 0:
            continue
        name = buf.value.decode()
        if variants and not any(t in name for t in variants):
            continue
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            L.check(lib.romp_conv_forward(C.byref(op), L.ptr(x), L.ptr(res), L.ptr(out), B, 0, v, st))
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        n = 20
        for _ in range(n):
            L.check(lib.romp_conv_forward(C.byref(op), L.ptr(x), L.ptr(res), L.ptr(out), B, 0, v, st))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        rows.append((name, ms * 1e3, flops / ms / 1e9))
    return rows

if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        case = tuple(json.loads(sys.argv[2])); variants = json.loads(sys.argv[3])
        for name, us, tf in run_case(case, int(os.environ.get('ABLATE_B', '32')), variants, int(os.environ.get('ROMP_CONV_DEBUG', '0'))):
            print('  dbg=%-3s %-36s %8.1f us %7.1f TF' % (os.environ.get('ROMP_CONV_DEBUG', '0'), name, us, tf))
        sys.exit(0)
    kind = os.environ.get('ABLATE_KIND', 'mfma')
    kinds = kind.split(',')
    vs = lambda *tags: [k + '_' + t for k in kinds for t in tags]
    cases = [((64, 64, 3, 1, 64, True), vs('k3s1_mt2_nt2_tw16_ck16', 'k3s1_mt2_nt1_tw16', 'k3s1_mt1_nt2_tw16', 'k3s1_mt1_nt2_tw32', 'k3s1_mt1_nt1_tw16')),
             ((32, 32, 3, 1, 128, True), vs('k3s1_mt2_nt1_tw16', 'k3s1_mt2_nt1_tw32', 'k3s1_mt4_nt1_tw32', 'k3s1_mt1_nt1_tw16')),
             ((128, 128, 3, 1, 32, True), vs('k3s1_mt2_nt2_tw16_ck16', 'k3s1_mt1_nt2_tw16'))]
    for case, variants in cases:
        print('case', case)
        for dbg in [int(x) for x in os.environ.get('ABLATE_DBG', '0,32,15,7,8,4,3,12').split(',')]:
            env = dict(os.environ, ROMP_CONV_DEBUG=str(dbg))
            r = subprocess.run([sys.executable, __file__, 'child', json.dumps(case), json.dumps(variants)], env=env,
                               capture_output=True, text=True)
            print(r.stdout.rstrip() or r.stderr[-500:])
