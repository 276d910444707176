"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/romp_hip.h
declares, the ctypes struct mirrors the C struct, the layer program is well formed, host helpers
behave, and the product path refuses to run without a HIP device (no silent CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'romp_hip.h')


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b((?:romp|smpl)_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from romp_amd import lib as L
    h = L.load()
    names = _declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(h, n), 'libromp_hip.so does not export %s' % n
    assert set(names) == set(L.EXPORTS), set(names) ^ set(L.EXPORTS)
    assert h.romp_abi_version() == L.ABI_VERSION == 7


def test_romp_op_struct_layout_matches_header():
    from romp_amd.lib import RompOp
    prog = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "romp_hip.h"
    int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(romp_op), offsetof(romp_op, groups),
        offsetof(romp_op, term_buf), offsetof(romp_op, weight), offsetof(romp_op, shift), offsetof(romp_op, act_shift),
        offsetof(romp_op, scale_h2), offsetof(romp_op, flags), offsetof(romp_op, relu_from), offsetof(romp_op, term_coff)); return 0; }
    '''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, 't.c')
        open(c, 'w').write(prog)
        exe = os.path.join(td, 't')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        vals = [int(v) for v in subprocess.check_output([exe]).split()]
    assert vals == [C.sizeof(RompOp), RompOp.groups.offset, RompOp.term_buf.offset, RompOp.weight.offset,
                    RompOp.shift.offset, RompOp.act_shift.offset, RompOp.scale_h2.offset, RompOp.flags.offset,
                    RompOp.relu_from.offset, RompOp.term_coff.offset]


def test_fused_block_dispatch_flag_follows_the_weight_pack(monkeypatch):
    """the fused-block launchers dispatch on ROMP_OPF_WAVE16, never on weight_aux != NULL.  conv_math='all'
    leaves the bf16x3 pack in every conv's weight_aux; a single-image plan fuses the 32-channel blocks WITHOUT the per-wave
    repack, so its BBLOCK32 ops must carry weight_aux (stale bf16x3 bytes) but NOT the flag; the batch plan repacks and flags.
    (The bf16x3 KERNELS are an optional part of the library since round 6; the pack is host code and is forced on here.)"""
    from romp_amd import lib as L, synthetic as S
    from romp_amd.plan import build_romp_hrnet32
    monkeypatch.setattr(L, 'has_bf16x3', lambda: True)
    sd = S.make_romp_state_dict(0)
    for math in ('all', 'f16x2'):
        single = build_romp_hrnet32(sd, 'cpu', 512, bf16x3=math, split_k_items=128)
        single.op_array()
        b32 = [i for i, o in enumerate(single.ops) if o.kind == L.OP_BBLOCK32]
        assert len(b32) == 32 and not any(o.kind == L.OP_BBLOCK64 for o in single.ops)
        for i in b32:
            for o in (single.ops[i - 1], single.ops[i]):
                assert not (o.flags & L.OPF_WAVE16)
                assert bool(o.weight_aux) == (math == 'all')
        batch = build_romp_hrnet32(sd, 'cpu', 512, bf16x3=math)
        batch.op_array()
        fused = [i for i, o in enumerate(batch.ops) if o.kind in (L.OP_BBLOCK32, L.OP_BBLOCK64, L.OP_SEAM1X1)]
        assert len(fused) == 32 + 32 + 3
        for i in fused:
            for o in (batch.ops[i - 1], batch.ops[i]):
                assert (o.flags & L.OPF_WAVE16) and o.weight_aux


def test_bf16x3_family_is_an_optional_part_of_the_library():
    """Round 6: no committed variant table selects a bf16x3 kernel, so conv_bx3.hip is compiled only on request
    (ROMP_WITH_BX3=1 python -m romp_amd.build).  A library without it says so (romp_conv_family_variants), `conv_math='bf16x3'`
    fails loudly instead of silently running other kernels, 'all' means every split family the build offers."""
    from romp_amd import lib as L
    from romp_amd.plan import Program, set_conv_math
    h = L.load()
    assert h.romp_conv_family_variants(0) > 0 and h.romp_conv_family_variants(11) > 0          # f32 MFMA, conv_h2g
    P = Program('cpu')
    if L.has_bf16x3():
        set_conv_math(P, 'bf16x3')
        assert P.bf16x3 and not P.f16x2
    else:
        with pytest.raises(L.RompHipError, match='ROMP_WITH_BX3'):
            set_conv_math(P, 'bf16x3')
        set_conv_math(P, 'all')
        assert P.f16x2 and not P.bf16x3


def test_no_cpu_fallback_in_product_path():
    import romp_amd
    from romp_amd.lib import RompHipError
    s = romp_amd.romp_settings([])
    s.GPU = -1
    with pytest.raises(RompHipError):
        romp_amd.ROMP(s, state_dict={}, smpl_model={})
    from romp_amd.post_parser import parsing_outputs, CenterMap
    with pytest.raises(RompHipError):
        parsing_outputs(torch.zeros(1, 1, 64, 64), torch.zeros(1, 64, 64, 145), CenterMap(0.25))
    from romp_amd.smpl import SMPL
    from romp_amd.synthetic import make_smpl_model
    with pytest.raises(RompHipError):
        SMPL(make_smpl_model())(torch.zeros(1, 10), torch.zeros(1, 72))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'romp_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt and 'oracle/' not in txt, f


def test_settings_match_reference_defaults():
    import romp_amd
    s = romp_amd.main.default_settings
    assert s.center_thresh == 0.25 and s.calc_smpl is True and s.render_mesh is False
    assert s.root_align is False and s.mode == 'image' and s.onnx is False and s.smooth_coeff == 3.0
    assert s.smpl_path.endswith(os.path.join('.romp', 'SMPL_NEUTRAL.pth'))
    assert s.model_path.endswith(os.path.join('.romp', 'ROMP.pkl'))
    assert {'ROMP', 'romp_settings', 'ResultSaver', 'WebcamVideoStream'} <= set(dir(romp_amd))


@pytest.fixture(scope='module')
def program():
    from romp_amd import synthetic as S
    from romp_amd.plan import build_romp_hrnet32
    return build_romp_hrnet32(S.make_romp_state_dict(0), 'cpu')


def test_program_matches_reference_inventory(program):
    """310 convs (stem conv1 + 309), 85.71 GFLOP; the three head first convs
    merge into one and the six tower convs into... (grouped), 23 fuse outputs."""
    from romp_amd.lib import OP_CONV, OP_FUSESUM, OP_STEM
    kinds = [o.kind for o in program.ops]
    assert kinds.count(OP_STEM) == 1 and kinds.count(OP_FUSESUM) == 23
    import re
    # round 4: the sibling stride-2 convs of a fuse layer (same input, model.py:198-221) are ONE op named fuse_layers.<i0>-<i1>.<j>.0
    merged = [re.search(r'fuse_layers\.(\d+)-(\d+)\.', n) for n in program.names]
    siblings = sum(int(m.group(2)) - int(m.group(1)) for m in merged if m)
    # stride-2 firsts: 4 stage-3 modules x 1 + 2 stage-4 modules x (2 + 1); 1x1 up-convs sharing a source: 4 x 1 + 3 x (1 + 2), the
    # last stage-4 module has output 0 only (nothing to merge)
    assert siblings == 4 * 1 + 2 * (2 + 1) + 4 * 1 + 2 * (1 + 2), siblings
    assert all(o.relu_from in (64, 128) for o, m, n in zip(program.ops, merged, program.names) if m and not n.endswith('.up'))
    n_conv_ref = sum(o.groups for o in program.ops if o.kind == OP_CONV) + 2 + 1 + siblings   # +2 merged head convs, +1 stem
    assert n_conv_ref == 310
    gflop = sum(program.flops) / 1e9
    assert abs(gflop - 85.71) < 0.2, gflop
    mb = sum(program.bytes) / 1e6
    assert 1100 < mb < 1200, mb                      # 1164 MB/img algorithmic


def test_program_buffer_liveness(program):
    """No op may read a buffer whose producer has since been overwritten, and the streams of a program may not race: replay the
    program symbolically, tagging each buffer with the op that last wrote it; plan.stream_races is the happens-before check (stream
    order + FORK / JOIN + the RECORD / WAIT edges of the open stage region, round 4)."""
    from romp_amd.lib import OP_CONV, OP_FORK, OP_FUSESUM, OP_FUSEUP, OP_JOIN, OP_RECORD, OP_STEM, OP_WAIT
    from romp_amd.plan import stream_races
    program.op_array()
    writer, region, recorded = {}, None, {}
    for i, op in enumerate(program.ops):
        if op.kind == OP_FORK:
            assert region is None and 1 <= op.Cin <= 3
            region = op.Cin
            continue
        if op.kind == OP_JOIN:
            assert region is not None and op.Cin == region
            region = None
            continue
        assert (op.stream == 0) if region is None else (0 <= op.stream <= region)
        if op.kind == OP_RECORD:
            assert region is not None and op.Cin not in recorded      # event numbers are unique per program
            recorded[op.Cin] = op.stream
            continue
        if op.kind == OP_WAIT:
            assert region is not None and recorded[op.Cin] != op.stream
            continue
        reads = []
        if op.kind in (OP_CONV, OP_STEM) and op.in_buf >= 0:
            reads.append(op.in_buf)
        if op.kind == OP_CONV and op.res_buf >= 0:
            reads.append(op.res_buf)
        if op.kind in (OP_FUSESUM, OP_FUSEUP):
            reads += [op.term_buf[k] for k in range(op.n_terms)]
        for b in reads:
            assert b in writer, 'op %d (%s) reads buffer %d before it was written' % (i, program.names[i], b)
            assert b != op.out_buf or b == program.head_in_buf, 'op %d runs in place on buffer %d' % (i, b)
        if op.out_buf >= 0:
            writer[op.out_buf] = i
    assert region is None and len(recorded) > 30
    assert program.head_in_buf in writer
    assert stream_races(program) == []
    # arena stays small because of reuse: < 100 MB per image although 323 ops produce ~560 MB of activations
    assert sum(program.buf_floats) * 4 / 1e6 < 100


@pytest.mark.parametrize('kind', ['romp_b1', 'romp_barriers', 'bev', 'resnet50'])
def test_every_plan_kind_is_race_free(kind, monkeypatch):
    """plan.stream_races on the other programs (Program.op_array asserts it too), and the check itself: dropping any one of a
    sample of WAIT edges, or releasing a cross-stream tensor an epoch early, must be reported."""
    from romp_amd import lib as L, synthetic as S
    from romp_amd.plan import build_romp_hrnet32, stream_races
    if kind == 'romp_barriers':
        monkeypatch.setenv('ROMP_DATAFLOW', '0')
    if kind == 'bev':
        from romp_amd.bev_plan import build_bev_hrnet32
        P = build_bev_hrnet32(S.make_bev_state_dict(0), 'cpu', 512, bf16x3='f16x2')
    elif kind == 'resnet50':
        from romp_amd.resnet_plan import build_romp_resnet50
        P = build_romp_resnet50(S.make_resnet_state_dict(0), 'cpu', 512, bf16x3='f16x2')
    else:
        P = build_romp_hrnet32(S.make_romp_state_dict(0), 'cpu', 512, bf16x3='f16x2', **(dict(split_k_items=128) if kind == 'romp_b1' else {}))
    P.op_array()
    assert stream_races(P) == []
    waits = [i for i, o in enumerate(P.ops) if o.kind == L.OP_WAIT]
    assert (len(waits) == 0) == (kind in ('romp_barriers', 'resnet50'))
    for i in waits[::5]:
        P.ops[i].kind = L.OP_NOP
        assert stream_races(P), 'dropping the wait at op %d went unnoticed' % i
        P.ops[i].kind = L.OP_WAIT


def test_conv_describe_every_op(program):
    from romp_amd import lib as L
    h = L.load()
    buf = C.create_string_buffer(128)
    for B in (1, 32):
        for op in program.ops:
            L.check(h.romp_conv_describe(C.byref(op), B, -1, buf, 128))
            assert buf.value


def test_weight_packing_roundtrip():
    from romp_amd.plan import pack_conv_weight, conv_pads, fold_bn
    w = torch.randn(70, 34, 3, 3)
    cin_pad, cout_pad = conv_pads(40, 70, 3)
    assert (cin_pad, cout_pad) == (40, 128)
    p = pack_conv_weight(torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 6)), cin_pad, cout_pad)
    assert p.shape == (9, 10, 128, 4)
    for (co, ci, ky, kx) in [(0, 0, 0, 0), (69, 33, 2, 1), (5, 17, 1, 2)]:
        assert p[ky * 3 + kx, ci // 4, co, ci % 4] == w[co, ci, ky, kx]
    assert p[:, :, 70:].abs().sum() == 0 and p[:, 8, :, 2:].abs().sum() == 0
    sd = {'b.weight': torch.tensor([2.0]), 'b.bias': torch.tensor([0.5]), 'b.running_mean': torch.tensor([1.0]),
          'b.running_var': torch.tensor([4.0 - 1e-5])}
    s, b = fold_bn(sd, 'b', 1, bias=torch.tensor([3.0]))
    assert abs(s.item() - 1.0) < 1e-6 and abs(b.item() - (0.5 - 1.0 + 3.0)) < 1e-6


def test_img_preprocess_contract():
    from romp_amd.utils import img_preprocess, resize_bicubic_u8
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (90, 160, 3)).astype(np.uint8)
    x, pad = img_preprocess(img)
    assert x.shape == (1, 512, 512, 3) and x.dtype == torch.float32
    assert pad.tolist() == [35.0, 125.0, 0.0, 160.0, 90.0, 160.0]
    assert x[0, :100].abs().sum() == 0                       # zero padding above the image
    # identity-size resize is exact; constant image stays constant
    same = resize_bicubic_u8(img[:64, :64], 64)
    assert np.array_equal(same, img[:64, :64])
    assert np.all(resize_bicubic_u8(np.full((40, 40, 3), 77, np.uint8), 512) == 77)
    # BGR -> RGB
    img2 = np.zeros((64, 64, 3), np.uint8); img2[..., 0] = 200
    y, _ = img_preprocess(img2)
    assert y[0, 256, 256].tolist() == [0.0, 0.0, 200.0]


def test_host_preprocess_fallback_matches_oracle():
    """Without cv2 the host pre-processing path runs OpenCV's fixed-point INTER_CUBIC restated in numpy: identical to the oracle
    (and so to the device kernel, which tests/test_bev_post.py holds to the same oracle)."""
    from oracle import cv_resize_oracle as CV
    from romp_amd.utils import img_preprocess
    rs = np.random.RandomState(3)
    for shp in ((90, 160), (240, 135), (64, 64), (700, 333)):
        img = rs.randint(0, 256, shp + (3,)).astype(np.uint8)
        a, pa = CV.img_preprocess(img)
        b, pb = img_preprocess(img)
        assert np.array_equal(a, b.numpy()) and pa.tolist() == pb.tolist()
    # the 11-bit tables: weights of a whole-pixel position are (0, 2048, 0, 0); every row sums to 2048 +- 1
    s0, w = CV.cubic_tables(64, 64)
    assert np.array_equal(w, np.tile([0, 2048, 0, 0], (64, 1))) and np.array_equal(s0, np.arange(64) - 1)
    _, w2 = CV.cubic_tables(1920, 512)
    assert np.abs(w2.sum(1) - 2048).max() <= 1


def test_translation_lsq_recovers_known_translation():
    from romp_amd.post_parser import estimate_translation_lsq
    rs = np.random.RandomState(1)
    X = rs.randn(5, 24, 3) * 0.3
    t = rs.randn(5, 3) * 0.2 + np.array([0, 0, 5.0])
    P = X + t[:, None]
    uv = 443.4 * P[:, :, :2] / P[:, :, 2:3] + 256
    est = estimate_translation_lsq(X, uv)
    np.testing.assert_allclose(est, t, atol=1e-4)


def test_shard_range_partitions():
    from romp_amd.distributed import shard_range
    for n, w in [(1024, 8), (10, 4), (3, 8), (32, 1)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_resnet50_plan_inventory():
    """ROMP ResNet-50 lowered on the CPU: 74 ops, 53.9 GFLOP / image (SURVEY.md §8a totals: backbone 49.1 + head), each
    transposed conv as four 2x2 parity convs with interleaving output strides."""
    from oracle import resnet_oracle as RO
    from romp_amd.resnet_plan import build_romp_resnet50
    P = build_romp_resnet50(RO.make_resnet_state_dict(0), 'cpu')
    assert len(P.ops) == 82      # (+ the fork / join around the three output convs and, round 6, around each layer's four parity convs)
    assert P.names.count('fork') == 4 and sorted({o.stream for n, o in zip(P.names, P.ops) if n.startswith('deconv0')}) == [0, 1, 2, 3]
    assert abs(sum(P.flops) / 1e9 - 53.79) < 0.05      # (the head.conv0 term counts the reference's 66 input channels, not the padded 80)
    dec = [(n, o) for n, o in zip(P.names, P.ops) if n.startswith('deconv')]
    assert len(dec) == 12 and all(o.ksize == 2 and o.stride == 1 and o.out_rstride > 0 and o.out_bstride > 0 for _, o in dec)
    assert sorted((o.pad_h, o.pad_w) for _, o in dec[:4]) == [(0, 0), (0, 1), (1, 0), (1, 1)]
    assert all(o.pad_h == -1 and o.pad_w == -1 for n, o in zip(P.names, P.ops) if o.kind == 2 and not n.startswith('deconv'))
    assert P.coord_off == 64 and P.head_in_ch == 80


@pytest.mark.parametrize('model', ['romp', 'bev', 'resnet50'])
def test_every_conv_has_a_kernel_for_its_formats(model):
    """Lower each network with conv_math='f16x2' on the CPU: every tensor gets a format (float32 or the pre-split H2), producer and
    consumers agree on it, and every conv op still has at least one kernel variant able to run it (an H2 input can only be read by
    the f16x2 kernels)."""
    from romp_amd import lib as L, synthetic as S
    h = L.load()
    if model == 'romp':
        from romp_amd.plan import build_romp_hrnet32 as build
        sd = S.make_romp_state_dict(0)
    elif model == 'bev':
        from romp_amd.bev_plan import build_bev_hrnet32 as build
        sd = S.make_bev_state_dict(0)
    else:
        from romp_amd.resnet_plan import build_romp_resnet50 as build
        sd = S.make_resnet_state_dict(0)
    P = build(sd, 'cpu', 512, bf16x3='f16x2')
    P.op_array()
    lowered = [(op.kind, op.in_fmt, op.res_fmt, op.out_fmt) for op in P.ops]
    P.op_array()                                             # (export.save_plan asks again: the lowering must not run twice)
    assert lowered == [(op.kind, op.in_fmt, op.res_fmt, op.out_fmt) for op in P.ops]
    buf = C.create_string_buffer(128)
    written = {}
    n_h2 = 0
    for name, op in zip(P.names, P.ops):
        if op.kind == L.OP_CONV:
            for b, f in ((op.in_buf, op.in_fmt), (op.res_buf, op.res_fmt)):
                if b >= 0 and b in written:
                    assert written[b] == f, '%s reads buffer %d as format %d, it was written as %d' % (name, b, f, written[b])
            valid = [v for v in range(h.romp_conv_num_variants()) if h.romp_conv_describe(C.byref(op), 32, v, buf, 128) == 0]
            assert valid, 'no kernel variant for %s (k%d s%d Cin %d Cout %d, in fmt %d)' % (name, op.ksize, op.stride, op.Cin, op.Cout, op.in_fmt)
            n_h2 += op.in_fmt == L.FMT_H2
            if op.out_buf >= 0:
                written[op.out_buf] = op.out_fmt
        elif op.kind == L.OP_FUSESUM:
            for k in range(op.n_terms):
                assert written.get(op.term_buf[k], op.term_fmt[k]) == op.term_fmt[k], name
            written[op.out_buf] = op.out_fmt
        elif op.kind == L.OP_FUSEUP:                          # a fuse sum with its 1x1 up-convs inside: every tensor H2
            for k in range(op.n_terms):
                assert written.get(op.term_buf[k]) == L.FMT_H2 and op.term_fmt[k] == L.FMT_H2, name
            assert op.out_fmt == L.FMT_H2 and (op.flags & L.OPF_WAVE16) and op.weight_aux, name
            written[op.out_buf] = op.out_fmt
        elif op.kind == L.OP_STEM:
            written[op.out_buf] = op.out_fmt
        elif op.kind == L.OP_STEM7P:                          # ResNet-50's fused stem + pool (round 6): reads the image, writes the pooled tensor
            assert P.ops[0].kind == L.OP_NOP and op is P.ops[1] and op.in_buf == L.BUF_IMAGE and op.ksize == 7 and op.weight, name
            written[op.out_buf] = op.out_fmt
        elif op.kind == L.OP_STEM2:                           # the fused stem (round 6): reads the image, writes y in H2
            assert op.out_fmt == L.FMT_H2 and (op.flags & L.OPF_WAVE16) and op.weight_aux and P.ops[0].kind == L.OP_NOP and op is P.ops[1], name
            n_h2 += 1
            written[op.out_buf] = op.out_fmt
        elif op.kind == L.OP_SEAM1X1:                         # fused Bottleneck seam: writes t (the NOP's output) and u, both H2
            assert op.in_fmt == L.FMT_H2 and op.out_fmt == L.FMT_H2, name
            n_h2 += 1
            written[op.out_buf] = op.out_fmt
        elif op.kind in (L.OP_BBLOCK32, L.OP_BBLOCK64):      # fused BasicBlock: reads x (its residual buffer), writes y, both H2
            assert written.get(op.res_buf) == L.FMT_H2 and op.res_fmt == L.FMT_H2 and op.out_fmt == L.FMT_H2, name
            n_h2 += 1
            written[op.out_buf] = op.out_fmt
        elif op.kind not in (L.OP_FORK, L.OP_JOIN, L.OP_NOP):
            for b in (op.in_buf, op.res_buf):
                assert b < 0 or written.get(b, L.FMT_F32) == L.FMT_F32, '%s (kind %d) would read an H2 tensor' % (name, op.kind)
            if op.out_buf >= 0:
                written[op.out_buf] = L.FMT_F32
    assert n_h2 > 50, n_h2


def test_range_guard_moves_out_of_range_tensors_to_float32():
    """f16x2 range safety (plan.assign_formats + RompNet calibration): given measured max|x| per op output, a tensor outside
    [H2_LO, H2_HI) is stored as float32, its consumer convs lose their f16x2 weights (and still have a kernel), everything
    else keeps the H2 format; without measurements nothing changes."""
    from romp_amd import lib as L, synthetic as S
    from romp_amd.plan import build_romp_hrnet32, h2_range_ok, H2_HI, H2_LO
    assert h2_range_ok(None) and h2_range_ok(0.0) and h2_range_ok(1.0) and h2_range_ok(H2_LO) and not h2_range_ok(H2_HI)
    assert not h2_range_ok(1e4) and not h2_range_ok(1e-6) and not h2_range_ok(float('inf'))
    h = L.load()
    sd = S.make_romp_state_dict(0)
    base = build_romp_hrnet32(sd, 'cpu', 512, bf16x3='f16x2')
    base.op_array()
    assert base.range_fallback == []
    conv_ids = [i for i, o in enumerate(base.ops) if o.kind == L.OP_CONV and o.out_fmt == L.FMT_H2 and o.out_buf >= 0]
    victim = conv_ids[len(conv_ids) // 2]
    P = build_romp_hrnet32(sd, 'cpu', 512, bf16x3='f16x2')
    P.op_maxabs = [1.0 if (o.out_buf >= 0 and o.kind not in (L.OP_FORK, L.OP_JOIN)) else None for o in P.ops]
    P.op_maxabs[victim] = 3.0e4
    P.op_array()
    assert P.ops[victim].out_fmt == L.FMT_F32 and base.ops[victim].out_fmt == L.FMT_H2
    hit = [i for i, _, m in P.range_fallback]
    assert hit and all(m == 3.0e4 for _, _, m in P.range_fallback)
    buf = C.create_string_buffer(128)
    for i in hit:
        op = P.ops[i]
        assert op.in_buf == P.ops[victim].out_buf and op.in_fmt == L.FMT_F32 and not op.weight_h2
        valid = [v for v in range(h.romp_conv_num_variants()) if h.romp_conv_describe(C.byref(op), 32, v, buf, 128) == 0]
        assert valid, 'no kernel left for %s' % P.names[i]
        for v in valid:
            h.romp_conv_describe(C.byref(op), 32, v, buf, 128)
            assert b'h2' not in buf.value
    same = sum(a.out_fmt == b.out_fmt and a.in_fmt == b.in_fmt for a, b in zip(P.ops, base.ops))
    assert same >= len(P.ops) - len(hit) - 4          # only the victim's live range changes


def test_plan_file_roundtrip_on_host(tmp_path):
    """export.save_plan / read_plan: every op field and every constant of the lowered ROMP program (single-image plan with
    split-K layers, f16x2 weights) survives the file; pointer fields become blob offsets; the head input's CoordConv
    channels travel as a buffer initialiser."""
    import ctypes as C
    from romp_amd import export, lib as L
    from romp_amd import synthetic as S
    from romp_amd.plan import build_romp_hrnet32, coord_channels, encode_h2
    P = build_romp_hrnet32(S.make_romp_state_dict(0), 'cpu', 512, bf16x3='f16x2', split_k_items=128)
    path = str(tmp_path / 'romp_b1.plan')
    export.save_plan(P, path, tuned={1: [-1] * len(P.ops)})
    plan = export.read_plan(path)
    ops = P.op_array()
    assert plan['abi'] == L.ABI_VERSION and plan['input_size'] == 512 and plan['buf_floats'] == list(P.buf_floats)
    assert plan['split_k_items'] == 128                     # the plan KIND is a header field (a batch plan writes 0, below)
    assert (plan['center_floats'], plan['params_floats']) == (64 * 64, 64 * 64 * 145) and list(plan['tuned']) == [1]
    by_ptr = {c.data_ptr(): c for c in P.consts if hasattr(c, 'data_ptr')}
    checked = 0
    for a, b in zip(ops, plan['ops']):
        for name, _ in L.RompOp._fields_:
            va, vb = getattr(a, name), getattr(b, name)
            if name in export.PTR_FIELDS:
                assert bool(va) == bool(vb)
                if va:
                    t = by_ptr[va]
                    raw = export.decode_pointer(plan, vb, t.numel() * t.element_size())
                    assert raw == t.contiguous().view(torch.uint8).numpy().tobytes()
                    checked += 1
            elif hasattr(va, '__len__'):
                assert list(va) == list(vb), name
            else:
                assert va == vb, name
    assert checked > 600
    (buf, floats, off), = plan['inits']
    coords = coord_channels(1, 128, 'cpu', P.head_in_ch, P.coord_off)
    if P.buf_fmt.get(P.head_in_buf) == L.FMT_H2:
        coords = encode_h2(coords)
    assert buf == P.head_in_buf and floats == coords.numel()
    assert plan['dev'][off:off + 4 * floats] == coords.contiguous().view(torch.uint8).numpy().tobytes()


def test_export_cli_bev_host_tables(tmp_path):
    """python -m romp_amd.export on a BEV checkpoint without a GPU: the program's HOST-side tables (scale anchors, the Conv3d
    refiners' weights) land in the plan's host blob and are referenced with the host bit set."""
    from romp_amd import export, synthetic as S
    from romp_amd.lib import OP_BEV_MAPS, OP_CONV3D
    ckpt, plan_path = str(tmp_path / 'BEV.pth'), str(tmp_path / 'bev.plan')
    torch.save(S.make_bev_state_dict(0), ckpt)
    export.main(['--model_path', ckpt, '-o', plan_path, '--bev'])
    plan = export.read_plan(plan_path)
    assert (plan['center_floats'], plan['params_floats']) == (64 * 128 * 128, 3 * 64 * 128 * 128) and plan['inits'] == []
    assert plan['split_k_items'] == 0                       # --max_batch 32 (the default): a batch plan
    host_ops = [o for o in plan['ops'] if o.kind in (OP_BEV_MAPS, OP_CONV3D)]
    assert len(host_ops) == 5 and all(o.weight & export.HOST_BIT for o in host_ops)
    maps = [o for o in host_ops if o.kind == OP_BEV_MAPS][0]
    anchors = np.frombuffer(export.decode_pointer(plan, maps.weight, 64 * 4), np.float32)
    from romp_amd.bev_plan import cam3dmap_anchor
    assert np.allclose(anchors, cam3dmap_anchor(60, 128))
    dev_ops = [o for o in plan['ops'] if o.weight and not (o.weight & export.HOST_BIT)]
    assert len(dev_ops) > 270


def test_pack_h2_wave16_is_the_documented_permutation():
    """plan.pack_h2_wave16 (weights of the 64-channel fused block, csrc/conv_h2c.hip): lane 16 kq + oc of wave w holds input
    channels 32 kc + 8 kq .. + 7 of output channel 16 w + oc, for every tap and piece."""
    from romp_amd.plan import pack_h2_wave16
    t = torch.arange(9 * 4 * 2 * 2 * 64 * 8, dtype=torch.int32).reshape(9, 4, 2, 2, 64, 8)
    n = pack_h2_wave16(t)
    assert tuple(n.shape) == (4, 9, 2, 2, 64, 8) and tuple(pack_h2_wave16(t[:, :2, :, :, :32]).shape) == (2, 9, 1, 2, 64, 8)
    rs = np.random.RandomState(0)
    for _ in range(200):
        w, tap, kc, pc, kq, oc, e = (rs.randint(k) for k in (4, 9, 2, 2, 4, 16, 8))
        assert n[w, tap, kc, pc, 16 * kq + oc, e] == t[tap, 2 * kc + kq // 2, pc, kq % 2, 16 * w + oc, e]


def test_hrnet_program_fusions_are_the_documented_ones(monkeypatch):
    """The lowered HRNet-32 program (f16x2, batch plan) fuses exactly what DESIGN.md section 4 says: 32 + 32 BasicBlocks (32- and
    64-channel) and the 3 Bottleneck seams of layer1 -- the first of them with Bottleneck 0's downsample conv folded in (round 5,
    OPF_SEAM_DS: one more NOP, two ops before that seam; ROMP_SEAM_DS=0 keeps it a launch); a single-image plan fuses the 32-channel
    blocks only; ROMP_FUSE_BLOCKS=0 / ROMP_FUSE_SEAMS=0 switch each off.  Every fused op sits right behind the NOP that carries its
    first conv."""
    from romp_amd import lib as L, synthetic as S
    from romp_amd.plan import build_romp_hrnet32
    sd = S.make_romp_state_dict(0)

    def kinds(**kw):
        P = build_romp_hrnet32(sd, 'cpu', 512, bf16x3='f16x2', **kw)
        P.op_array()
        ks = [op.kind for op in P.ops]
        for i, k in enumerate(ks):
            if k in (L.OP_BBLOCK32, L.OP_BBLOCK64, L.OP_SEAM1X1):
                assert ks[i - 1] == L.OP_NOP, (i, P.names[i])
            if k == L.OP_SEAM1X1 and (P.ops[i].flags & L.OPF_SEAM_DS):
                assert ks[i - 2] == L.OP_NOP and P.names[i - 2].endswith('layer1.0.downsample') and P.ops[i - 2].out_buf == P.ops[i - 1].res_buf
                assert P.ops[i - 2].in_buf not in (P.ops[i - 1].out_buf, P.ops[i].out_buf)      # x0 is intact while the seam reads it
        assert sum(bool(o.flags & L.OPF_SEAM_DS) for o in P.ops) == getattr(P, 'folded_downsamples', 0)
        return ks.count(L.OP_BBLOCK32), ks.count(L.OP_BBLOCK64), ks.count(L.OP_SEAM1X1), ks.count(L.OP_FUSEUP), ks.count(L.OP_FUSESUM), ks.count(L.OP_NOP)
    for v in ('ROMP_FUSE_BLOCKS', 'ROMP_FUSE_SEAMS', 'ROMP_FUSEUP', 'ROMP_MERGE_S2', 'ROMP_SEAM_DS', 'ROMP_FUSE_STEM2'):
        monkeypatch.delenv(v, raising=False)
    # round 6: the stem and conv2 are one launch (ROMP_OP_STEM2 behind the NOP that holds the stem): one more NOP in every lowering
    P6 = build_romp_hrnet32(sd, 'cpu', 512, bf16x3='f16x2')
    P6.op_array()
    assert P6.fused_stem2 == 1 and [o.kind for o in P6.ops[:2]] == [L.OP_NOP, L.OP_STEM2] and P6.ops[0].in_buf == L.BUF_IMAGE
    assert kinds() == (32, 32, 3, 16, 7, 67 + 18 + 1 + 1)
    monkeypatch.setenv('ROMP_FUSE_STEM2', '0')
    # round 4: 16 of the 23 fuse-layer outputs have up-terms: each runs as FUSEUP and the 18 (merged) 1x1 up-convs become NOPs
    assert kinds() == (32, 32, 3, 16, 7, 67 + 18 + 1)
    assert kinds(split_k_items=256) == (32, 0, 0, 0, 23, 32)
    monkeypatch.setenv('ROMP_SEAM_DS', '0')
    assert kinds() == (32, 32, 3, 16, 7, 67 + 18)
    monkeypatch.delenv('ROMP_SEAM_DS')
    monkeypatch.setenv('ROMP_FUSEUP', '0')
    assert kinds() == (32, 32, 3, 0, 23, 67 + 1)
    monkeypatch.setenv('ROMP_FUSE_BLOCKS', '0')
    assert kinds() == (0, 0, 3, 0, 23, 3 + 1)
    monkeypatch.setenv('ROMP_FUSE_SEAMS', '0')
    assert kinds() == (0, 0, 0, 0, 23, 0)


@pytest.mark.parametrize('cfg', [('romp', 'hrnet32', 32), ('romp', 'hrnet32', 128), ('romp', 'bev-hrnet32', 32), ('romp', 'resnet50', 32)],
                         ids=lambda c: '%s_%s_b%d' % c)
def test_committed_variant_tables_resolve(cfg):
    """romp_amd/tune/*.json are keyed by layer name: each must have an entry for every conv layer of today's program of its
    configuration, and every named kernel must exist in this build and be able to run its layer (romp_conv_describe needs no GPU).
    A plan change that renames layers, or a variant dropped from the library, fails here -- not as a silent fall-back to the
    heuristics on the GPU box."""
    import json, types
    from romp_amd import lib as L, synthetic as S, tuning
    workload, backbone, B = cfg
    if backbone == 'hrnet32':
        from romp_amd.plan import build_romp_hrnet32
        P = build_romp_hrnet32(S.make_romp_state_dict(0), 'cpu', 512, bf16x3='f16x2')
    elif backbone == 'bev-hrnet32':
        from romp_amd.bev_plan import build_bev_hrnet32
        P = build_bev_hrnet32(S.make_bev_state_dict(0), 'cpu', 512, bf16x3='f16x2')
    else:
        from romp_amd.resnet_plan import build_romp_resnet50
        P = build_romp_resnet50(S.make_resnet_state_dict(0), 'cpu', 512, bf16x3='f16x2')
    P.op_array()
    t = json.load(open(tuning.default_table_path(backbone, 'f16x2', B, workload)))
    assert t['batch'] == B and 'layers' in t
    convs = [n for n, o in zip(P.names, P.ops) if o.kind == L.OP_CONV]
    # every conv layer has an entry; an entry beyond that must name a layer a fusion has absorbed (a NOP today: layer1.0.downsample,
    # folded into the first seam in round 5 -- its entry keeps the ROMP_SEAM_DS=0 arm of an A/B run on the same table)
    absorbed = {n for n, o in zip(P.names, P.ops) if o.kind in (L.OP_NOP, L.OP_STEM2)}       # (stem.conv2: inside ROMP_OP_STEM2 since round 6; ROMP_FUSE_STEM2=0 arm)
    assert set(convs) <= set(t['layers']) and set(t['layers']) - set(convs) <= absorbed, set(convs) ^ set(t['layers'])
    variants, why = tuning.resolve_table(types.SimpleNamespace(lib=L.load(), program=P), B, t['layers'])
    assert variants is not None, why
    assert sum(v >= 0 for v in variants) == len(convs)


@pytest.mark.parametrize('env', [{'ROMP_MERGE_S2': '0'}, {'ROMP_FUSEUP': 'all', 'ROMP_FUSE_BLOCKS': 'all', 'ROMP_FUSE_SEAMS': 'all'},
                                 {'ROMP_KSPLIT_WG': '0'}, {'ROMP_MERGE_S2': '0', 'ROMP_FUSEUP': '0', 'ROMP_FUSE_BLOCKS': '0'}],
                         ids=lambda e: '+'.join('%s=%s' % kv for kv in e.items()))
def test_plan_switches_stay_race_free(env, monkeypatch):
    """The A/B switches of DESIGN.md change which ops exist and which buffers they share: every combination used in the notes must
    still lower to a program whose streams do not race (Program.op_array asserts it; this keeps the assertion exercised)."""
    from romp_amd import synthetic as S
    from romp_amd.plan import build_romp_hrnet32, stream_races
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sd = S.make_romp_state_dict(0)
    for kw in (dict(split_k_items=128), dict()):
        P = build_romp_hrnet32(sd, 'cpu', 512, bf16x3='f16x2', **kw)
        P.op_array()
        assert stream_races(P) == []


@pytest.mark.skipif(not (os.path.exists('/opt/rocm/bin/hipcc') or __import__('shutil').which('hipcc') or os.environ.get('HIPCC')),
                    reason='needs hipcc (compiles csrc/conv_h2x.hip to assembly)')
def test_counted_waits_are_covered_by_the_compiled_kernels():
    """Round 5: the seam kernel ends a tile on `s_waitcnt vmcnt(52)` / `vmcnt(20)` instead of a full drain -- right only while the
    COMPILED tile loop issues at least that many vector-memory instructions after the next tile's DMA (a wave's memory operations
    retire in issue order).  hipcc may merge or split loads and stores: scripts/check_counted_waits.py compiles csrc/conv_h2x.hip to
    assembly and counts (the fused-block kernels' vmcnt(6) / vmcnt(3) take a minute each to compile: `python
    scripts/check_counted_waits.py h2c h2c32`, run by hand when conv_h2c.h changes)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'check_counted_waits.py'), 'h2x'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(' ok') == 2 and 'NOT COVERED' not in r.stdout, r.stdout


def test_downsample_fold_needs_an_intact_block_input(monkeypatch):
    """plan.fuse_bottleneck_seams (round 5, OPF_SEAM_DS) lets the seam kernel compute Bottleneck 0's downsample conv from the block
    input x0 -- legal only while x0's arena buffer is still intact when the seam runs and nobody else reads the downsample's output.
    Same five convs, three lowerings: x0 kept (fold), x0 freed right after the downsample so that a later tensor re-uses its buffer
    (no fold: plain seam), a second reader of the downsample output (no fold)."""
    import torch
    from romp_amd import lib as L
    from romp_amd.plan import Program, Act, set_conv_math
    for v in ('ROMP_FUSE_SEAMS', 'ROMP_SEAM_DS'):
        monkeypatch.delenv(v, raising=False)
    g = torch.Generator().manual_seed(3)
    H = 16
    dims = [(64, 64), (64, 64), (64, 256), (64, 256), (256, 64), (256, 64)]
    ws = [torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5 for ci, co in dims]
    sc = [torch.ones(co) for _, co in dims]
    sh = [torch.zeros(co) for _, co in dims]

    def lower(free_x0_early, second_reader):
        P = Program('cpu')
        set_conv_math(P, 'f16x2')
        a0 = P.conv('x0', Act(L.BUF_IMAGE, 64, H, H, 64), [ws[0]], [sc[0]], [sh[0]], 1, 1, True)
        am = P.conv('m', a0, [ws[1]], [sc[1]], [sh[1]], 1, 1, True)
        ad = P.conv('d', a0, [ws[2]], [sc[2]], [sh[2]], 1, 1, False)
        if free_x0_early:
            P.free(a0)
        at = P.conv('t', am, [ws[3]], [sc[3]], [sh[3]], 1, 1, True, res=ad)
        au = P.conv('u', at, [ws[4]], [sc[4]], [sh[4]], 1, 1, True)
        if second_reader:
            P.conv('u2', ad, [ws[5]], [sc[5]], [sh[5]], 1, 1, True)
        P.op_array()
        return P, a0, at, au
    P, a0, at, au = lower(False, False)
    assert P.fused_seams == 1 and getattr(P, 'folded_downsamples', 0) == 1
    assert [o.kind for o in P.ops[:5]] == [L.OP_CONV, L.OP_CONV, L.OP_NOP, L.OP_NOP, L.OP_SEAM1X1] and P.ops[4].flags & L.OPF_SEAM_DS
    P, a0, at, au = lower(True, False)
    assert a0.buf in (at.buf, au.buf), 'the early free was meant to make a later tensor re-use the buffer of x0'
    assert P.fused_seams == 1 and getattr(P, 'folded_downsamples', 0) == 0 and P.ops[2].kind == L.OP_CONV
    assert not (P.ops[4].flags & L.OPF_SEAM_DS)
    P, a0, at, au = lower(False, True)
    assert P.fused_seams == 1 and getattr(P, 'folded_downsamples', 0) == 0 and P.ops[2].kind == L.OP_CONV

