"""Plan files: the lowered network as ONE file a host can run with nothing but libromp_hip.so.

The reference converts its model once to ``ROMP.onnx`` (``simple_romp/romp/model.py:484-497``) and its inference entry point
then needs no model code (``main.py:89,109``: ``onnxruntime.InferenceSession(path).run(...)``).  The counterpart here:
``save_plan(net_or_program, path)`` writes what ``plan.py`` lowered -- the op list, the arena sizes, every packed constant,
the initial contents of buffers that carry constants (the head's CoordConv channels) and the kernel-variant tables measured
by ``autotune`` -- and ``romp_net_load(path, max_batch)`` (C ABI; ``RompNet.from_plan`` in Python) builds the net from it.

Layout (little endian; the C reader is ``csrc/net.hip`` ``romp_net_load``)::

    PlanHeader   magic "ROMPPLAN", u32 version = 2, abi, n_ops, n_bufs, n_inits, n_tuned, input_size, sizeof(romp_op),
                 u32 split_k_items (the KIND of plan: > 0 = single-image plan lowered with that work-item target, 0 = batch plan),
                 u32 flags (0), u64 center_floats, params_floats (per image), dev_bytes, host_bytes
    i64          buf_floats[n_bufs]
    romp_op      ops[n_ops]            pointer fields: offset + 1 into the device blob; bit 63 set: into the host blob; 0: null
    PlanInit     inits[n_inits]        {i32 buf, i32 0, u64 floats, u64 device-blob offset}: per-image content of an arena buffer
    i32          tuned[n_tuned][2 + n_ops]   batch size, number of kernel variants of the exporting build, variant per op
    bytes        device blob (entries 256-byte aligned), host blob
"""
import ctypes as C
import struct

import numpy as np
import torch

from . import lib as L
from .lib import RompOp
from .plan import coord_channels, encode_h2

MAGIC = b'ROMPPLAN'
VERSION = 2
HEADER = struct.Struct('<8s10I4Q')
INIT = struct.Struct('<iiQQ')
PTR_FIELDS = ('weight', 'scale', 'shift', 'weight_aux', 'weight_h2', 'scale_h2')
HOST_BIT = 1 << 63


def _blobs(program):
    """-> (device blob, host blob, {address: encoded pointer}) over the constants the program keeps alive."""
    dev, host, where = bytearray(), bytearray(), {}
    for c in program.consts:
        if isinstance(c, torch.Tensor):
            raw = c.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes() if c.numel() else b''
            dev.extend(b'\0' * (-len(dev) % 256))
            where[c.data_ptr()] = len(dev) + 1
            dev.extend(raw)
        else:                                                    # ctypes array: a small table passed by value from the host
            raw = bytes(c)
            host.extend(b'\0' * (-len(host) % 16))
            where[C.addressof(c)] = (len(host) + 1) | HOST_BIT
            host.extend(raw)
    return dev, host, where


def save_plan(net_or_program, path, input_size=512, out_floats=None, tuned=None):
    """Write the plan file.  `net_or_program`: a RompNet (its program, output shapes and the variant tables it has measured)
    or a bare plan.Program (then pass `out_floats` = per-image floats of the two outputs).  The plan kind (`split_k_items` of
    the lowering: single-image or batch plan) goes into the header -- it is a property of how the program was lowered, not
    something the op list shows."""
    net = None if hasattr(net_or_program, 'ops') else net_or_program
    P = net.program if net is not None else net_or_program
    ops = P.op_array()
    n_ops = len(P.ops)
    if net is not None:
        input_size = net.input_size
        out_floats = tuple(int(np.prod(s)) for s in net.out_shapes)
        if tuned is None:
            tuned = {B: net.tuned_variants(B) for B in sorted(net._tuned)}
    if out_floats is None:
        ms = input_size // 8
        out_floats = (ms * ms, ms * ms * 145)
    tuned = tuned or {}
    dev, host, where = _blobs(P)
    packed = (RompOp * n_ops)()
    for i in range(n_ops):
        C.memmove(C.byref(packed[i]), C.byref(ops[i]), C.sizeof(RompOp))
        for f in PTR_FIELDS:
            v = getattr(ops[i], f)
            if v:
                if v not in where:
                    raise L.RompHipError('save_plan: op %d (%s) field %s points outside the program constants' % (i, P.names[i], f))
                setattr(packed[i], f, where[v])
    inits = []
    if P.coord_off is not None:                                  # the head input's constant CoordConv channels (model.py:473)
        fs = input_size // 4
        coords = coord_channels(1, fs, 'cpu', P.head_in_ch, P.coord_off)
        if P.buf_fmt.get(P.head_in_buf) == L.FMT_H2:
            coords = encode_h2(coords)
        raw = coords.contiguous().view(torch.uint8).numpy().tobytes()
        dev.extend(b'\0' * (-len(dev) % 256))
        inits.append((P.head_in_buf, coords.numel(), len(dev)))
        dev.extend(raw)
    n_variants = L.load().romp_conv_num_variants()
    with open(path, 'wb') as f:
        f.write(HEADER.pack(MAGIC, VERSION, L.ABI_VERSION, n_ops, len(P.buf_floats), len(inits), len(tuned), input_size, C.sizeof(RompOp),
                            int(getattr(P, 'split_k_items', 0) or 0), 0, out_floats[0], out_floats[1], len(dev), len(host)))
        f.write(np.asarray(P.buf_floats, dtype='<i8').tobytes())
        f.write(bytes(packed))
        for buf, floats, off in inits:
            f.write(INIT.pack(buf, 0, floats, off))
        for B, variants in sorted(tuned.items()):
            assert len(variants) == n_ops
            f.write(np.asarray([B, n_variants] + list(variants), dtype='<i4').tobytes())
        f.write(bytes(dev))
        f.write(bytes(host))
    return path


def read_plan(path):
    """Parse a plan file on the host (tests, inspection): dict with the header fields, `buf_floats`, `ops` (RompOp array with
    the ENCODED pointer fields), `inits`, `tuned`, `dev`, `host` (bytes)."""
    raw = open(path, 'rb').read()
    if len(raw) < HEADER.size:
        raise L.RompHipError('%s is truncated' % path)
    magic, version, abi, n_ops, n_bufs, n_inits, n_tuned, input_size, op_bytes, split_k_items, _flags, cf, pf, dev_bytes, host_bytes = HEADER.unpack_from(raw, 0)
    if magic != MAGIC or version != VERSION or op_bytes != C.sizeof(RompOp):
        raise L.RompHipError('%s is not a version-%d plan file of this ABI' % (path, VERSION))
    at = HEADER.size
    buf_floats = np.frombuffer(raw, '<i8', n_bufs, at).tolist(); at += 8 * n_bufs
    ops = (RompOp * n_ops).from_buffer_copy(raw, at); at += n_ops * op_bytes
    inits = [INIT.unpack_from(raw, at + k * INIT.size) for k in range(n_inits)]; at += n_inits * INIT.size
    tuned = {}
    for _ in range(n_tuned):
        row = np.frombuffer(raw, '<i4', 2 + n_ops, at); at += 4 * (2 + n_ops)
        tuned[int(row[0])] = (int(row[1]), row[2:].tolist())
    dev = raw[at:at + dev_bytes]; at += dev_bytes
    host = raw[at:at + host_bytes]; at += host_bytes
    if at != len(raw):
        raise L.RompHipError('%s: %d trailing bytes' % (path, len(raw) - at))
    return dict(abi=abi, input_size=input_size, split_k_items=split_k_items, center_floats=cf, params_floats=pf, buf_floats=buf_floats, ops=ops,
                inits=[(b, fl, off) for b, _, fl, off in inits], tuned=tuned, dev=dev, host=host)


def decode_pointer(plan, value, nbytes):
    """Bytes a packed pointer field of a parsed plan refers to."""
    if not value:
        return None
    off = (value & ~HOST_BIT) - 1
    blob = plan['host'] if value & HOST_BIT else plan['dev']
    return blob[off:off + nbytes]


def main(argv=None):
    """python -m romp_amd.export --model_path ROMP.pkl -o romp_b32.plan [--max_batch 32] [--tune 32 16] [--backbone resnet50] [--bev]

    Converts a checkpoint once (like the reference's ONNX export).  Lowering needs no GPU; with a HIP device present the kernel
    variant tables for the `--tune` batch sizes are measured and stored too."""
    import argparse
    ap = argparse.ArgumentParser(description='checkpoint -> libromp_hip plan file')
    ap.add_argument('--model_path', required=True)
    ap.add_argument('-o', '--out', required=True)
    ap.add_argument('--max_batch', type=int, default=32, help='<= 2 writes the single-image plan (split-K layers)')
    ap.add_argument('--conv_math', default='f16x2', choices=['f32', 'bf16x3', 'f16x2', 'all'])
    ap.add_argument('--backbone', default='hrnet32', choices=['hrnet32', 'resnet50'])
    ap.add_argument('--bev', action='store_true', help='the BEV network (BEV.pth)')
    ap.add_argument('--tune', type=int, nargs='*', default=[], help='batch sizes to measure kernel variants for (needs the GPU)')
    args = ap.parse_args(argv)
    sd = torch.load(args.model_path, map_location='cpu')
    if args.bev:
        from .bev_plan import build_bev_hrnet32 as builder
    elif args.backbone == 'resnet50':
        from .resnet_plan import build_romp_resnet50 as builder
    else:
        from .plan import build_romp_hrnet32 as builder
    split = dict(split_k_items=128) if args.max_batch <= 2 else {}
    if torch.cuda.is_available():
        from .net import RompNet
        shapes = ((64, 128, 128), (3, 64, 128, 128)) if args.bev else None
        net = RompNet(sd, 'cuda:0', max_batch=args.max_batch, builder=builder, bf16x3=args.conv_math, out_shapes=shapes)
        for B in args.tune:
            net.autotune(B)
        save_plan(net, args.out)
    else:
        out_floats = (64 * 128 * 128, 3 * 64 * 128 * 128) if args.bev else None
        save_plan(builder(sd, 'cpu', 512, bf16x3=args.conv_math, **split), args.out, out_floats=out_floats)
    print('wrote', args.out)


if __name__ == '__main__':
    main()
