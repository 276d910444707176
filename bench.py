#!/usr/bin/env python
"""bench.py -- throughput of the ROMP inference hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (BASELINE.json): a synthetic job of `--global-batch` = 1024 pre-processed 512x512 images, sharded
contiguously over the N GPUs (BASELINE configs[2]); every rank walks its shard in forward calls of `--batch` = 32 images
(BASELINE configs[1], the batch the headline metric is quoted on): network (323 fused layer kernels, one hipGraph replay)
-> center-map parse -> SMPL meshes for every detection, then ONE RCCL all-gather of the per-person records of the whole
step over xGMI (N > 1).  A "step" is one pass over the whole job, so N = 1 times exactly the batch-32 hot path (32 calls
per step) and the per-N values form a STRONG-scaling curve.  `--global-batch 0` gives the weak-scaling mode of round 1
(`--batch` images per GPU per step).  Inputs are resident in HBM when the timed region starts.  Weights are seeded
synthetic tensors of the reference's shapes (the licensed ROMP.pkl / SMPL_NEUTRAL.pth cannot be shipped); arithmetic is
float32-accurate end to end (convs as f16x2-split products with f32 accumulation; `mesh_max_abs_vs_oracle` reports it).

Rank 0 prints ONE JSON line (see README / DESIGN.md for the field meanings).  After the timed region rank 0 additionally
measures (a) the per-kernel-class roofline with HIP events on the launch stream, (b) parity of the timed batch against the
oracle (maps, detections, meshes), (c) the same job end to end from uint8 host frames (H2D + device pre-processing inside the
timed region), (d) the same job with conv_math=f32, (e) the single-image latency of the drop-in API and (f) the CPU baseline (the oracle restatement of the reference, timed on the host cores of this box on a
bounded sample of the same workload).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_{bf16,f16} dense peak
BX3_PRODUCTS = 6                  # bf16 piece products issued per f32 product by the bf16x3 kernels
H2_PRODUCTS = 3                   # fp16 piece products issued per f32 product by the f16x2 kernels
PEAK_HBM_GBS = 8000.0
PROFILE_TAG = 'r06'               # profiles/<tag>_pmc_traffic_by_op.json: the committed rocprofv3 PMC passes of this build


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='images per forward call (BASELINE configs[1]: 32)')
    ap.add_argument('--global-batch', type=int, default=1024,
                    help='images per step over ALL GPUs, sharded contiguously (BASELINE configs[2]: 1024); 0: weak scaling, --batch images per GPU per step')
    ap.add_argument('--center-thresh', type=float, default=1.3)
    ap.add_argument('--workload', type=str, default='romp', choices=['romp', 'bev', 'smpl'],
                    help="romp = BASELINE configs[1]/[2] (default, the headline metric); bev = configs[3] (BEV head, 3-D parse, SMPL-A); smpl = configs[4] (SMPL-only, 64 persons)")
    ap.add_argument('--graph', type=int, default=1, help='replay the network from a hipGraph')
    ap.add_argument('--conv-math', type=str, default='f16x2', choices=['f32', 'bf16x3', 'f16x2', 'all'],
                    help='f32: exact f32 MFMA kernels only; f16x2 / bf16x3 / all: also offer the f32-accurate split-precision kernels '
                         '(2 fp16 pieces x 3 products / 3 bf16 pieces x 6 products) to the autotuner')
    ap.add_argument('--streams', type=int, default=1, help='run independent HRNet branches on side HIP streams')
    ap.add_argument('--autotune', type=int, default=1, help='pick conv kernel variants by measurement at start-up')
    ap.add_argument('--backbone', type=str, default='hrnet32', choices=['hrnet32', 'resnet50'],
                    help='resnet50: BASELINE configs[0]\'s model (the reference runs it on the CPU only) at the headline batch size')
    ap.add_argument('--tune-file', type=str, default=None,
                    help='kernel-variant table by name (romp_amd/tuning.py): installed if it exists (no measuring launches), else '
                         'written after autotuning.  Default: the committed table of this configuration under romp_amd/tune/, '
                         'if there is one; "none": always autotune')
    ap.add_argument('--dump-op-kernels', type=str, default=None,
                    help='write the per-op kernel names / algorithmic bytes of the batch to this JSON (scripts/summarize_pmc.py aligns '
                         'rocprofv3 dispatches with ops through it)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='budget of the CPU baseline sample')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the oracle comparison of the timed batch')
    ap.add_argument('--no-end-to-end', action='store_true', help='skip the uint8-host-frames (H2D + pre-processing) timing')
    ap.add_argument('--no-f32-companion', action='store_true', help='skip the extra conv_math=f32 timing of the same workload')
    ap.add_argument('--no-latency', action='store_true', help='skip the single-image ROMP(settings)(frame) latency leg')
    ap.add_argument('--with-verts', type=int, default=0, help='N>1: also all-gather the 6890x3 vertices')
    ap.add_argument('--cross-step', type=int, default=1,
                    help='1: keep the chunk pipeline primed across steps (the next step\'s first network is launched under this step\'s last '
                         'parse + SMPL + all-gather) and the record exchange at a fixed capacity (counts inside the one all-gather); 0: every '
                         'step fills and drains its own pipeline (rounds 1-5)')
    ap.add_argument('--preheat-cap', type=float, default=10.0,
                    help='seconds: UNcounted pre-heat steps in front of --warmup until two consecutive steps agree within 1 %% '
                         '(a fresh box runs its first steps slowly); 0: none.  The timed region stays exactly --steps steps')
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ roofline
def pmc_traffic(op_ids, op_names, pmc_dir, workload='', layer_names=None):
    """Measured HBM bytes per launch over the op index set `op_ids`, from the committed rocprofv3 --pmc passes of this same
    command (profiles/<tag><workload>_pmc_traffic_by_op.json, written by scripts/summarize_pmc.py --by-op from separate
    FETCH_SIZE / WRITE_SIZE passes: per op index the mean over the profiled forwards of 2 * FETCH_SIZE + WRITE_SIZE, the
    gfx950 correction of MI355X_MICROARCH.md).  The file records the kernel name each op ran as; a mismatch with this run
    (a different variant table) means the counters are about other kernels: None, never a stale quote."""
    path = os.path.join(pmc_dir, '%s%s_pmc_traffic_by_op.json' % (PROFILE_TAG, workload))
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    by_layer = {e['layer']: e for e in t['ops'].values() if 'layer' in e}      # (op indices move when stream markers are added; layer names do not)
    tot = 0.0
    for i in op_ids:
        e = by_layer.get(layer_names[i]) if (by_layer and layer_names is not None) else t['ops'].get(str(i))
        if e is None or e['kernel'] != op_names[i]:
            return None
        tot += e['bytes']
    return dict(bytes=tot / max(1, len(op_ids)), source=os.path.relpath(path, ROOT))


def roofline_report(net, images, pmc_workload=None):
    """Per kernel variant: sum of algorithmic FLOPs / bytes over its launches / sum of HIP-event durations (events recorded
    on the launch stream around every layer kernel, ops serialised on one stream)."""
    B = images.shape[0]
    ms = net.profile(images, iters=3)
    agg = {}
    op_names = net.variant_names(B)
    for i, (name, t, fl, by) in enumerate(zip(op_names, ms, net.program.flops, net.program.bytes)):
        if name in ('fork', 'join', 'nop', 'record', 'wait'):       # stream markers / the first conv of a fused pair: no launch of their own
            continue
        a = agg.setdefault(name, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, ops=[]))
        a['ms'] += t; a['flops'] += fl * B; a['bytes'] += by * B; a['launches'] += 1; a['ops'].append(i)
    classes = {}
    for k, a in agg.items():
        classes[k] = dict(launches=a['launches'], ms=round(a['ms'], 4),
                          tflops=round(a['flops'] / (a['ms'] * 1e-3) / 1e12, 2) if a['ms'] > 0 else 0.0,
                          gbs=round(a['bytes'] / (a['ms'] * 1e-3) / 1e9, 1) if a['ms'] > 0 else 0.0)
    name, a = max(agg.items(), key=lambda kv: kv[1]['ms'])
    achieved = a['flops'] / (a['ms'] * 1e-3) / 1e12
    # `achieved` counts ALGORITHMIC flops (2*M*N*K of the f32 convolution).  A split-precision kernel issues 6 (bf16x3) or
    # 3 (f16x2) 16-bit MFMA products per algorithmic product, so its matrix roof is the 16-bit dense peak / products; the f32
    # kernels are priced against the f32 MFMA peak.  The HBM view of the same kernel is reported beside it (hbm_frac): with
    # three products the 3x3 layers sit near the ridge, and whichever fraction is larger is the binding roof.
    bx3, h2 = 'conv_bx' in name, ('conv_h2' in name or name.startswith('bblock') or name in ('seam1x1', 'seam1x1_ds', 'fuseup', 'stem2'))
    products = BX3_PRODUCTS if bx3 else H2_PRODUCTS if h2 else 1
    peak = PEAK_BF16_MFMA_TFLOPS / products if (bx3 or h2) else PEAK_F32_MFMA_TFLOPS
    hbm_gbs = a['bytes'] / (a['ms'] * 1e-3) / 1e9
    mfma_frac, hbm_frac = achieved / peak, hbm_gbs / PEAK_HBM_GBS
    roof = dict(bound='hbm' if hbm_frac > mfma_frac else 'mfma', kernel=name)
    if roof['bound'] == 'hbm':
        roof.update(achieved=round(hbm_gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(hbm_frac, 4))
    else:
        roof.update(achieved=round(achieved, 2), peak=round(peak, 1), unit='TFLOP/s', frac=round(mfma_frac, 4))
    fused = name.startswith('bblock')          # the fused BasicBlock ops run csrc/conv_h2c.h's kernel in batch plans (16x16x32 MFMAs)
    if fused:
        roof['kernel_symbol'] = 'romp::bblockr_kernel<%s, 0, true> (the halo-carrying strip form; <%s, 0, false> where no run length pays)' % (name[len('bblock'):], name[len('bblock'):]) + \
                                (' (single-image plans: romp::bblock32_kernel<0>)' if name == 'bblock32' else '')
    roof.update(traffic=None,
                pipe=('%s MFMA %s, %d piece products per f32 product' % ('bf16' if bx3 else 'f16', '16x16x32' if fused else '32x32x16', products)) if (bx3 or h2) else 'f32 MFMA 32x32x2',
                tflops=round(achieved, 2), mfma_peak_tflops=round(peak, 1), mfma_frac=round(mfma_frac, 4),
                hbm_gbs=round(hbm_gbs, 1), hbm_frac=round(hbm_frac, 4),
                issued_tflops=round(achieved * products, 1), frac_of_f32_mfma_peak=round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                launches=a['launches'], avg_launch_ms=round(a['ms'] / a['launches'], 5),
                flops_per_launch=a['flops'] / a['launches'], alg_bytes_per_launch=a['bytes'] / a['launches'],
                net_ms_per_batch=round(sum(ms), 3), batch=B)
    # PMC counters cannot be read from inside the process: `traffic` comes from the committed rocprofv3 --pmc passes of the
    # DEFAULT workload (profiles/README.md) and is reported only for that workload (a kernel name alone does not identify the
    # layers behind it: the ResNet-50 / BEV / other-batch lines carry null) and only if the dominant kernel is in those passes
    # pmc_workload: '' = the default workload, '_resnet50' / '_bev' = their own passes, None = no passes for this configuration
    t = pmc_traffic(a['ops'], op_names, os.path.join(ROOT, 'profiles'), pmc_workload, list(net.program.names)) if pmc_workload is not None else None
    if t is None:
        roof['traffic_note'] = ('null: no committed PMC passes (profiles/%s*_pmc_traffic_by_op.json) for this configuration, or they were '
                                'taken with another kernel-variant table' % PROFILE_TAG)
    else:                                    # measured in separate rocprofv3 --pmc passes of this command, same variant table
        roof['traffic'] = round(t['bytes'])
        roof['traffic_source'] = '%s: mean of 2*FETCH_SIZE + WRITE_SIZE over exactly the %d ops of this kernel class' % (t['source'], len(a['ops']))
        roof['traffic_over_algorithmic'] = round(t['bytes'] / (a['bytes'] / a['launches']), 3)
    roof['op_ids'] = a['ops']
    return roof, classes


def usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota and by
    the physical core count (SMT siblings do not help MKLDNN convolutions; 256 threads on the
    2x64-core box ran 50x slower than 64)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        cores = set()
        phys = core = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
                cores.add((phys, core))
        if cores:
            n = min(n, len(cores))
    except Exception:
        pass
    return max(1, min(n, 64))


# ------------------------------------------------------------------------------------------------ CPU baseline / parity
def cpu_baseline(sd, smpl_model, thresh, seconds, backbone='hrnet32'):
    """The reference's CPU path on a bounded sample of the same workload, on this box's host cores: the REFERENCE ITSELF
    (ROMPv1 + parsing_outputs + SMPL of simple_romp/romp, run from oracle/_ref/romp/*.pyc, kind "reference") when
    oracle/Makefile staged it, else the oracle restatement (kind "port").  --backbone resnet50: the reference's ResNet-50 lives
    in its training tree (romp/lib/models/resnet_50.py), whose package cannot be staged sourceless the same way -- the restated
    oracle (oracle/resnet_oracle.py, pinned to the reference module by tests/golden/resnet50_b1.npz) is timed: kind "port"."""
    from oracle import romp_oracle as O, ref_cpu
    from romp_amd import synthetic as S
    torch.set_num_threads(usable_cores())
    Bc = 4
    img = S.make_images(Bc, seed=1)
    kind = 'reference' if (ref_cpu.available() and backbone == 'hrnet32') else 'port'
    net_oracle = O.romp_net_forward
    if backbone == 'resnet50':
        from oracle import resnet_oracle as RO
        net_oracle = RO.resnet_romp_forward
    if kind == 'reference':
        pipe = ref_cpu.ReferencePipeline(sd, smpl_model, thresh)

        def step():
            pipe(img)
    else:
        def step():
            cm, pm = net_oracle(sd, img)
            r = O.parsing_outputs(cm.numpy(), pm.numpy(), thresh)
            if r is not None:
                O.smpl_forward(smpl_model, r['smpl_betas'], r['smpl_thetas'])
    t0 = time.time()
    step()                                    # warm-up (also bounds the sample on a slow host)
    warm = time.time() - t0
    t0, n = time.time(), 0
    while warm < seconds / 2:
        step(); n += 1
        if time.time() - t0 > seconds or n >= 8:
            break
    dt = time.time() - t0
    if n == 0:
        n, dt = 1, warm
    what = ('the reference itself: ROMPv1 (simple_romp/romp/model.py:420-481) + parsing_outputs (post_parser.py:135-146) + SMPL '
            '(smpl.py:62-108), PyTorch-CPU float32, byte-compiled from /root/reference into oracle/_ref/romp (its onnxruntime '
            'session, main.py:86-89, needs a module that is not installed: this is the --onnx=False path, main.py:74-77)'
            if kind == 'reference' else
            'the oracle restatement of the reference ResNet-50 ROMP (romp/lib/models/resnet_50.py:19-120 + the ROMPv1 head, parse, SMPL) on '
            'torch-CPU float32: oracle/resnet_oracle.py' if backbone == 'resnet50' else
            'the oracle restatement of the reference on torch-CPU float32 (oracle/_ref/romp is not staged on this box)')
    return dict(value=round(Bc * n / dt, 3), unit='images/s', cores=torch.get_num_threads(), kind=kind,
                sample='%d iterations of batch %d (net+parse+SMPL) of the same synthetic workload: %s' % (n, Bc, what))


def parity_report(model, images, sd, smpl_model, thresh, mesh_pick=(0, 7, 17, 31), net_oracle=None):
    """The WHOLE first call of the timed job (all `images`, 32 by default): HIP maps and detections of every image against the
    oracle pipeline, and the meshes of every person of the images `mesh_pick` (BASELINE metric: 'mesh max-abs-err vs ref').
    `net_oracle(sd, images_nhwc) -> (center_maps, params_maps)`: the oracle network (default: HRNet-32, oracle/romp_oracle.py;
    --backbone resnet50 passes oracle/resnet_oracle.py's)."""
    import numpy as np
    from oracle import romp_oracle as O
    torch.set_num_threads(usable_cores())
    net_oracle = net_oracle or O.romp_net_forward
    n = images.shape[0]
    cm, pm = model.model(images)
    out, bids = model.forward_batch(images)
    torch.cuda.synchronize()
    cms, pms = [], []
    for c0 in range(0, n, 8):                                 # (8 images at a time: bounds the host memory of the float32 oracle)
        c_, p_ = net_oracle(sd, images[c0:c0 + 8].cpu())
        cms.append(c_); pms.append(p_)
    cm_o, pm_o = torch.cat(cms), torch.cat(pms)
    per_image = torch.maximum((cm.cpu() - cm_o).abs().flatten(1).max(1).values, (pm.cpu() - pm_o).abs().flatten(1).max(1).values)
    rep = {'images_compared': list(range(n)), 'maps_max_abs_vs_oracle': float(per_image.max()),
           'maps_max_abs_worst_image': int(per_image.argmax())}
    ref = O.parsing_outputs(cm_o.numpy(), pm_o.numpy(), thresh)
    if out is None or ref is None:
        rep['detections_equal'] = out is None and ref is None
        return rep
    b = bids.cpu().numpy()
    same = (len(b) == len(ref['batch_ids']) and np.array_equal(b, ref['batch_ids']) and
            np.array_equal(out['center_preds'].cpu().numpy(), ref['center_preds']))
    rep['detections_equal'] = bool(same)
    rep['persons_compared'] = int(len(b))
    rep['thetas_max_abs_vs_oracle'] = float(np.abs(out['smpl_thetas'].cpu().numpy() - ref['smpl_thetas']).max()) if same else None
    mesh_pick = [i for i in mesh_pick if i < n]
    rows = np.nonzero(np.isin(b, mesh_pick))[0]
    rep['mesh_images_compared'] = mesh_pick
    if same and len(rows):
        th, be = out['smpl_thetas'].cpu().numpy()[rows], out['smpl_betas'].cpu().numpy()[rows]
        v = out['verts'].cpu().numpy()[rows]
        vo, _, _ = O.smpl_forward(smpl_model, be, th)                                   # identical theta / beta (the 1e-4 gate)
        vr, _, _ = O.smpl_forward(smpl_model, ref['smpl_betas'][rows], ref['smpl_thetas'][rows])    # the oracle's own theta / beta
        rep['mesh_persons_compared'] = int(len(rows))
        rep['mesh_max_abs_vs_oracle'] = float(np.abs(v - vo).max())
        rep['mesh_max_abs_end_to_end'] = float(np.abs(v - vr).max())
    return rep


def end_to_end(model, lib, L, dev, batch, n_calls, stream):
    """The same job starting from uint8 frames in pinned host memory: H2D of 0.79 MB/image + batched device pre-processing
    (BGR->RGB, pad, cv::resize-exact INTER_CUBIC; csrc/post.hip) + net + parse + SMPL, all inside the timed region.  The upload of
    call i+1 runs on a copy stream under the compute of call i (two pinned host buffers, two device frame buffers, events both
    ways); `serial` is the same loop with the copy on the compute stream (round 2's figure)."""
    g = torch.Generator().manual_seed(11)
    frames = [torch.randint(0, 256, (batch, 512, 512, 3), generator=g, dtype=torch.uint8).pin_memory() for _ in range(2)]
    fdev = [torch.empty(batch, 512, 512, 3, device=dev, dtype=torch.uint8) for _ in range(2)]
    x = torch.empty(batch, 512, 512, 3, device=dev, dtype=torch.float32)
    pad = (C.c_float * 6)()
    copy_s = torch.cuda.Stream(dev)
    copied = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]

    def upload(i, overlap):
        k = i & 1
        if overlap:
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(consumed[k])               # the pre-processing that last read fdev[k] is done
                fdev[k].copy_(frames[k], non_blocking=True)
                copied[k].record(copy_s)
        else:
            fdev[k].copy_(frames[k], non_blocking=True)

    def run(overlap):
        for k in range(2):
            consumed[k].record(torch.cuda.current_stream(dev))
        if overlap:
            upload(0, True)                                  # pipeline prologue; the loop below issues n_calls uploads as well
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(n_calls):
            k = i & 1
            if overlap:
                upload(i + 1, True)                          # next call's frames: in flight under this call's network
                torch.cuda.current_stream(dev).wait_event(copied[k])
            else:
                upload(i, False)
            L.check(lib.romp_preprocess_batch(L.ptr(fdev[k]), batch, 512, 512, L.ptr(x), 512, pad, L.stream_ptr(dev)))
            consumed[k].record(torch.cuda.current_stream(dev))
            model.forward_batch(x)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    with torch.cuda.stream(stream):
        run(True)                                            # warm-up
        dt = run(True)
        dts = run(False)
    return dict(value=round(batch * n_calls / dt, 2), unit='images/s', ms_per_call=round(dt / n_calls * 1e3, 3), calls=n_calls,
                serial=dict(value=round(batch * n_calls / dts, 2), ms_per_call=round(dts / n_calls * 1e3, 3)),
                includes='per call: H2D of %d uint8 512x512x3 frames from pinned host memory (copy stream, overlapped with the previous '
                         "call's compute) + device pre-processing + net + parse + SMPL" % batch)


def single_image_latency(sd, smpl_model, args, dev, stream, n=40):
    """The reference's only published speed is one image at a time (docs/romp_evaluation.md:96-102: 23.8 FPS on a GTX 1070Ti):
    `romp.ROMP(settings)(bgr_frame)` on a 720p frame -- upload + device pre-processing + single-image network plan (split-K
    lowering, hipGraph) + parse + SMPL + projection + download of the result dict -- and the network alone at B = 1."""
    import numpy as np
    import romp_amd
    from romp_amd import synthetic as S
    s = romp_amd.romp_settings([])
    s.GPU, s.center_thresh, s.max_batch, s.conv_math = dev.index or 0, args.center_thresh, 1, args.conv_math
    m = romp_amd.ROMP(s, state_dict=sd, smpl_model=smpl_model)
    m.model.set_graph(True)
    frame = np.random.RandomState(0).randint(0, 256, (720, 1280, 3)).astype(np.uint8)
    with torch.cuda.stream(stream):
        for _ in range(5):
            out = m(frame)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            out = m(frame)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / n
        x = S.make_images(1, seed=1, device=dev)
        c, p = m.model.forward_nhwc(x)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            m.model.forward_nhwc(x, c, p)
        torch.cuda.synchronize(dev)
        dn = (time.perf_counter() - t0) / n
    return {'fps': round(1.0 / dt, 1), 'ms_per_frame': round(dt * 1e3, 3), 'network_ms': round(dn * 1e3, 3), 'frame': '1280x720 uint8 BGR from host memory',
            'persons': 0 if out is None else int(out['cam'].shape[0]), 'calls': n,
            'includes': 'romp.ROMP(settings)(frame): H2D + pad/resize + network (single-image plan: %d launches) + parse + SMPL + projection + D2H of the result dict'
                        % sum(nm not in ('fork', 'join', 'nop', 'record', 'wait') for nm in m.model.variant_names(1))}


# ------------------------------------------------------------------------------------------------ other workloads
def bench_smpl(args, dev):
    """BASELINE configs[4]: SMPL-only microbench, 64 persons x 6890 verts (blend shapes + LBS + 71 joints), HIP vs
    the torch-CPU oracle.  HBM-bound: per launch the kernels read betas/thetas, the blend-shape bases (v_template,
    shapedirs, posedirs 17.1 MB, skinning weights) once per person tile from L2/HBM and write N x (6890 + 71) x 3 floats."""
    import numpy as np
    from romp_amd import synthetic as S
    from romp_amd.smpl import SMPL
    N = 64
    model = S.make_smpl_model(0)
    smpl = SMPL(model).to(dev)
    g = torch.Generator().manual_seed(3)
    betas = torch.randn(N, 10, generator=g).to(dev)
    poses = (0.3 * torch.randn(N, 72, generator=g)).to(dev)
    for _ in range(max(args.warmup, 3)):
        v, j, _ = smpl(betas, poses)
    torch.cuda.synchronize(dev)
    steps = max(args.steps, 50)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        v, j, _ = smpl(betas, poses)
    e1.record()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1) / steps
    # algorithmic bytes per launch: bases read once + outputs written once
    nb = 10
    base_bytes = 4 * (6890 * 3 * (1 + nb) + 207 * 20670 + 6890 * 24 + (24 + 9 + 17) * 6890)
    out_bytes = 4 * N * (6890 + 71) * 3
    flops = 2.0 * N * (6890 * 3 * nb + 207 * 20670 + 6890 * 24 * 12 + 6890 * 12 + (9 + 17) * 6890 * 3)
    res = {'metric': 'SMPL meshes/sec (64 persons x 6890 verts)', 'value': round(N * steps / dt, 1), 'unit': 'meshes/s', 'n_gpus': 1,
           'steps': steps, 'warmup': max(args.warmup, 3), 'ms_per_step': round(dt / steps * 1e3, 4), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': 'SMPL-only: 64 persons, betas (64,10) + thetas (64,72) -> verts (64,6890,3), joints (64,71,3) (BASELINE configs[4])',
                      'gpu_ms_per_launch': round(gpu_ms, 4)},
           'roofline': {'bound': 'hbm', 'achieved': round((base_bytes + out_bytes) / (gpu_ms * 1e-3) / 1e9, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': round((base_bytes + out_bytes) / (gpu_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), 'traffic': None,
                        'alg_bytes_per_launch': base_bytes + out_bytes, 'gflops_per_launch': round(flops / 1e9, 3),
                        'note': 'the SMPL kernels of one call (pose, skin, joints); latency bound at N=64'}}
    if not args.no_cpu_baseline:
        from oracle import romp_oracle as O, ref_cpu
        torch.set_num_threads(usable_cores())
        bn, pn = betas.cpu().numpy(), poses.cpu().numpy()
        vo, jo, _ = O.smpl_forward(model, bn, pn)
        if ref_cpu.available():          # the reference's own SMPL module (oracle/_ref/romp/smpl.pyc: simple_romp/romp/smpl.py:62-108), PyTorch-CPU
            ref_smpl = ref_cpu.reference_smpl(model)
            bc, pc = betas.cpu(), poses.cpu()
            with torch.no_grad():
                vr = ref_smpl(bc, pc)[0]
                t0, n = time.time(), 0
                while time.time() - t0 < min(args.cpu_seconds, 10.0) and n < 200:
                    ref_smpl(bc, pc); n += 1
            cdt = time.time() - t0
            res['cpu_baseline'] = dict(value=round(N * n / cdt, 1), unit='meshes/s', cores=torch.get_num_threads(), kind='reference',
                                       sample='%d calls of the reference\'s SMPL.forward (simple_romp/romp/smpl.py:62-108, byte-compiled from '
                                              '/root/reference into oracle/_ref/romp) at N=64, PyTorch-CPU float32' % n)
            res['config']['verts_max_abs_vs_reference'] = float((v.cpu() - vr).abs().max())
        else:
            t0, n = time.time(), 0
            while time.time() - t0 < min(args.cpu_seconds, 10.0) and n < 200:
                O.smpl_forward(model, bn, pn); n += 1
            cdt = time.time() - t0
            res['cpu_baseline'] = dict(value=round(N * n / cdt, 1), unit='meshes/s', cores=torch.get_num_threads(), kind='port',
                                       sample='%d calls of the torch-CPU restatement of smpl.py lbs at N=64' % n)
        res['config']['verts_max_abs_vs_oracle'] = float(np.abs(v.cpu().numpy() - vo).max())
    print(json.dumps(res), flush=True)


def bev_parity_report(model, images, sd, thresh, pick=(0, 7, 19, 31)):
    """Images `pick` of the timed BEV batch: the 3-D centre / camera maps, the detections and the regressed parameters of the HIP
    path against the oracle's BEV forward (oracle/bev_oracle.py; the Conv3d refiners make its float32 forward ~1 s per image)."""
    import numpy as np
    from oracle import romp_oracle as O, bev_oracle as BO
    torch.set_num_threads(usable_cores())
    pick = [p for p in pick if p < images.shape[0]]
    if 'coordmap_3d' not in sd:                                  # the oracle's constant buffer (bev/model.py:9-17); the product's synthetic
        sd = dict(sd, coordmap_3d=BO.coordmap_3d())              # state_dict has the learnable tensors only
    c3d, cam3d = model.model.localization(images)
    out = model.model(images)
    torch.cuda.synchronize()
    img = images[pick].cpu()
    co, mo, _ = BO.coarse2fine_localization(sd, O.backbone_forward(sd, img))
    rep = {'images_compared': pick,
           'maps_max_abs_vs_oracle': float(max(np.abs(c3d[pick].cpu().numpy() - co.numpy()).max(), np.abs(cam3d[pick].cpu().numpy() - mo.numpy()).max()))}
    ref = BO.bev_forward(sd, img, thresh)
    if out is None or ref is None:
        rep['detections_equal'] = out is None and ref is None
        return rep
    b = out['pred_batch_ids'].cpu().numpy()
    rows = np.concatenate([np.nonzero(b == p)[0] for p in pick])

    def canon(bb, zyx):
        return np.lexsort(((zyx[:, 0] * 128 + zyx[:, 1]) * 128 + zyx[:, 2], bb))
    zyx = out['pred_czyxs'].cpu().numpy()[rows]
    ko, kr = canon(np.searchsorted(pick, b[rows]), zyx), canon(ref['pred_batch_ids'], ref['pred_czyxs'])
    same = len(rows) == len(ref['pred_batch_ids']) and np.array_equal(zyx[ko], ref['pred_czyxs'][kr])
    rep['detections_equal'] = bool(same)
    rep['persons_compared'] = int(len(rows))
    if same and len(rows):
        rep['params_pred_max_abs_vs_oracle'] = float(np.abs(out['params_pred'].cpu().numpy()[rows][ko] - ref['params_pred'][kr]).max())
    return rep


def bench_bev(args, dev):
    """BASELINE configs[3]: BEV HRNet-32 + bird's-eye-view head, 512x512, batch 32, 1 GPU (not the headline line)."""
    from romp_amd import bev, synthetic as S
    s = bev.bev_settings([])
    s.GPU, s.max_batch, s.conv_math = dev.index or 0, args.batch, args.conv_math
    sd = S.make_bev_state_dict(0)
    smpla, smil = S.make_smpl_model(0, 11), S.make_smpl_model(5, 10)
    model = bev.BEV(s, state_dict=sd, smpla_model=smpla, smil_model=smil)
    images = S.make_images(args.batch, seed=4, device=dev)
    net = model.model.net
    net.set_streams(args.streams)
    variant_table = install_variants(args, net, args.batch, 0, 'bev-hrnet32', lambda m: print('bench.py: ' + m, file=sys.stderr, flush=True))
    pads = torch.tensor([[0., 512., 0., 512., 512., 512.]]).repeat(args.batch, 1)
    # random weights have no calibrated confidence: bisect (outside the timed region) for the threshold that keeps
    # ~12 persons per image, the load the ROMP line runs at
    t_lo, t_hi = 0.0, 8.0
    for _ in range(14):
        mid = 0.5 * (t_lo + t_hi)
        model.model.centermap_parser.conf_thresh = mid
        r = model.forward_batch(images)
        kept = 0 if r is None else r['cam'].shape[0] / args.batch
        if 10.0 <= kept <= 14.0:
            break
        t_lo, t_hi = (mid, t_hi) if kept > 14.0 else (t_lo, mid)
    n = 0
    for _ in range(args.warmup):
        r = model.forward_batch(images, pads)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = model.forward_batch(images, pads)
        n = 0 if r is None else r['cam'].shape[0]
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    res = {'metric': 'images/sec (512x512, BEV HRNet-32)', 'value': round(args.batch * args.steps / dt, 2),
           'unit': 'images/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f32' if args.conv_math == 'f32' else 'f32 (%s-split conv products)' % args.conv_math, 'data': 'synthetic',
           'config': {'workload': 'BEV HRNet-32 + BEV head 512x512 batch=%d (BASELINE configs[3]); net+3D parse+'
                                  'regression+SMPL-A+post-processing' % args.batch,
                      'persons_kept_per_image': round(n / args.batch, 2), 'center_thresh': round(mid, 4), 'variant_table': variant_table}}
    if args.dump_op_kernels:
        json.dump({'batch': args.batch, 'names': net.variant_names(args.batch), 'op_names': list(net.program.names),
                   'kinds': [int(o.kind) for o in net.program.ops],
                   'bytes': [float(b) * args.batch for b in net.program.bytes], 'flops': [float(f) * args.batch for f in net.program.flops]},
                  open(args.dump_op_kernels, 'w'))
    if not args.no_roofline:
        s1 = torch.cuda.Stream(dev)
        with torch.cuda.stream(s1):
            roof, classes = roofline_report(net, images, pmc_workload='_bev' if args.batch == 32 else None)
            ms = net.profile(images, iters=2)
        head_ms = sum(t for t, nm in zip(ms, net.program.names) if nm.startswith('bev.'))
        c3 = [(t, by) for t, nm, by in zip(ms, net.program.names, net.program.bytes) if 'refine' in nm]
        roof['head_ms_per_batch'] = round(head_ms, 3)
        if c3:
            gbs = sum(by for _, by in c3) * args.batch / (sum(t for t, _ in c3) * 1e-3) / 1e9
            roof['conv3d_refiners'] = {'launches': len(c3), 'ms': round(sum(t for t, _ in c3), 4), 'hbm_gbs': round(gbs, 1),
                                       'hbm_frac': round(gbs / PEAK_HBM_GBS, 4)}
        res['roofline'] = roof
        res['kernel_classes'] = classes
    if not args.no_parity:
        res['config'].update(bev_parity_report(model, images, sd, mid))
    if not args.no_cpu_baseline:
        from oracle import bev_oracle as BO, ref_cpu
        torch.set_num_threads(usable_cores())
        img1 = images[:1].cpu()
        sd = dict(sd)
        if ref_cpu.bev_available():      # the reference's own BEVv1 (oracle/_ref/bev/model.pyc: simple_romp/bev/model.py:104-250), PyTorch-CPU
            ref = ref_cpu.ReferenceBev(sd, mid)
            run, kind = (lambda: ref(img1.clone())), 'reference'
            what = ('the reference itself: BEVv1.forward (simple_romp/bev/model.py:232-250: network + 3-D parse + regression), PyTorch-CPU '
                    'float32, byte-compiled from /root/reference into oracle/_ref/bev')
        else:
            sd['coordmap_3d'] = BO.coordmap_3d()                 # the oracle's constant buffer (bev/model.py:9-17)
            run, kind = (lambda: BO.bev_forward(sd, img1, mid)), 'port'
            what = 'the torch-CPU oracle restatement (network + 3-D parse + regression; oracle/_ref/bev is not staged on this box)'
        t0 = time.time()
        run()
        warm = time.time() - t0
        t0, k = time.time(), 0
        while warm < args.cpu_seconds / 2 and time.time() - t0 < args.cpu_seconds and k < 8:
            run(); k += 1
        cdt = (time.time() - t0) if k else warm
        res['cpu_baseline'] = dict(value=round(max(k, 1) / cdt, 3), unit='images/s', cores=torch.get_num_threads(), kind=kind,
                                   sample='%d single-image BEV forwards of %s' % (max(k, 1), what))
    print(json.dumps(res), flush=True)



# ------------------------------------------------------------------------------------------------ clocks / power / per-step times
class GpuSensors:
    """Engine clock (MHz) and socket power (W) of one device, sampled by a background thread while the timed region runs -- the line
    then says at what clock the number was taken (this job is power-limited: 2.2 GHz of the nominal 2.4 at 1.3 kW of the 1.4-kW cap,
    profiles/r05_clocks_power.txt).  Sources, first that answers: the amdgpu hwmon files of the device's PCI function
    (freq1_input = current sclk in Hz, power1_average / power1_input in microwatts: two file reads per sample, no subprocess),
    else `rocm-smi --showclocks --showpower` (the sampler of scripts/gpu_clocks.sh).  No source -> empty summaries, never an error."""

    def __init__(self, device_index, period=0.1):
        import glob
        import threading
        self.period, self.samples, self._stop, self._thread = period, [], threading.Event(), None
        self.index, self.source, self._files, self.static = device_index, None, None, {}
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            hw = sorted(glob.glob('/sys/bus/pci/devices/%s/hwmon/hwmon*' % bdf))
            if hw:
                fr = os.path.join(hw[0], 'freq1_input')
                pw = [f for f in (os.path.join(hw[0], 'power1_average'), os.path.join(hw[0], 'power1_input')) if os.path.exists(f)]
                if os.path.exists(fr) and self._read(fr) is not None:
                    self._files = (fr, pw[0] if pw and self._read(pw[0]) is not None else None)
                    self.source = 'sysfs hwmon (%s)' % bdf
                    # read once, outside the sampling loop: memory clock, hottest sensor, the power cap the job runs into
                    mc, cap = self._read(os.path.join(hw[0], 'freq2_input')), self._read(os.path.join(hw[0], 'power1_cap'))
                    temps = [self._read(f) for f in glob.glob(os.path.join(hw[0], 'temp*_input'))]
                    self.static = {'mclk_mhz': None if mc is None else round(mc / 1e6), 'power_cap_w': None if cap is None else round(cap / 1e6),
                                   'temp_c_at_start': max([t / 1e3 for t in temps if t is not None], default=None)}
        except Exception:                              # noqa: BLE001 -- a sensor is optional
            self._files = None
        if self._files is None:
            import shutil
            if shutil.which('rocm-smi'):
                self.source, self.period = 'rocm-smi --showclocks --showpower', max(period, 0.25)

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except Exception:                              # noqa: BLE001
            return None

    def _sample(self):
        if self._files is not None:
            fr, pw = self._files
            hz = self._read(fr)
            uw = self._read(pw) if pw else None
            return (None if hz is None else hz / 1e6, None if uw is None else uw / 1e6)
        import re
        import subprocess
        try:
            out = subprocess.run(['rocm-smi', '-d', str(self.index), '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
        except Exception:                              # noqa: BLE001
            return (None, None)
        sc = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', out)
        pw = re.search(r'Power \(W\): ([0-9.]+)', out)
        return (float(sc.group(1)) if sc else None, float(pw.group(1)) if pw else None)

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(self._sample())
            self._stop.wait(self.period)

    def start(self):
        import threading
        if self.source is None:
            return self
        self.samples, self._stop = [], threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=6)
            self._thread = None
        return self.summary()

    def summary(self):
        out = {'clock_mhz': None, 'power_w': None, 'sensor_source': self.source, 'sensor_samples': len(self.samples)}
        out.update(self.static)
        for key, col in (('clock_mhz', 0), ('power_w', 1)):
            v = sorted(s[col] for s in self.samples if s[col] is not None)
            if v:
                out[key] = {'median': round(v[len(v) // 2], 1), 'min': round(v[0], 1), 'max': round(v[-1], 1)}
        return out


def step_stats(ms):
    v = sorted(ms)
    return {'min': round(v[0], 3), 'median': round(v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2]), 3),
            'max': round(v[-1], 3), 'all': [round(x, 3) for x in ms]}


# ------------------------------------------------------------------------------------------------ headline
def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start N ranks of this same command under
    torch.distributed.run on this node (rendezvous on 127.0.0.1) and hand back its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def preheat(step, dev, world, cap_s=10.0, tol=0.01, max_steps=40):
    """UNcounted steps in front of the warm-up until two consecutive ones agree within `tol` (cap `cap_s` seconds): a fresh lease
    runs its first steps slowly (clock ramp, first-touch page tables, code-object loads -- profiles/r05_notes.md section 8 saw 15-20 %
    on every kernel of the sweep that ran first), and the driver's bench IS the first thing a fresh box runs.  Every rank takes the
    same decision (the step times are MAX-reduced first).  -> (seconds spent, [ms of every pre-heat step])."""
    sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == 'cuda' else (lambda: None)
    ms, t_start = [], time.perf_counter()
    spent = 0.0
    while len(ms) < max_steps:
        sync()
        t0 = time.perf_counter()
        step()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        ms.append(dt * 1e3)
        spent += dt
        if (len(ms) >= 2 and abs(ms[-1] - ms[-2]) <= tol * ms[-1]) or spent >= cap_s:
            break
    return time.perf_counter() - t_start, ms


def timed_steps(step, warmup, steps, dev, world, sensors=None):
    """The contract's timing discipline: `warmup` untimed steps, then exactly `steps` steps bracketed by a barrier and a device
    synchronisation on both sides; the MAX over ranks is the job's time.  Inside the bracket nothing is added but one event record
    per step on the launch stream (and a host clock read): -> (seconds, {'host': [...], 'gpu': [...]} per-step milliseconds of this
    rank -- host = between the returns of consecutive step() calls, gpu = between the events; a step's last SMPL launch may still
    be running when step() returns, so single entries can trade a fraction of a millisecond with their neighbour; the sums are the
    bracket's time).  `sensors` (GpuSensors) sample clock / power during exactly the bracket."""
    cuda = dev.type == 'cuda'
    sync = (lambda: torch.cuda.synchronize(dev)) if cuda else (lambda: None)
    for _ in range(warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if cuda else None
    host = [0.0] * (steps + 1)
    if sensors is not None:
        sensors.start()
    t0 = time.perf_counter()
    host[0] = t0
    if cuda:
        ev[0].record()
    for i in range(steps):
        step()
        if cuda:
            ev[i + 1].record()
        host[i + 1] = time.perf_counter()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if sensors is not None:
        sensors.stop()
    per = {'host': [(host[i + 1] - host[i]) * 1e3 for i in range(steps)]}
    if cuda:
        per['gpu'] = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, per


def collective_info(world, dev):
    """What the result line says about the fabric: ranks in the process group, backend, RCCL version."""
    info = {'rccl_ranks': world, 'backend': 'none (single process)' if world == 1 else dist.get_backend()}
    if world > 1:
        assert dist.get_world_size() == world
    if dev.type == 'cuda':
        try:
            v = torch.cuda.nccl.version()
            info['rccl_version'] = '.'.join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception as e:                               # noqa: BLE001 -- informational only
            info['rccl_version'] = 'unknown (%s)' % type(e).__name__
    return info


def run_job(args, model, images, lo, rank, world, dev, D, sensors=None):
    """The timed job of the headline line: every rank walks its resident shard in calls of --batch images (net + parse + SMPL);
    N > 1: one all-gather of the per-person records per step.  -> (seconds for args.steps steps (max over ranks), persons,
    timing record: per-step times, the pre-heat's steps)."""
    B = args.batch
    persons = [0]
    # steady state of a stream of identical jobs (round 6): the shard is walked again next step, so the chunk pipeline stays primed
    # across steps (the next step's first network under this step's last parse + SMPL + all-gather) and the record exchange keeps
    # its capacity (counts inside the one all-gather).  Every step still launches exactly its own number of networks.
    nxt = images if getattr(args, 'cross_step', 1) else None
    gather_state = {} if getattr(args, 'cross_step', 1) else None

    def step():
        if world > 1:
            out, counts = D.sharded_forward(model, images, lo, with_joints=True, with_verts=bool(args.with_verts), chunk=B, next_images=nxt,
                                            gather_state=gather_state)
            persons[0] = sum(counts)
        else:
            rec = D.local_records(model, images, lo, chunk=B, with_joints=True, with_verts=bool(args.with_verts), next_images=nxt)
            persons[0] = 0 if rec is None else rec.shape[0]
    pre_s, pre_ms = preheat(step, dev, world, cap_s=getattr(args, 'preheat_cap', 10.0)) if getattr(args, 'preheat_cap', 10.0) > 0 else (0.0, [])
    dt, per = timed_steps(step, args.warmup, args.steps, dev, world, sensors)
    pipe = getattr(model, '_pipe', None)
    if pipe is not None:
        pipe['primed'] = None            # (the last step announced a step that will not come: timed_steps has synchronised, nothing is in flight)
    timing = {'preheat_s': round(pre_s, 3), 'preheat_step_ms': [round(x, 3) for x in pre_ms],
              'step_ms': step_stats(per.get('gpu', per['host'])), 'step_ms_clock': 'HIP events on the launch stream' if 'gpu' in per else 'host',
              'step_ms_host': step_stats(per['host'])}
    return dt, persons[0], timing


def headline_result(args, dt, persons, G, n_local, world, dev, variant_table, timing=None, sensors=None):
    B = args.batch
    strong = args.global_batch > 0
    bb = 'HRNet-32' if args.backbone == 'hrnet32' else 'ResNet-50'
    if strong:
        workload = ('ROMP %s 512x512, batch=%d synthetic images sharded across %d GPU%s (BASELINE configs[2]), each shard walked in forward calls of '
                    'batch=%d (BASELINE configs[1]); net+parse+SMPL per call%s' %
                    (bb, G, world, '' if world == 1 else 's', B, ', one RCCL all-gather of the per-person records per step' if world > 1 else ''))
    else:
        workload = 'ROMP %s 512x512, batch=%d synthetic images per GPU per step (weak scaling); net+parse+SMPL%s' % (
            bb, B, '+RCCL all-gather of per-person records' if world > 1 else '')
    cfg = {'workload': workload, 'batch_per_call': B, 'global_batch': G, 'images_per_gpu_per_step': n_local,
           'ms_per_call': round(dt / args.steps / max(1, -(-n_local // B)) * 1e3, 3),
           'persons_per_image': round(persons / G, 2), 'center_thresh': args.center_thresh, 'hipgraph': bool(args.graph),
           'autotune': bool(args.autotune), 'variant_table': variant_table, 'branch_streams': bool(args.streams),
           'conv_math': args.conv_math, 'parallelism': 'dp%d' % world, 'cross_step_pipeline': bool(getattr(args, 'cross_step', 1))}
    cfg.update(collective_info(world, dev))
    res = {
        'metric': 'images/sec (512x512, %s)' % bb, 'value': round(G * args.steps / dt, 2), 'unit': 'images/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
        'dtype': 'f32' if args.conv_math == 'f32' else 'f32 (convs as %s-split products, f32 accumulate; same 1e-4 parity gate as f32 MFMA)' % args.conv_math,
        'data': 'synthetic', 'config': cfg,
    }
    # what the number was taken under (rank 0's view): every timed step's duration, the un-counted pre-heat in front of the warm-up
    # (its step times show a cold box warming up), engine clock / socket power sampled during exactly the timed steps
    if timing is not None:
        res.update(step_ms=timing['step_ms'], preheat_s=timing['preheat_s'], preheat_step_ms=timing['preheat_step_ms'])
        res['step_ms']['clock'] = timing['step_ms_clock']
        res['step_ms_host'] = timing['step_ms_host']
    if sensors is not None:
        res.update(sensors.summary())
    return res


def install_variants(args, net, B, rank, backbone, log):
    """One variant table for everything measured: the committed table of this configuration (or --tune-file) if it resolves
    against this build, else autotune (and say so).  -> the string reported as config.variant_table."""
    from romp_amd import tuning as T
    path = args.tune_file
    if path is None:
        path = T.default_table_path(backbone, args.conv_math, B)
        committed = True
    else:
        committed = False
    if path and path != 'none':
        ok, why = T.install_table(net, B, path)
        if ok:
            return ('committed ' if committed else 'file ') + os.path.relpath(path, ROOT)
        if os.path.exists(path):
            log('variant table %s not usable (%s): autotuning' % (path, why))
    if args.autotune:
        net.autotune(B)
        if path and path != 'none' and not committed and rank == 0:
            T.save_table(net, B, path, note='bench.py autotune')
        return 'autotuned in this run'
    return 'heuristic'


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    if world != args.gpus:
        print('bench.py: --gpus %d but the launcher started %d rank%s: reporting n_gpus=%d' % (args.gpus, world, '' if world == 1 else 's', world),
              file=sys.stderr, flush=True)

    import romp_amd
    from romp_amd import lib as L, synthetic as S, distributed as D
    lib = L.load()
    if args.workload == 'bev':
        return bench_bev(args, dev)
    if args.workload == 'smpl':
        return bench_smpl(args, dev)
    B = args.batch
    settings = romp_amd.romp_settings([])
    settings.GPU, settings.center_thresh, settings.max_batch = local_rank, args.center_thresh, B
    settings.conv_math, settings.backbone = args.conv_math, args.backbone
    if args.backbone == 'resnet50':
        sd = S.make_resnet_state_dict(0, center_bias=2.0)
    else:
        sd = S.make_romp_state_dict(0)
    smpl_model = S.make_smpl_model(0)
    model = romp_amd.ROMP(settings, state_dict=sd, smpl_model=smpl_model)
    model.model.set_streams(args.streams)
    variant_table = install_variants(args, model.model, B, rank, args.backbone, lambda m: print('bench.py: ' + m, file=sys.stderr, flush=True))
    if args.graph:
        model.model.set_graph(True)
    strong = args.global_batch > 0
    G = args.global_batch if strong else B * world
    lo, hi = D.shard_range(G, rank, world)
    n_local = hi - lo
    # the shard, resident in HBM: seeded uint8 noise generated on the device chunk by chunk (1024 x 3.1 MB of float32)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    images = torch.empty(n_local, 512, 512, 3, device=dev, dtype=torch.float32)
    for c0 in range(0, n_local, 64):
        images[c0:c0 + 64] = torch.randint(0, 256, images[c0:c0 + 64].shape, device=dev, generator=gen, dtype=torch.uint8).float()
    stream = torch.cuda.Stream(dev)
    if args.backbone == 'resnet50':
        # random weights have no calibrated confidence: bisect (outside the timed region) for the threshold that keeps
        # ~12 persons per image, the load the HRNet-32 line runs at
        t_lo, t_hi = -8.0, 16.0
        with torch.cuda.stream(stream):
            for _ in range(16):
                mid = 0.5 * (t_lo + t_hi)
                model.centermap_parser.conf_thresh = mid
                out, _ = model.forward_batch(images[:B])
                kept = 0 if out is None else out['cam'].shape[0] / min(B, n_local)
                if 10.0 <= kept <= 14.0:
                    break
                t_lo, t_hi = (mid, t_hi) if kept > 14.0 else (t_lo, mid)
        args.center_thresh = round(mid, 4)

    sensors = GpuSensors(local_rank) if rank == 0 else None
    with torch.cuda.stream(stream):
        dt, persons, timing = run_job(args, model, images, lo, rank, world, dev, D, sensors)
    result = headline_result(args, dt, persons, G, n_local, world, dev, variant_table, timing, sensors)
    if rank == 0:
        first = images[:B]
        if args.dump_op_kernels:
            net = model.model
            json.dump({'batch': B, 'names': net.variant_names(B), 'op_names': list(net.program.names),
                       'kinds': [int(o.kind) for o in net.program.ops],
                       'bytes': [float(b) * B for b in net.program.bytes], 'flops': [float(f) * B for f in net.program.flops]},
                      open(args.dump_op_kernels, 'w'))
        if not args.no_roofline:
            with torch.cuda.stream(stream):
                roof, classes = roofline_report(model.model, first, pmc_workload=(('' if B == 32 else '_b%d' % B) if args.backbone == 'hrnet32' else
                                                                                  ('_' + args.backbone if B == 32 else None)))
            result['roofline'] = roof
            result['kernel_classes'] = classes
        if not args.no_parity:
            with torch.cuda.stream(stream):
                net_oracle = None
                if args.backbone == 'resnet50':
                    from oracle import resnet_oracle as RO
                    net_oracle = RO.resnet_romp_forward
                result['config'].update(parity_report(model, first, sd, smpl_model, args.center_thresh, net_oracle=net_oracle))
        if world == 1 and not args.no_end_to_end:
            result['end_to_end'] = end_to_end(model, lib, L, dev, B, max(8, min(32, n_local // B)), stream)
        if world == 1 and args.conv_math != 'f32' and args.backbone == 'hrnet32' and not args.no_f32_companion:
            # the same job with every conv on the exact-f32 MFMA kernels only, for readers who do not accept the split
            # arithmetic as float32 (it passes the same 1e-4 gate): same calls, same timing discipline, fewer steps
            del model
            settings.conv_math = 'f32'
            m32 = romp_amd.ROMP(settings, state_dict=sd, smpl_model=smpl_model)
            m32.model.set_streams(args.streams)
            if args.autotune:
                m32.model.autotune(B)
            if args.graph:
                m32.model.set_graph(True)
            k32 = max(1, min(3, args.steps))
            with torch.cuda.stream(stream):
                D.local_records(m32, images, lo, chunk=B)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(k32):
                    D.local_records(m32, images, lo, chunk=B)
                torch.cuda.synchronize(dev)
                dt32 = time.perf_counter() - t0
            result['f32_mfma_companion'] = {'value': round(G * k32 / dt32, 2), 'unit': 'images/s', 'dtype': 'f32', 'steps': k32,
                                            'ms_per_step': round(dt32 / k32 * 1e3, 3),
                                            'note': 'same job, conv_math=f32 (v_mfma_f32_32x32x2_f32 only)'}
            del m32
        if world == 1 and args.backbone == 'hrnet32' and not args.no_latency:
            result['single_image_latency'] = single_image_latency(sd, smpl_model, args, dev, stream)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(sd, smpl_model, args.center_thresh, args.cpu_seconds, args.backbone)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
