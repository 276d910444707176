"""World-size-2 test of the multi-GPU result exchange on CPU (gloo): the all-gather-v of the
per-person records must reproduce the single-process ordering (rank-major == image order),
including a rank with zero detections and uneven counts."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_outputs(n, seed):
    g = torch.Generator().manual_seed(seed)
    flat = torch.randint(0, 4096, (n,), generator=g)
    return {
        'cam': torch.randn(n, 3, generator=g), 'smpl_thetas': torch.randn(n, 72, generator=g),
        'smpl_betas': torch.randn(n, 10, generator=g), 'center_confs': torch.rand(n, 1, generator=g),
        'center_preds': torch.stack([flat % 64, flat // 64], 1) * 8, 'joints': torch.randn(n, 71, 3, generator=g),
    }, torch.sort(torch.randint(0, 4, (n,), generator=g))[0]


def _worker(rank, world, port, counts, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from romp_amd import distributed as D
    lo, hi = D.shard_range(8, rank, world)
    n = counts[rank]
    rec = None
    if n:
        out, bids = _fake_outputs(n, 100 + rank)
        rec = D.pack_records(out, bids, lo, with_joints=True)
    allrec, got = D.all_gather_records(rec, D.record_width(True), torch.device('cpu'))
    un = D.unpack_records(allrec, with_joints=True)
    q.put((rank, got, un['image_ids'].tolist(), un['smpl_thetas'].sum().item(), tuple(un['joints'].shape)))
    dist.barrier()
    dist.destroy_process_group()


def _run(counts):
    world = len(counts)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, counts, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return sorted(res)


def _expected(counts):
    sys.path.insert(0, ROOT)
    from romp_amd import distributed as D
    ids, tsum = [], 0.0
    for r, n in enumerate(counts):
        if n:
            out, bids = _fake_outputs(n, 100 + r)
            lo, _ = D.shard_range(8, r, len(counts))
            ids += (bids + lo).tolist()
            tsum += out['smpl_thetas'].sum().item()
    return ids, tsum


def test_allgather_records_world2_uneven():
    for counts in ([5, 3], [0, 4], [7, 0]):
        res = _run(counts)
        ids, tsum = _expected(counts)
        for rank, got, image_ids, s, jshape in res:
            assert got == counts
            assert image_ids == ids
            assert abs(s - tsum) < 1e-3
            assert jshape == (sum(counts), 71, 3)


def test_allgather_records_nobody_anywhere():
    res = _run([0, 0])
    for rank, got, image_ids, s, jshape in res:
        assert got == [0, 0] and image_ids == [] and jshape == (0, 71, 3)


class _StubModel:
    """forward_batch of a fake ROMP: image b (whose pixel [0,0,0] holds its global id v) yields v % 3 persons whose thetas are
    all v -- enough to check ids / order / chunking without a GPU."""

    def forward_batch(self, images):
        ids = images[:, 0, 0, 0].long()
        bids = torch.cat([torch.full((int(v) % 3,), b, dtype=torch.int64) for b, v in enumerate(ids.tolist())] or [torch.zeros(0, dtype=torch.int64)])
        n = bids.numel()
        if n == 0:
            return None, None
        v = ids[bids].float()
        out = {'cam': v.view(n, 1).repeat(1, 3), 'smpl_thetas': v.view(n, 1).repeat(1, 72), 'smpl_betas': torch.zeros(n, 10),
               'center_confs': torch.ones(n, 1), 'center_preds': torch.zeros(n, 2, dtype=torch.int64), 'joints': torch.zeros(n, 71, 3)}
        return out, bids


class _StubPipelinedModel(_StubModel):
    """The stub with ROMP.forward_chunks' interface, including the cross-call priming protocol of round 6: `next_images` names the
    tensor the next call will walk; its first chunk is then taken from the primed slot (here: computed eagerly when it was
    announced).  `primed_used` counts the calls that started from a primed first chunk."""

    def __init__(self):
        self.primed, self.primed_used = None, 0

    def forward_chunks(self, images, chunk, next_images=None):
        starts = list(range(0, images.shape[0], chunk))
        first = None
        if self.primed is not None and self.primed[0] == (images.data_ptr(), tuple(images.shape), chunk):
            first = self.primed[1]
            self.primed_used += 1
        self.primed = None
        for i, c0 in enumerate(starts):
            out, bids = first if (i == 0 and first is not None) else self.forward_batch(images[c0:c0 + chunk])
            if i + 1 == len(starts) and next_images is not None:
                self.primed = ((next_images.data_ptr(), tuple(next_images.shape), chunk), self.forward_batch(next_images[:chunk]))
            yield out, bids, c0


def _chunk_worker(rank, world, port, n_images, chunk, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from romp_amd import distributed as D
    lo, hi = D.shard_range(n_images, rank, world)
    images = torch.zeros(hi - lo, 2, 2, 3)
    images[:, 0, 0, 0] = torch.arange(lo, hi).float()
    out, counts = D.sharded_forward(_StubModel(), images, lo, with_joints=True, chunk=chunk)
    q.put((rank, counts, out['image_ids'].tolist(), out['smpl_thetas'][:, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_forward_chunked_global_ids():
    """The north-star job shape in miniature: a global batch sharded contiguously over 2 ranks, each shard walked in chunks
    (also a ragged last chunk), ONE all-gather per step: every rank ends with all persons in global image order, ids global."""
    n_images, world = 11, 2
    for chunk in (2, 3, None):
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_chunk_worker, args=(r, world, port, n_images, chunk, q)) for r in range(world)]
        for p in ps:
            p.start()
        res = [q.get(timeout=120) for _ in ps]
        for p in ps:
            p.join(60)
            assert p.exitcode == 0
        want = [v for v in range(n_images) for _ in range(v % 3)]
        for rank, counts, ids, th in res:
            assert ids == want and th == [float(v) for v in want]
            assert sum(counts) == len(want)


def _bench_worker(rank, world, port, q):
    """bench.py's own job loop (run_job -> timed_steps -> headline_result) on 2 gloo ranks with the stub model."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    from romp_amd import distributed as D
    sys.argv = ['bench.py', '--gpus', str(world), '--steps', '3', '--warmup', '1', '--batch', '4', '--global-batch', '22']
    args = bench.parse_args()
    dev = torch.device('cpu')
    G = args.global_batch
    lo, hi = D.shard_range(G, rank, world)
    images = torch.zeros(hi - lo, 2, 2, 3)
    images[:, 0, 0, 0] = torch.arange(lo, hi).float()
    dt, persons, timing = bench.run_job(args, _StubModel(), images, lo, rank, world, dev, D)
    res = bench.headline_result(args, dt, persons, G, hi - lo, world, dev, 'stub', timing)
    # the cross-step form (round 6: pipeline primed across steps, fixed-capacity record exchange) against the plain one: the same
    # records, step after step, and the protocol really ran (every step but the first starts from a primed chunk; one counted
    # exchange, then one all-gather per step)
    pm = _StubPipelinedModel()
    state = {}
    plain, _ = D.sharded_forward(_StubModel(), images, lo, with_joints=True, chunk=args.batch)
    same = True
    for _ in range(4):
        got, counts = D.sharded_forward(pm, images, lo, with_joints=True, chunk=args.batch, next_images=images, gather_state=state)
        same = same and all(torch.equal(got[k], plain[k]) for k in plain)
    res['_cross_step'] = dict(same=bool(same), primed_used=pm.primed_used, exchanges=state.get('exchanges', 0), cap=state.get('cap', 0),
                              regrown=state.get('regrown', 0))
    # a step with MORE persons than the capacity: every rank regrows together and the result is still complete
    state['cap'] = 1
    got, counts = D.sharded_forward(_StubModel(), images, lo, with_joints=True, chunk=args.batch, gather_state=state)
    res['_cross_step']['regrow_ok'] = bool(all(torch.equal(got[k], plain[k]) for k in plain)) and state['cap'] >= max(counts) and state.get('regrown', 0) == 1
    q.put((rank, res, persons, dt))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_job_loop_world2():
    """the 8-GPU line must not die on launch.  The real control flow of bench.py -- shard, walk in calls of
    --batch, all-gather once per step, barrier-bracketed timing, MAX over ranks, the result line -- on world size 2 (gloo)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want_persons = sum(v % 3 for v in range(22))
    dts = {round(r[3], 9) for r in res}
    assert len(dts) == 1, 'every rank must report the same (max-over-ranks) time'
    for rank, line, persons, dt in res:
        assert persons == want_persons
        assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['steps'] == 3 and line['warmup'] == 1
        assert line['config']['rccl_ranks'] == 2 and line['config']['backend'] == 'gloo' and line['config']['parallelism'] == 'dp2'
        assert line['config']['global_batch'] == 22 and line['config']['images_per_gpu_per_step'] == 11
        assert abs(line['value'] - 22 * 3 / dt) < 0.01 * line['value']
        assert abs(line['config']['persons_per_image'] - round(want_persons / 22, 2)) < 1e-9
        # the line says what it was taken under: one duration per timed step (exactly --steps of them), the un-counted pre-heat
        assert len(line['step_ms']['all']) == 3 and line['step_ms']['min'] <= line['step_ms']['median'] <= line['step_ms']['max']
        assert 2 <= len(line['preheat_step_ms']) <= 40 and line['preheat_s'] > 0
        assert line['config']['cross_step_pipeline'] is True
        cs = line['_cross_step']
        assert cs['same'] and cs['primed_used'] == 3 and cs['exchanges'] == 3 and cs['cap'] >= 64 and cs['regrow_ok'], cs
    assert len({len(r[1]['preheat_step_ms']) for r in res}) == 1, 'every rank must take the same pre-heat decision'


def test_bench_respawns_itself_for_multi_gpu(monkeypatch):
    """`python bench.py --gpus 8` without a launcher environment re-executes under torch.distributed.run instead of asserting."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 0
    import subprocess
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '2'])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-4:] == ['--gpus', '8', '--steps', '2']
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'
