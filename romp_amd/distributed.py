"""Image-batch data parallelism: one process per GPU, contiguous static shards, one RCCL
all-gather of compact per-person records over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` wrapper (simple_romp/romp/main.py:77),
which scatters the batch, replicates the module per call and gathers the DENSE maps
((B,1,64,64)+(B,145,64,64) = 2.39 MB/image) to GPU 0 before parsing.  Here each rank runs
net -> parse -> SMPL on its own shard and only per-person records cross the fabric:
record = [global image id, flat index, confidence, cam 3, thetas 72, betas 10] = 88 floats
(352 B/person), optionally + joints (71x3) and vertices (6890x3).  Image ids and flat indices travel as float32 values: exact
below 2^24 = 16.7 M images per job (the north-star job has 1 024).

RCCL has no all-gather-v: ranks first all-gather their counts, pad their block to the
maximum count, all-gather the padded blocks and strip the padding.  Images are independent
(BN uses running statistics), so there is no other data-path collective.
"""
import torch
import torch.distributed as dist

RECORD_BASE = 88          # img id, flat ind, conf, cam(3), thetas(72), betas(10)


def shard_range(n_items, rank, world_size):
    """Contiguous shard [lo, hi) of rank; the first (n % world) ranks get one extra item."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(outputs, batch_ids, img_offset, flat_inds=None, with_joints=False, with_verts=False):
    """Per-person records of the local shard as one float32 matrix (N, R)."""
    if outputs is None:
        return None
    N = outputs['cam'].shape[0]
    dev = outputs['cam'].device
    flat = flat_inds if flat_inds is not None else (outputs['center_preds'][:, 1] // 8 * 64 + outputs['center_preds'][:, 0] // 8)
    cols = [(batch_ids + img_offset).float().view(N, 1), flat.float().view(N, 1), outputs['center_confs'].view(N, 1),
            outputs['cam'], outputs['smpl_thetas'], outputs['smpl_betas']]
    if with_joints:
        cols.append(outputs['joints'].reshape(N, -1))
    if with_verts:
        cols.append(outputs['verts'].reshape(N, -1))
    return torch.cat([c.to(dev).float() for c in cols], 1).contiguous()


def record_width(with_joints=False, with_verts=False):
    return RECORD_BASE + (213 if with_joints else 0) + (6890 * 3 if with_verts else 0)


def all_gather_records(local, width, device, group=None, state=None):
    """All-gather-v of (N_r, width) float32 blocks -> (sum N_r, width), rank-major (= image order
    because shards are contiguous).  `local` may be None (no detections on this rank).
    `state` (a dict the caller keeps between calls; None: the two-exchange form): FIXED-CAPACITY form (round 6).  Every rank contributes
    `state['cap']` rows plus ONE header row that carries its row count (a float32: exact below 2^24), so the counts travel
    inside the one all-gather and the host reads them AFTER it -- one synchronisation for the whole exchange instead of a count
    all-gather + host sync in front of the payload.  The capacity starts from a first counted exchange (1.25 x the largest count,
    at least 64 rows) and grows the same way whenever the headers show a rank that did not fit: every rank sees the same headers
    and repeats the exchange once, together."""
    world = dist.get_world_size(group)
    n_local = 0 if local is None else local.shape[0]
    if state is not None and state.get('cap', 0) > 0:
        cap = state['cap']
        block = torch.zeros(cap + 1, width, device=device, dtype=torch.float32)
        block[0, 0] = float(n_local)
        if n_local:
            block[1:1 + min(n_local, cap)] = local[:cap]
        gathered = torch.empty(world * (cap + 1), width, device=device, dtype=torch.float32)
        _all_gather_block(gathered, block, world, device, group)
        counts = [int(v) for v in gathered.view(world, cap + 1, width)[:, 0, 0].tolist()]      # the exchange's one host sync
        if max(counts) <= cap:
            state['exchanges'] = state.get('exchanges', 0) + 1
            parts = [gathered[r * (cap + 1) + 1: r * (cap + 1) + 1 + counts[r]] for r in range(world)]
            return (torch.cat(parts, 0) if sum(counts) else torch.zeros(0, width, device=device)), counts
        state['cap'] = _capacity(max(counts))                              # (every rank takes this branch together)
        state['regrown'] = state.get('regrown', 0) + 1
        return all_gather_records(local, width, device, group, state)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n_local], dtype=torch.int64, device=device), group=group)
    counts = torch.cat(counts).tolist()                                 # one host sync for all ranks' counts
    n_max = max(counts)
    if state is not None:
        state['cap'] = _capacity(n_max)
    if n_max == 0:
        return torch.zeros(0, width, device=device), counts
    block = torch.zeros(n_max, width, device=device, dtype=torch.float32)
    if n_local:
        block[:n_local] = local
    gathered = torch.empty(world * n_max, width, device=device, dtype=torch.float32)
    _all_gather_block(gathered, block, world, device, group)
    parts = [gathered[r * n_max: r * n_max + counts[r]] for r in range(world)]
    return torch.cat(parts, 0), counts


def _capacity(n_max):
    return max(64, -(-int(n_max) * 5 // 4))


def _all_gather_block(gathered, block, world, device, group):
    if device.type == 'cuda' and hasattr(dist, 'all_gather_into_tensor'):
        dist.all_gather_into_tensor(gathered, block, group=group)      # one RCCL all-gather
    else:
        _all_gather_list(gathered, block, world, group)                # gloo (CPU tests)


def _all_gather_list(gathered, block, world, group):
    outs = list(gathered.chunk(world, 0))
    dist.all_gather(outs, block, group=group)


def unpack_records(rec, with_joints=False, with_verts=False):
    out = {
        'image_ids': rec[:, 0].long(), 'flat_inds': rec[:, 1].long(), 'center_confs': rec[:, 2:3],
        'cam': rec[:, 3:6], 'smpl_thetas': rec[:, 6:78], 'smpl_betas': rec[:, 78:88],
    }
    o = RECORD_BASE
    if with_joints:
        out['joints'] = rec[:, o:o + 213].reshape(-1, 71, 3)
        o += 213
    if with_verts:
        out['verts'] = rec[:, o:o + 6890 * 3].reshape(-1, 6890, 3)
    return out


def local_records(model, images_local, img_offset, chunk=None, with_joints=True, with_verts=False, next_images=None):
    """net -> parse -> SMPL over this rank's shard, walked in chunks of `chunk` images (the batch size the context was
    built and tuned for; None: one call), as one (N, R) record matrix with GLOBAL image ids (shard offset + position in the
    shard: the rule of the reference's training tree, romp/lib/maps_utils/result_parser.py:280-282), or None.
    `next_images`: the shard the NEXT call will walk (ROMP.forward_chunks keeps its pipeline primed across calls: the next shard's
    first network runs under this shard's last parse + SMPL and under the all-gather that follows)."""
    n = images_local.shape[0]
    chunk = n if not chunk else int(chunk)
    recs = []
    if chunk < n and hasattr(model, 'forward_chunks'):              # pipelined: network of chunk i+1 under parse + SMPL of chunk i
        it = model.forward_chunks(images_local, chunk, next_images=next_images) if next_images is not None else model.forward_chunks(images_local, chunk)
    else:
        it = ((*model.forward_batch(images_local[c0:c0 + chunk]), c0) for c0 in range(0, n, chunk))
    for outputs, batch_ids, c0 in it:
        r = pack_records(outputs, batch_ids, img_offset + c0, with_joints=with_joints, with_verts=with_verts)
        if r is not None:
            recs.append(r)
    if not recs:
        return None
    return recs[0] if len(recs) == 1 else torch.cat(recs, 0)


def sharded_forward(model, images_local, img_offset, with_joints=True, with_verts=False, group=None, chunk=None, next_images=None,
                    gather_state=None):
    """Run `model.forward_batch` (romp_amd.ROMP) on this rank's shard (in chunks of `chunk` images) and all-gather the
    records ONCE.  Returns (dict of gathered tensors, per-rank counts).  `next_images`: see local_records; `gather_state`: see
    all_gather_records (a dict kept by the caller across steps: the fixed-capacity exchange)."""
    dev = images_local.device
    rec = local_records(model, images_local, img_offset, chunk, with_joints, with_verts, next_images=next_images)
    allrec, counts = all_gather_records(rec, record_width(with_joints, with_verts), dev, group, gather_state)
    return unpack_records(allrec, with_joints, with_verts), counts
