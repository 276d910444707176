"""The RCCL code path of the multi-GPU bench on ONE GPU (world_size 1): process-group init with the
nccl backend (= RCCL on ROCm), all-gather of the counts, padded all_gather_into_tensor of the
per-person records, unpacking -- must reproduce forward_batch exactly.  (World size 2 is covered on
CPU with gloo in test_distributed_gloo.py; the 8-GPU run is the driver's.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle import romp_oracle as O

pytestmark = pytest.mark.gpu


def test_sharded_forward_nccl_world1():
    import romp_amd
    from romp_amd import distributed as D
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        settings = romp_amd.romp_settings([])
        settings.GPU, settings.center_thresh, settings.max_batch = 0, 1.3, 4
        model = romp_amd.ROMP(settings, state_dict=O.make_romp_state_dict(0, center_bias=2.0), smpl_model=O.make_synthetic_smpl(0))
        img = O.make_images(4, seed=8).to(dev)
        ref, bids = model.forward_batch(img)
        lo, hi = D.shard_range(4, 0, 1)
        assert (lo, hi) == (0, 4)
        out, counts = D.sharded_forward(model, img, lo, with_joints=True, with_verts=True)
        dist.barrier()
        assert counts == [ref['cam'].shape[0]]
        assert torch.equal(out['image_ids'], bids)
        assert torch.equal(out['smpl_thetas'], ref['smpl_thetas']) and torch.equal(out['cam'], ref['cam'])
        assert torch.equal(out['joints'], ref['joints']) and torch.equal(out['verts'], ref['verts'])
        flat = ref['center_preds'][:, 1] // 8 * 64 + ref['center_preds'][:, 0] // 8
        assert torch.equal(out['flat_inds'], flat)
    finally:
        dist.destroy_process_group()
