"""tests/golden/resnet50_b1.npz from THE REFERENCE backbone: romp/lib/models/resnet_50.py (ResNet_50) imported by file
path.  Its imports that do not exist here are stubbed with the only pieces it uses: torchvision.transforms.functional
.normalize (per-channel (x-mean)/std), utils.BHWC_to_BCHW (the reference's own, simple_romp/romp/model.py:39-44),
config.args, and a `models` package whose __path__ is the reference directory (basic_modules / CoordConv are the real
files).  Needs /root/reference.  The backbone is pinned by this fixture; the three head towers are the ROMPv1 structure
already pinned by romp_net_b1.npz."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import resnet_oracle as RO, romp_oracle as O  # noqa: E402

LIB = '/root/reference/romp/lib'


def load_reference_resnet():
    tv = types.ModuleType('torchvision'); tvm = types.ModuleType('torchvision.models'); tvr = types.ModuleType('torchvision.models.resnet')
    tvt = types.ModuleType('torchvision.transforms'); tvf = types.ModuleType('torchvision.transforms.functional')

    def normalize(t, mean, std, inplace=False):
        m = torch.as_tensor(mean, dtype=t.dtype).view(-1, 1, 1)
        s = torch.as_tensor(std, dtype=t.dtype).view(-1, 1, 1)
        return (t - m) / s
    tvf.normalize = normalize
    tv.models, tvm.resnet, tv.transforms, tvt.functional = tvm, tvr, tvt, tvf
    for name, mod in (('torchvision', tv), ('torchvision.models', tvm), ('torchvision.models.resnet', tvr),
                      ('torchvision.transforms', tvt), ('torchvision.transforms.functional', tvf)):
        sys.modules[name] = mod
    cfg = types.ModuleType('config')
    cfg.args = lambda: types.SimpleNamespace(resnet_pretrain='')
    sys.modules['config'] = cfg
    spec = importlib.util.spec_from_file_location('ref_simple_model', '/root/reference/simple_romp/romp/model.py')
    sm = importlib.util.module_from_spec(spec); spec.loader.exec_module(sm)
    ut = types.ModuleType('utils')
    ut.BHWC_to_BCHW, ut.copy_state_dict = sm.BHWC_to_BCHW, (lambda *a, **k: None)
    sys.modules['utils'] = ut
    pkg = types.ModuleType('models'); pkg.__path__ = [os.path.join(LIB, 'models')]
    sys.modules['models'] = pkg
    spec = importlib.util.spec_from_file_location('models.resnet_50', os.path.join(LIB, 'models', 'resnet_50.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    return mod.ResNet_50


def main():
    ResNet_50 = load_reference_resnet()
    net = ResNet_50().eval()
    sd = RO.make_resnet_state_dict(0)
    ref_keys = [k for k in net.state_dict().keys() if not k.endswith('num_batches_tracked')]
    mine = [k[len('backbone.'):] for k in sd if k.startswith('backbone.')]
    assert ref_keys == mine, (len(ref_keys), len(mine), [a for a, b in zip(ref_keys, mine) if a != b][:5])
    missing = net.load_state_dict({k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}, strict=False)
    assert not missing.unexpected_keys and all(k.endswith('num_batches_tracked') for k in missing.missing_keys)
    img = O.make_images(1, seed=7)
    with torch.no_grad():
        ref = net(img)
    out = RO.backbone_forward(sd, img)
    err = (ref - out).abs().max().item()
    print('reference ResNet_50 vs restatement: max-abs %.3e (range %.3f..%.3f)' % (err, ref.min(), ref.max()))
    assert ref.shape == (1, 64, 128, 128) and err < 2e-5
    cm, pm = RO.resnet_romp_forward(sd, img)
    g = torch.Generator().manual_seed(1)
    pos = torch.randint(0, 128 * 128, (512,), generator=g).numpy()
    path = os.path.join(ROOT, 'tests', 'golden', 'resnet50_b1.npz')
    np.savez_compressed(path, feat_samples=ref[0].reshape(64, -1)[:, pos].numpy(), sample_pos=pos,
                        feat_chan_sum=ref[0].double().sum((1, 2)).numpy(), center_maps=cm.numpy(),
                        params_chan_sum=pm[0].double().sum((1, 2)).numpy())
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
