"""Single-image latency of the drop-in API: romp.ROMP(settings)(bgr_image) on a 720p frame (the reference's webcam
setting, docs/romp_evaluation.md:96-102 quotes 23.8 FPS on a GTX 1070Ti), and of the network alone at B=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import romp_amd
from romp_amd import synthetic as S
s = romp_amd.romp_settings([])
s.GPU, s.center_thresh, s.max_batch = 0, 1.3, 1
model = romp_amd.ROMP(s, state_dict=S.make_romp_state_dict(0), smpl_model=S.make_smpl_model(0))
model.model.set_graph(True)
rs = np.random.RandomState(0)
frame = rs.randint(0, 256, (720, 1280, 3)).astype(np.uint8)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(5):
        out = model(frame)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        out = model(frame)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    x = S.make_images(1, seed=1, device=torch.device('cuda:0'))
    c, p = model.model.forward_nhwc(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        model.model.forward_nhwc(x, c, p)
    torch.cuda.synchronize(); dn = (time.perf_counter() - t0) / n
print('ROMP(image) 720p frame, %d persons: %.2f ms = %.0f FPS end to end (upload + preprocess + net + parse + SMPL + projection + download); network alone B=1: %.2f ms'
      % (0 if out is None else out['cam'].shape[0], dt * 1e3, 1 / dt, dn * 1e3))

# ---- stage breakdown (synchronised between stages: slower in total than the pipelined call above)
from romp_amd.utils import img_preprocess_device, convert_tensor2numpy
from romp_amd.post_parser import parsing_outputs
import collections
acc = collections.OrderedDict()
def tick(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.) + (t1 - t0)
    return t1
with torch.cuda.stream(st):
    for it in range(30):
        torch.cuda.synchronize(); t = time.perf_counter()
        x, pad = img_preprocess_device(frame, model.tdevice); t = tick('upload + preprocess', t)
        cm, pm = model.model(x); t = tick('network', t)
        o = parsing_outputs(cm, pm, model.centermap_parser); t = tick('parse', t)
        o = model._finish(o, pad); t = tick('SMPL + projection (+ host PnP)', t)
        o = convert_tensor2numpy(o); t = tick('download', t)
print('stages (ms, synchronised): ' + ', '.join('%s %.3f' % (k, v / 30 * 1e3) for k, v in acc.items()))
