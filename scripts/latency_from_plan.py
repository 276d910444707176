"""Single-image latency starting from a plan file (python scripts/latency_from_plan.py make|run <plan>): `make` lowers the
synthetic checkpoint, measures the B=1 kernel table and writes the plan; `run` loads it (no autotune launches: a clean
rocprofv3 trace) and times ROMP(settings)(720p frame)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import romp_amd
from romp_amd import synthetic as S
mode, path = sys.argv[1], sys.argv[2]
if mode == 'make':
    from romp_amd.export import save_plan
    from romp_amd.net import RompNet
    net = RompNet(S.make_romp_state_dict(0), 'cuda:0', max_batch=1, bf16x3='f16x2')
    net.autotune(1)
    save_plan(net, path)
    print('wrote', path)
else:
    s = romp_amd.romp_settings(['--plan_path', path])
    s.GPU, s.center_thresh, s.max_batch = 0, 1.3, 1
    model = romp_amd.ROMP(s, smpl_model=S.make_smpl_model(0))
    model.model.set_graph(True)
    frame = np.random.RandomState(0).randint(0, 256, (720, 1280, 3)).astype(np.uint8)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(5):
            out = model(frame)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            out = model(frame)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print('ROMP(image) from plan, 720p frame, %d persons: %.2f ms = %.0f FPS' % (0 if out is None else out['cam'].shape[0], dt * 1e3, 1 / dt))
