"""Replicates bench.py's roofline pass and prints the slowest ops (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import romp_amd
from romp_amd import synthetic as S, distributed as D
dev = torch.device('cuda:0')
st = romp_amd.romp_settings([])
st.GPU, st.center_thresh, st.max_batch, st.conv_math = 0, 1.3, 32, 'f16x2'
model = romp_amd.ROMP(st, state_dict=S.make_romp_state_dict(0), smpl_model=S.make_smpl_model(0))
model.model.autotune(32)
model.model.set_graph(True)
x = S.make_images(64, seed=1, device=dev)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(3):
        D.local_records(model, x, 0, chunk=32)
    torch.cuda.synchronize()
    for rep in range(2):
        ms = model.model.profile(x[:32], iters=3)
        names = model.model.variant_names(32)
        P = model.model.program
        rows = sorted(zip(ms, P.names, names, [(o.H, o.Cin, o.Cout, o.groups) for o in P.ops]), key=lambda r: -r[0])[:8]
        print('pass', rep, 'sum %.3f ms' % sum(ms))
        for r in rows:
            print('   %8.3f ms  %-40s %-36s %s' % r)
