"""The network alone at B = 1 (graph replay), N forwards: the thing to put under rocprofv3 --kernel-trace for scripts/timeline.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from romp_amd import synthetic as S
from romp_amd.net import RompNet
dev = torch.device('cuda:0')
B = int(os.environ.get('NET_B', '1'))
net = RompNet(S.make_romp_state_dict(0), dev, max_batch=B, bf16x3='f16x2')
net.autotune(B, iters=3)
net.set_graph(os.environ.get('NET_GRAPH', '1') == '1')
x = S.make_images(B, seed=1, device=dev)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    c, p = net.forward_nhwc(x)
    for _ in range(5):
        net.forward_nhwc(x, c, p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for _ in range(n):
        net.forward_nhwc(x, c, p)
    torch.cuda.synchronize()
print('network alone B=%d: %.3f ms' % (B, (time.perf_counter() - t0) / n * 1e3))
