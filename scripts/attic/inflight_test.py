"""Experiment: throughput of the network alone with 1 vs 2 forwards in flight (two contexts, two streams)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from romp_amd import synthetic as S
from romp_amd.net import RompNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
sd = S.make_romp_state_dict(0)
nets = [RompNet(sd, dev, max_batch=B, bf16x3=True) for _ in range(2)]
xs = [S.make_images(B, seed=1 + i, device=dev) for i in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
nets[0].autotune(B)
tv = nets[0].tuned_variants(B)
nets[1].set_tuned(B, tv)
outs = []
for n, x, s in zip(nets, xs, streams):
    n.set_graph(True)
    with torch.cuda.stream(s):
        outs.append(n.forward_nhwc(x))
torch.cuda.synchronize()
def run(k, steps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        j = i % k
        with torch.cuda.stream(streams[j]):
            nets[j].forward_nhwc(xs[j], outs[j][0], outs[j][1])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for k in (1, 2, 1, 2):
    ms = run(k)
    print('in flight %d: %.3f ms/forward  %.1f img/s' % (k, ms, B / ms * 1e3), flush=True)
