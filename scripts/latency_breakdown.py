"""B = 1 network latency: graph / eager, branch streams on / off, per-op kernel time (serial sum) and op count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from romp_amd import synthetic as S
from romp_amd.net import RompNet
dev = torch.device('cuda:0')
net = RompNet(S.make_romp_state_dict(0), dev, max_batch=1, bf16x3=os.environ.get('CONV_MATH', 'f16x2'), split_k=int(os.environ.get('SPLIT_K', '128')))
x = S.make_images(1, seed=1, device=dev)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
c, p = net.forward_nhwc(x)
ops = net.program.ops
kinds = {}
for o in ops:
    kinds[o.kind] = kinds.get(o.kind, 0) + 1
print('ops', len(ops), 'by kind', kinds)
def timeit(n=50):
    for _ in range(5):
        net.forward_nhwc(x, c, p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        net.forward_nhwc(x, c, p)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for graph in (1,):
    for streams in (1, 0):
        net.set_graph(graph); net.set_streams(streams)
        print('graph %d streams %d: %.3f ms' % (graph, streams, timeit()))
net.set_graph(0); net.set_streams(0)
prof = net.profile(x, iters=5)
print('serial per-op sum %.3f ms over %d timed ops' % (sum(prof), sum(1 for v in prof if v > 0)))
import collections
by = collections.defaultdict(lambda: [0, 0.0])
names = net.variant_names(1)
for i, (o, ms) in enumerate(zip(ops, prof)):
    name = '%s %dx%d c%d-%d' % (names[i], o.H, o.W, o.Cin, o.Cout)
    by[name][0] += 1; by[name][1] += ms
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%-60s n %3d  %.3f ms  %.1f us each' % (k, v[0], v[1], v[1] / v[0] * 1e3))
