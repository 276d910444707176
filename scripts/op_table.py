"""Per-op table of the tuned network at batch B (GPU): variant, ms, GFLOP, TFLOP/s, algorithmic GB/s.
usage: python scripts/op_table.py [B] [f32|bf16x3|f16x2|all] [romp|bev]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from romp_amd import synthetic as S
from romp_amd.net import RompNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
math = sys.argv[2] if len(sys.argv) > 2 else 'f16x2'
dev = torch.device('cuda:0')
if len(sys.argv) > 3 and sys.argv[3] == 'bev':
    from romp_amd.bev_plan import build_bev_hrnet32
    net = RompNet(S.make_bev_state_dict(0), dev, max_batch=B, builder=build_bev_hrnet32,
                  out_shapes=((64, 128, 128), (3, 64, 128, 128)), bf16x3=math)
else:
    net = RompNet(S.make_romp_state_dict(0), dev, max_batch=B, bf16x3=math)
x = S.make_images(B, seed=1, device=dev)
net.autotune(B)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ms = net.profile(x, iters=5)
names = net.variant_names(B)
P = net.program
rows = []
for i, (nm, t) in enumerate(zip(names, ms)):
    if nm in ('fork', 'join', 'record', 'wait'):
        continue
    op = P.ops[i]
    if P.names[i].startswith('bev.'):
        print('  %-28s %-36s %8.3f ms  %7.1f GF  %8.1f MB' % (P.names[i], nm, t, P.flops[i] * B / 1e9, P.bytes[i] * B / 1e6))
    rows.append((t, P.names[i], nm, op.H, op.W, op.Cin, op.Cout, P.flops[i] * B / 1e9, P.bytes[i] * B / 1e6))
tot = sum(r[0] for r in rows)
print('total serial ms %.3f over %d ops' % (tot, len(rows)))
agg = {}
for t, lname, nm, H, W, ci, co, gf, mb in rows:
    key = (nm.split('_mt')[0].replace('conv_', ''), H, W, ci, co)
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0, set()])
    a[0] += 1; a[1] += t; a[2] += gf; a[3] += mb; a[4].add(nm)
print('%-14s %5s %5s %5s %5s %4s %8s %7s %8s %8s  variants' % ('kind', 'H', 'W', 'Cin', 'Cout', 'n', 'ms', 'us/op', 'TF', 'GB/s'))
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-14s %5d %5d %5d %5d %4d %8.3f %7.1f %8.1f %8.0f  %s' % (*key, a[0], a[1], a[1] / a[0] * 1e3, a[2] / a[1] if a[1] else 0,
                                                                  a[3] / a[1] if a[1] else 0, ','.join(sorted(v.split('_k')[1] if '_k' in v else v for v in a[4]))))
