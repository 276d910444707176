"""Where does a forward's WALL time go?  From a rocprofv3 --kernel-trace CSV (begin / end timestamp per dispatch): for the network
forwards in the trace (stem kernel .. last kernel before the next stem) the span, the sum of kernel durations, the time with
0 / 1 / 2 / 3+ kernels in flight, the largest gaps (and the kernels either side of them) and the busy time per kernel class.
usage: python scripts/timeline.py <kernel_trace.csv> [n_forwards_to_skip]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*$', '', name).replace('void romp::', '').replace('romp::', '')
    return name[:60]


def main(path, skip=2):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith(('stem_mfma_kernel', 'stem_conv_kernel', 'stem2_kernel'))]
    if len(starts) < skip + 2:
        raise SystemExit('only %d forwards in the trace' % len(starts))
    spans = []
    for a, b in zip(starts[skip:-1], starts[skip + 1:]):
        ks = rows[a:b]
        # cut the parse / SMPL kernels of the pipelined previous chunk out of the statistics of the NETWORK (they run on another stream)
        net = [k for k in ks if re.match(r'(conv_|bblock|seam1x1|fuse|stem_|ksum)', k[2])]
        t0, t1 = net[0][0], max(k[1] for k in net)
        ev = sorted([(k[0], 1) for k in net] + [(k[1], -1) for k in net])
        conc, last, inflight = defaultdict(int), t0, 0
        gaps = []
        for t, d in ev:
            conc[min(inflight, 3)] += t - last
            if inflight == 0 and t - last > 0:
                gaps.append((t - last, last))
            last, inflight = t, inflight + d
        busy = defaultdict(int)
        for k in net:
            busy[re.sub(r'<.*', '', k[2])] += k[1] - k[0]
        allgaps = [g[0] for g in gaps]
        spans.append(dict(allgaps=allgaps, net=net, t0=t0, span=t1 - t0, ksum=sum(k[1] - k[0] for k in net), n=len(net), conc=dict(conc), gaps=sorted(gaps, reverse=True)[:5], busy=dict(busy),
                          other=sum(k[1] - k[0] for k in ks if k not in net)))
    n = len(spans)
    avg = lambda f: sum(f(s) for s in spans) / n
    print('%d forwards: span %.3f ms, sum of network kernel durations %.3f ms over %.0f kernels, other kernels in the window %.3f ms' % (
        n, avg(lambda s: s['span']) / 1e6, avg(lambda s: s['ksum']) / 1e6, avg(lambda s: s['n']), avg(lambda s: s['other']) / 1e6))
    for c in range(4):
        print('  %s network kernels in flight: %.3f ms' % ('3+' if c == 3 else str(c), avg(lambda s: s['conc'].get(c, 0)) / 1e6))
    g = spans[-1]['allgaps']
    print('  idle gaps of the last forward: %d, sum %.1f us; <2us %d, 2-5us %d, 5-10us %d, >10us %d' % (
        len(g), sum(g) / 1e3, sum(x < 2000 for x in g), sum(2000 <= x < 5000 for x in g), sum(5000 <= x < 10000 for x in g), sum(x >= 10000 for x in g)))
    if len(sys.argv) > 3:                                      # the last forward, kernel by kernel: start (us from the stem), duration, name
        with open(sys.argv[3], 'w') as f:
            for k in spans[-1]['net']:
                f.write('%9.2f %8.2f %s\n' % ((k[0] - spans[-1]['t0']) / 1e3, (k[1] - k[0]) / 1e3, k[2]))
    print('  largest idle gaps of the last forward (us):', ', '.join('%.1f' % (g[0] / 1e3) for g in spans[-1]['gaps']))
    tot = defaultdict(float)
    for s in spans:
        for k, v in s['busy'].items():
            tot[k] += v / n
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
        print('  %-28s %.3f ms' % (k, v / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
