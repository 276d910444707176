"""hardware counters of the DEFAULT job with the branch streams ON (every committed PMC file of rounds 1-4 is a
--streams 0 pass).  Input: the counter_collection.csv and the kernel_trace.csv of ONE `rocprofv3 --kernel-trace --pmc ...` run.
Output (text): whether the profiler let kernels overlap at all (counter collection serialises dispatches on most stacks -- the
kernel trace of the same run says so: time with 0 / 1 / 2+ network kernels in flight), then per kernel class and summed over the
network kernels of one forward: the counters, and the derived shares (MFMA-busy / wave cycles, VMEM-issue / wave cycles,
LDS-wait / wave cycles, GRBM_GUI_ACTIVE).
usage: python scripts/pmc_concurrent.py <counter_collection.csv> <kernel_trace.csv> COUNTER [COUNTER ...]"""
import csv
import re
import sys
from collections import defaultdict

NET = re.compile(r'(conv_|bblock|seam1x1|fuse|stem_|ksum)')


def short(name):
    return re.sub(r'\(.*$', '', name).replace('void romp::', '').replace('romp::', '')[:64]


def overlap(trace):
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            n = short(r['Kernel_Name'])
            if NET.match(n):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith(('stem_mfma_kernel', 'stem_conv_kernel', 'stem2_kernel'))]
    if len(starts) < 2:
        return None
    a, b = starts[-2], starts[-1]                       # the last complete forward
    net = rows[a:b]
    ev = sorted([(k[0], 1) for k in net] + [(k[1], -1) for k in net])
    conc, last, infl = defaultdict(int), net[0][0], 0
    for t, d in ev:
        conc[min(infl, 3)] += t - last
        last, infl = t, infl + d
    span = max(k[1] for k in net) - net[0][0]
    return dict(kernels=len(net), span_us=span / 1e3, sum_us=sum(k[1] - k[0] for k in net) / 1e3,
                inflight_us={k: v / 1e3 for k, v in sorted(conc.items())})


def main(cc, trace, counters):
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    forwards = 0
    with open(cc) as f:
        for row in csv.DictReader(f):
            n = short(row['Kernel_Name'])
            if not NET.match(n):
                continue
            agg[n][row['Counter_Name']] += float(row['Counter_Value'])
            if row['Counter_Name'] == counters[0]:
                cnt[n] += 1
                forwards += n.startswith(('stem_mfma_kernel', 'stem_conv_kernel', 'stem2_kernel'))
    forwards = max(forwards, 1)
    ov = overlap(trace)
    print('# rocprofv3 --kernel-trace --pmc %s, default job, branch streams ON' % ' '.join(counters))
    if ov:
        print('# kernel trace of the SAME run, last forward: %d network kernels, span %.1f us, sum of durations %.1f us; time with n kernels in flight: %s'
              % (ov['kernels'], ov['span_us'], ov['sum_us'], ', '.join('%s%d: %.1f us' % ('>=' if k == 3 else '', k, v) for k, v in ov['inflight_us'].items())))
        two = sum(v for k, v in ov['inflight_us'].items() if k >= 2)
        print('# => under counter collection %.1f %% of the span has two or more kernels in flight (%s)' % (
            100.0 * two / max(ov['span_us'], 1e-9), 'the profiler serialises dispatches: these counters are per-kernel, NOT of the shared-chip regime'
            if two < 0.05 * ov['span_us'] else 'kernels do overlap under the profiler'))
    print('kernel,dispatches_per_forward,' + ','.join(c + '_per_forward' for c in counters))
    tot = defaultdict(float)
    for k in sorted(agg, key=lambda k: -agg[k][counters[0]]):
        print('"%s",%.1f,' % (k, cnt[k] / forwards) + ','.join('%.4g' % (agg[k][c] / forwards) for c in counters))
        for c in counters:
            tot[c] += agg[k][c] / forwards
    print('"ALL NETWORK KERNELS",%.1f,' % (sum(cnt.values()) / forwards) + ','.join('%.4g' % tot[c] for c in counters))
    wc = tot.get('SQ_WAVE_CYCLES', 0.0)
    if wc:
        for c in counters:
            if c not in ('SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE'):
                print('# %s / SQ_WAVE_CYCLES = %.3f' % (c, tot[c] / wc))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
