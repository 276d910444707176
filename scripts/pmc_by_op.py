"""rocprofv3 --pmc passes -> HBM traffic per OP INDEX of the layer program.

usage: pmc_by_op.py <op_kernels.json> <FETCH counter_collection.csv> <WRITE counter_collection.csv> <out.json>

`op_kernels.json` is what `bench.py --dump-op-kernels` wrote: the kernel variant name of every op of the profiled batch.
The two CSVs come from separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the same bench
command with `--streams 0` (one stream: the network's kernels are dispatched in op order).  Every forward starts with the
stem kernel; the k-th network kernel after it is the k-th op that launches a kernel (fork / join markers launch none).  Each
aligned dispatch's kernel name is checked against the op's variant name, so a pass taken with another variant table cannot be
mis-attributed.  Output per op: mean over the profiled forwards of 2 * FETCH_SIZE + WRITE_SIZE in bytes (FETCH_SIZE counts
64 B per 128-B request on gfx950: MI355X_MICROARCH.md, HBM section; both counters are KiB per dispatch)."""
import csv
import json
import re
import sys
from collections import defaultdict

NET_KERNEL = re.compile(r'romp::(conv_\w+_kernel|bblock32_kernel|bblockr_kernel|seam1x1_kernel|fuseup_kernel|stem_conv_kernel|stem_mfma_kernel|stem2_kernel|stem7_conv_kernel|stem7p_kernel|fusesum_kernel|ksum_kernel|maxpool3s2_kernel|conv3d_kernel|bev_pack_kernel|bev_maps_kernel)')
FIRST = ('stem_conv_kernel', 'stem_mfma_kernel', 'stem2_kernel', 'stem7_conv_kernel', 'stem7p_kernel')


def kernel_of(variant_name):
    if variant_name == 'bblock32':                           # conv_h2b.hip's kernel (single-image plans) or the row-pipelined one
        return ('bblock32_kernel', 'bblockr_kernel<32')
    if variant_name == 'bblock64':
        return ('bblockr_kernel<64',)
    if variant_name == 'seam1x1':
        return ('seam1x1_kernel',)
    if variant_name == 'fuseup':
        return ('fuseup_kernel',)
    if variant_name == 'stem2':
        return ('stem2_kernel',)
    if variant_name == 'stem7p':
        return ('stem7p_kernel',)
    if variant_name == 'seam1x1_ds':
        return ('seam1x1_kernel<1',)
    m = re.match(r'conv_(mfma|pp|bx3|bxd|h2do|h2o|h2d|h2p|h2w|h2q|h2r|h2s|h2k|h2g|h2)_k(\d+)s(\d+)_mt(\d+)_nt(\d+)_tw(\d+)_ck(\d+)', variant_name)
    if not m:
        return None
    fam, ks, s, mt, nt, tw, ck = m.groups()
    if fam == 'h2r':                                         # conv_h2r_kernel<P, NS, TW, KSUB>: ck = 16 * KSUB
        return 'conv_h2r_kernel<%s, %s, %s, %d>' % (mt, nt, tw, int(ck) // 16)
    if fam == 'h2g':                                         # conv_h2g_kernel<KS, P, NS, TW, S, KSUB>: ck = 32 * KSUB
        return 'conv_h2g_kernel<%s, %s, %s, %s, %s, %d>' % (ks, mt, nt, tw, s, int(ck) // 32)
    if fam == 'h2k':
        return 'conv_h2k_kernel<%s, %s, %s, %s>' % (ks, s, mt, tw)
    if fam in ('h2q', 'h2s'):
        return 'conv_%s_kernel<%s, %s, %s>' % (fam, mt, nt, tw)
    if fam in ('h2p', 'h2w'):
        return 'conv_h2p_kernel<%s, %s, %s, %s>' % (mt, nt, tw, 'true' if fam == 'h2w' else 'false')
    if fam in ('h2o', 'h2do'):
        fam += '4'
    return 'conv_%s_kernel<%s, %s, %s, %s, %s, %s>' % (fam, ks, s, mt, nt, tw, ck)


def per_op(csv_path, counter, names):
    rows = []
    with open(csv_path) as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') != counter:
                continue
            rows.append((int(r.get('Dispatch_Id', len(rows))), r.get('Kernel_Name', ''), float(r.get('Counter_Value', 0))))
    rows.sort()
    merged = []                                   # one entry per dispatch (a counter may come as several rows: sum them)
    for d, k, v in rows:
        if merged and merged[-1][0] == d:
            merged[-1][2] += v
        else:
            merged.append([d, k, v])
    launching = [i for i, n in enumerate(names) if n not in ('fork', 'join', 'nop', 'record', 'wait')]   # (nop: the first conv of a fused block)
    acc, cnt, forwards, pos, cur, skipped = defaultdict(float), defaultdict(int), 0, None, {}, []
    for d, k, v in merged:
        m = NET_KERNEL.search(k)
        if not m:
            continue
        if m.group(1) in FIRST:
            pos = 0
        if pos is None:
            continue
        if pos == 0:
            cur = {}
        i = launching[pos]
        want = kernel_of(names[i])
        if want is not None and not any(w in k for w in ((want,) if isinstance(want, str) else want)):
            # not a forward of the profiled table (the float32 calibration forward at start-up, an autotune launch): drop it
            skipped.append('dispatch %d is %s, op %d of the table is %s' % (d, k[:60], i, names[i]))
            pos = None
            continue
        cur[i] = v * 1024.0
        pos += 1
        if pos == len(launching):
            for j, b in cur.items():
                acc[j] += b
                cnt[j] += 1
            forwards += 1
            pos = None
    if forwards == 0:
        raise SystemExit('no forward of the table found in %s; first mismatches: %s' % (csv_path, skipped[:3]))
    return {i: acc[i] / cnt[i] for i in acc}, forwards


if __name__ == '__main__':
    info = json.load(open(sys.argv[1]))
    names = info['names']
    fetch, nf = per_op(sys.argv[2], 'FETCH_SIZE', names)
    write, nw = per_op(sys.argv[3], 'WRITE_SIZE', names)
    ops = {}
    for i in sorted(set(fetch) & set(write)):
        ops[str(i)] = {'kernel': names[i], 'bytes': 2.0 * fetch[i] + write[i], 'fetch_bytes_x2': 2.0 * fetch[i], 'write_bytes': write[i],
                       'algorithmic_bytes': info['bytes'][i], 'layer': info['op_names'][i]}
    json.dump({'batch': info['batch'], 'forwards_fetch_pass': nf, 'forwards_write_pass': nw, 'ops': ops,
               'note': '2*FETCH_SIZE + WRITE_SIZE per op, mean over the profiled forwards (scripts/pmc_by_op.py)'}, open(sys.argv[4], 'w'))
    tot = sum(e['bytes'] for e in ops.values())
    alg = sum(e['algorithmic_bytes'] for e in ops.values())
    print('%d ops aligned over %d / %d forwards; measured %.1f MB vs algorithmic %.1f MB per forward (x%.3f)' % (len(ops), nf, nw, tot / 1e6, alg / 1e6, tot / max(alg, 1)))
