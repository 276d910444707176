"""The counted `s_waitcnt vmcnt(N)` of the tile loops (round 5) are only right while N does not exceed the number of vector-memory
instructions the COMPILED kernel issues between the LDS-DMA they wait for and the wait itself: a wave's memory operations retire in
issue order, so "at most N outstanding" covers the DMA iff at least N younger operations exist.  hipcc is free to merge or split
loads and stores; this script compiles the kernels to assembly and counts.

  seam1x1_kernel<0> (csrc/conv_h2x.hip): vmcnt(52) -- 16 t stores + 32 residual loads + 4 u stores after the m DMA of the next tile
  seam1x1_kernel<1>:                     vmcnt(20) -- 16 + 4 stores after the m and x0 DMAs
  bblockr_kernel<64, 0, *> / <32, 0, *> (csrc/conv_h2c.h, plain and strip form): vmcnt(6) / vmcnt(3) -- the output-row stores younger than the last halo piece

usage: python scripts/check_counted_waits.py [h2x] [h2c] [h2c32]     (default: h2x; h2c / h2c32 take about a minute each)
Exit status 0 iff every counted wait found is covered."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'romp_amd', 'csrc')
FILES = {'h2x': ('conv_h2x.hip', []), 'h2c': ('conv_h2c.hip', ['-mllvm', '-pragma-unroll-threshold=1000000']),
         'h2c32': ('conv_h2c32.hip', ['-mllvm', '-pragma-unroll-threshold=1000000'])}
EXPECT = {'seam1x1_kernelILi0ELi0E': (52, 'loop'), 'seam1x1_kernelILi1ELi0E': (20, 'loop'),
          'bblockr_kernelILi64ELi0ELb0E': (6, 'layout'), 'bblockr_kernelILi32ELi0ELb0E': (3, 'layout'),
          'bblockr_kernelILi64ELi0ELb1E': (6, 'layout'), 'bblockr_kernelILi32ELi0ELb1E': (3, 'layout'),       # (the halo-carrying form: conv2 and its DMA schedule are the plain form's)
          }
VMEM = re.compile(r'^\s*(global_load|global_store|buffer_load|buffer_store|global_atomic|buffer_atomic|flat_load|flat_store|scratch_)')


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (not os.path.isabs(c) or os.path.exists(c)):
            return c
    raise SystemExit('hipcc not found')


def kernels(asm):
    """-> {mangled name: [raw lines: labels and instructions]} for every function of the assembly text"""
    out, name = {}, None
    for l in asm.split('\n'):
        m = re.match(r'^(_Z\w+):\s*(;.*)?$', l)
        if m:
            name = m.group(1)
            out[name] = []
        elif l.startswith('.Lfunc_end'):
            name = None
        elif name and (l.startswith('.LBB') or (l.startswith('\t') and not l.lstrip().startswith(('.', ';')))):
            out[name].append(l)
    return out


def check(name, raw, n_expected, mode):
    """Every hand-written `s_waitcnt vmcnt(n_expected) lgkmcnt(0)` of the kernel (raw assembly lines, labels included).
    mode 'layout' (bblockr: the tile loop is one straight line): walk back to the nearest LDS-DMA, count the vector-memory
    instructions in between.  mode 'loop' (seam1x1: hipcc lays the residual loads out behind the loop's tail and jumps back, so
    layout order is not issue order): the DMA is the tile's FIRST memory operation by construction (asm volatile, memory clobber), so
    every other vector-memory instruction of the outermost loop around the wait is younger: count those.  -> [(n, younger)]"""
    res = []
    label_idx = {}
    for i, l in enumerate(raw):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            label_idx[m.group(1)] = i
    is_dma = lambda l: l.strip().startswith('buffer_load_dwordx4') and l.rstrip().endswith('lds')
    for i, l in enumerate(raw):
        m = re.match(r'\s*s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)', l)
        if not m or int(m.group(1)) != n_expected:
            continue
        if mode == 'layout':
            younger, j = 0, i - 1                        # (the 8-byte stores in between are ROMP_TRACE stamps behind a branch: not counted)
            while j >= 0 and not is_dma(raw[j]):
                younger += bool(VMEM.match(raw[j])) and 'global_store_dwordx2' not in raw[j]
                j -= 1
            if j < 0:
                raise SystemExit('%s: no LDS-DMA in front of the counted wait' % name)
        else:
            lo, hi = None, None
            for j, b in enumerate(raw):
                mb = re.search(r'\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', b)
                if mb and label_idx.get(mb.group(2), 1 << 30) < j and label_idx[mb.group(2)] <= i <= j:
                    lo = label_idx[mb.group(2)] if lo is None else min(lo, label_idx[mb.group(2)])
                    hi = j if hi is None else max(hi, j)
            if lo is None:
                raise SystemExit('%s: the counted wait is not inside a loop' % name)
            younger = sum(bool(VMEM.match(b)) and not is_dma(b) for b in raw[lo:hi + 1])
            if not any(is_dma(b) for b in raw[lo:hi + 1]):
                raise SystemExit('%s: no LDS-DMA inside the loop of the counted wait' % name)
        res.append((n_expected, younger))
    return res


def main(which):
    bad = 0
    for key in which:
        src, extra = FILES[key]
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, 'k.s')
            cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '--cuda-device-only', '-S'] + extra + [os.path.join(CSRC, src), '-o', out]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit('hipcc failed for %s:\n%s' % (src, r.stderr[-2000:]))
            ks = kernels(open(out).read())
        found = 0
        for name, body in ks.items():
            for tag, (n, mode) in EXPECT.items():
                if name.endswith(tag + 'EEvNS_10ConvParamsE'):
                    for n_wait, younger in check(name, body, n, mode):
                        found += 1
                        ok = younger == n_wait       # exactly: a duplicated / peeled path would raise the static count of the loop above the issue count of one path
                        bad += not ok
                        print('%-62s vmcnt(%d): %d vector-memory instructions younger than the DMA  %s' % (name[:62], n_wait, younger, 'ok' if ok else 'NOT COVERED'))
                    break
        if not found:
            raise SystemExit('%s: no counted wait found (kernel renamed or the wait removed?)' % src)
    return bad


if __name__ == '__main__':
    sys.exit(1 if main([a for a in sys.argv[1:] if a in FILES] or ['h2x']) else 0)
