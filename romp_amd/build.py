"""Build libromp_hip.so (gfx950) in-tree with hipcc.  `python -m romp_amd.build [--force]`."""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libromp_hip.so')
# (heaviest translation units first: the thread pool below starts them in this order and the link waits for the slowest)
SOURCES = ['conv_h2.hip', 'conv_h2c.hip', 'conv_h2c32.hip', 'conv_f32.hip', 'conv_h2d.hip', 'conv_h2b.hip', 'conv_h2r.hip',
           'conv_h2s.hip', 'conv_mfma.hip', 'conv_h2k.hip', 'conv_h2g.hip', 'conv_h2x.hip', 'conv_fup.hip', 'stem_fuse.hip', 'stem2.hip', 'stem7p.hip', 'net.hip', 'parse.hip', 'smpl.hip', 'bev.hip',
           'post.hip', 'render.hip', 'temporal.hip']
# optional: the bf16x3 family (`--conv_math bf16x3`; no committed variant table selects it; a minute of compile time): ROMP_WITH_BX3=1
OPTIONAL_BX3 = 'conv_bx3.hip'
# the fused-block kernels' tile loop is ONE fully unrolled body (270 MFMAs with a step of side work after each): beyond the default budget of `#pragma unroll`
_UNROLL = ['-mllvm', '-pragma-unroll-threshold=1000000']
EXTRA_FLAGS = {'conv_h2b.hip': _UNROLL, 'conv_h2c.hip': _UNROLL, 'conv_h2c32.hip': _UNROLL}
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-Wno-unused-value', '-Wno-unused-result']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), lib=None, objdir=None, with_bx3=None):
    """`extra_flags` / `lib` / `objdir`: a second, differently compiled library next to the product one (debug A/B builds,
    loaded through env ROMP_HIP_LIB).  `with_bx3` (default: env ROMP_WITH_BX3): also compile and link the bf16x3 kernel family."""
    hipcc = _hipcc()
    if with_bx3 is None:
        with_bx3 = os.environ.get('ROMP_WITH_BX3', '0') not in ('', '0')
    sources = SOURCES + ([OPTIONAL_BX3] if with_bx3 else [])
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + \
              [os.path.join(HERE, '..', 'include', 'romp_hip.h')]
    objdir = objdir or os.path.join(HERE, 'build')
    lib = lib or LIB
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in sources:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace('.hip', '.o'))
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + list(extra_flags) + EXTRA_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace('.hip', '.o')) for s in sources]
    marker = os.path.join(objdir, '.with_bx3')           # (the set of linked objects changed: relink)
    relink = os.path.exists(marker) != bool(with_bx3)
    if force or jobs or relink or _stale(lib, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', lib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
        if with_bx3:
            open(marker, 'w').close()
        elif os.path.exists(marker):
            os.remove(marker)
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
