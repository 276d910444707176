"""Kernel-variant tables by NAME: what `RompNet.autotune(B)` measured, saved so that benchmark lines, rocprofv3 passes and
tests of one build all run the same kernels (the autotuner's near-ties used to flip the "dominant" kernel between
a bench run and the PMC passes quoted beside it).

A table is JSON: {"batch": B, "layers": {layer name: kernel variant name} for the conv ops, "note": ...} (rounds 2-3: a positional
list "ops", still read).  Variant names (`conv_h2r_k3s1_mt2_nt2_tw16_ck16`) survive additions to the variant list of the library,
indices do not; layer names survive stream markers and fusions that move op indices.  The committed
tables live in romp_amd/tune/; `default_table_path()` names the one for a (backbone, conv_math, batch) configuration.
"""
import ctypes as C
import json
import os

from . import lib as L

HERE = os.path.dirname(os.path.abspath(__file__))


def default_table_path(backbone='hrnet32', conv_math='f16x2', batch=32, workload='romp'):
    return os.path.join(HERE, 'tune', '%s_%s_%s_b%d.json' % (workload, backbone, conv_math, int(batch)))


def save_table(net, B, path, note=''):
    """{layer name: kernel variant name} for the conv ops of the net's program (round 4: keyed by LAYER, so that stream markers and
    fusions that come and go between builds do not invalidate a table; rounds 2-3 wrote a positional list `ops`)."""
    names = net.variant_names(B)
    layers = {ln: n for ln, n, op in zip(net.program.names, names, net.program.ops) if op.kind == L.OP_CONV and n.startswith('conv_')}
    with open(path, 'w') as f:
        json.dump({'batch': int(B), 'layers': layers, 'note': note}, f, indent=0, sort_keys=True)
    return layers


def resolve_table(net, B, table):
    """names -> variant indices for THIS build (None if an op's kernel does not exist here or cannot run the op).  `table`: the
    {layer: variant} dict of save_table, or a positional list (one entry per op of the program)."""
    lib = net.lib
    if isinstance(table, dict):
        missing = [ln for ln, op in zip(net.program.names, net.program.ops) if op.kind == L.OP_CONV and ln not in table]
        if missing:
            return None, 'the table has no entry for %d conv layers of the program (%s ..)' % (len(missing), missing[0])
        ops = [table.get(ln, '') if op.kind == L.OP_CONV else '' for ln, op in zip(net.program.names, net.program.ops)]
    else:
        ops = table
    if len(ops) != len(net.program.ops):
        return None, 'table has %d ops, the program %d' % (len(ops), len(net.program.ops))
    buf = C.create_string_buffer(128)
    nv = lib.romp_conv_num_variants()
    out = []
    for i, (name, op) in enumerate(zip(ops, net.program.ops)):
        if op.kind != L.OP_CONV or not name:
            out.append(-1)
            continue
        hit = -1
        for v in range(nv):
            if lib.romp_conv_describe(C.byref(op), int(B), v, buf, 128) == 0 and buf.value.decode() == name:
                hit = v
                break
        if hit < 0:
            return None, 'op %d: no valid variant named %s in this build' % (i, name)
        out.append(hit)
    return out, ''


def install_table(net, B, path):
    """Install the table at `path` for batch B.  -> (True, '') or (False, reason); the net is untouched on failure."""
    if not path or not os.path.exists(path):
        return False, 'no table at %s' % path
    t = json.load(open(path))
    if int(t.get('batch', -1)) != int(B):
        return False, 'table is for batch %s' % t.get('batch')
    variants, why = resolve_table(net, B, t['layers'] if 'layers' in t else t['ops'])
    if variants is None:
        return False, why
    net.set_tuned(B, variants)
    return True, ''
