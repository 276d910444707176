#!/usr/bin/env python
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr saved to a file): one line per kernel."""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
blocks = re.split(r'remark: [^\n]*Function Name: ', t)[1:]
names = [b.split('\n')[0].strip() for b in blocks]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
for b, n in zip(blocks, dem):
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    n = n.replace('void romp::', '').replace('(romp::ConvParams)', '')
    scratch = g(r'ScratchSize \[bytes/lane\]')
    if flt in n or scratch > 0:
        print('%-50s vgpr %3d agpr %3d scratch %4d occ %d sgpr %3d lds %d' % (n, g('VGPRs'), g('AGPRs'), scratch, g(r'Occupancy \[waves/SIMD\]'), g('SGPRs'), g(r'LDS Size \[bytes/block\]')))
