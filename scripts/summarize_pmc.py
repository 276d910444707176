"""Aggregate a rocprofv3 counter_collection.csv by kernel name: dispatch count, sum and mean of
one counter.  usage: summarize_pmc.py <counter_collection.csv> <COUNTER>"""
import csv
import re
import sys
from collections import defaultdict

path, counter = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: [0, 0.0])
with open(path) as f:
    rd = csv.DictReader(f)
    for row in rd:
        if row.get('Counter_Name') != counter:
            continue
        name = row.get('Kernel_Name', '?')
        name = re.sub(r'\(.*$', '', name)
        a = agg[name]
        a[0] += 1
        a[1] += float(row.get('Counter_Value', 0))
print('kernel,dispatches,%s_sum,%s_mean' % (counter, counter))
for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%.1f,%.1f' % (k, n, s, s / n))
