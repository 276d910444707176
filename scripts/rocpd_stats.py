#!/usr/bin/env python
"""Per-kernel duration summary of a rocprofv3 rocpd database (the default output format when --output-format is not given)."""
import glob
import sqlite3
import subprocess
import sys

db = sys.argv[1] if not sys.argv[1].endswith('/') else glob.glob(sys.argv[1] + '*.db')[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
names = subprocess.run(['c++filt'], input='\n'.join(r[0].replace('.kd', '') for r in rows), capture_output=True, text=True).stdout.split('\n')
print('"Name","Calls","AverageNs","MinNs","MaxNs"')
for r, n in zip(rows, names):
    print('"%s",%d,%.0f,%d,%d' % (n.split('(')[0], r[1], r[2], r[3], r[4]))
