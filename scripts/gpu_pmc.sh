#!/bin/bash
# usage: gpu_pmc.sh "<COUNTER1 COUNTER2 ...>" [bench args]  -> per-kernel means in gpurun_out/pmc/
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COUNTERS="$1"; shift
TAG=$(echo $COUNTERS | tr ' ' '_' | cut -c1-60)
rm -rf /tmp/rp_pmc
timeout 900 rocprofv3 --kernel-trace --pmc $COUNTERS --output-format csv -d /tmp/rp_pmc -o pmc -- python $REPO/bench.py --no-cpu-baseline --no-f32-companion --no-roofline --steps 1 --warmup 1 "$@" > "$OUT/run_$TAG.log" 2>&1
echo "pmc exit $?"
f=$(find /tmp/rp_pmc -name "*counter_collection.csv" | head -1)
python - "$f" $COUNTERS > "$OUT/by_kernel_$TAG.csv" <<'PY'
import csv, re, sys
from collections import defaultdict
path, counters = sys.argv[1], sys.argv[2:]
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
with open(path) as f:
    for row in csv.DictReader(f):
        name = re.sub(r'\(.*$', '', row['Kernel_Name']).replace('void romp::', '').replace('romp::', '')
        agg[name][row['Counter_Name']] += float(row['Counter_Value'])
        if row['Counter_Name'] == counters[0]:
            cnt[name] += 1
print('kernel,dispatches,' + ','.join(c + '_mean' for c in counters))
for k in sorted(agg, key=lambda k: -agg[k][counters[0]]):
    if 'conv' in k or 'stem' in k or 'fuse' in k or 'smpl' in k or 'parse' in k:
        print('"%s",%d,' % (k, cnt[k]) + ','.join('%.0f' % (agg[k][c] / max(cnt[k], 1)) for c in counters))
PY
head -12 "$OUT/by_kernel_$TAG.csv"
