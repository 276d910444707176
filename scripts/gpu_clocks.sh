#!/bin/bash
# Engine clock and socket power WHILE the default job runs (1 x MI355X): rocm-smi sampled every 0.25 s beside bench.py's timed
# steps.  The roofs of DESIGN.md section 4 are quoted at the 2.4 GHz peak clock; this says what the chip actually sustains under
# the f16x2 kernels.  -> gpurun_out/clocks.txt (one line per sample: sclk MHz, mclk MHz, W) + a summary line.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"; mkdir -p gpurun_out
OUT=gpurun_out/clocks_raw.txt; : > $OUT
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" >> $OUT; echo "--" >> $OUT; sleep 0.25; done ) &
SAMPLER=$!
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-f32-companion --no-parity --no-end-to-end --no-latency --no-roofline > gpurun_out/clocks_bench.log 2>&1
kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
python - <<'PY'
import re
s = open('gpurun_out/clocks_raw.txt').read().split('--\n')
rows = []
for blk in s:
    sc = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', blk); mc = re.search(r'mclk clock level: \S+ \((\d+)Mhz\)', blk)
    pw = re.search(r'Power \(W\): ([0-9.]+)', blk)
    if sc:
        rows.append((int(sc.group(1)), int(mc.group(1)) if mc else 0, float(pw.group(1)) if pw else 0.0))
with open('gpurun_out/clocks.txt', 'w') as f:
    for r in rows:
        f.write('%d %d %.0f\n' % r)
    busy = [r for r in rows if r[2] > 0.5 * max(x[2] for x in rows)] if rows else []
    if busy:
        line = 'samples %d (under load %d): sclk under load min %d / median %d / max %d MHz, power median %.0f W max %.0f W' % (
            len(rows), len(busy), min(b[0] for b in busy), sorted(b[0] for b in busy)[len(busy) // 2], max(b[0] for b in busy),
            sorted(b[2] for b in busy)[len(busy) // 2], max(b[2] for b in busy))
    else:
        line = 'no samples parsed: ' + s[0][:200].replace('\n', ' | ')
    f.write('# ' + line + '\n'); print(line)
PY
grep -o '"value": [0-9.]*' gpurun_out/clocks_bench.log | head -1
