#!/bin/bash
# the symbol kernel with (1) and without (0) its feeding wave, bench.py headline
for a in 0 1; do
  ZS_K5_AHEAD=$a timeout -k 5 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ahead $a', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in d['stage_ms'].items() if v > 0.1})"
done
