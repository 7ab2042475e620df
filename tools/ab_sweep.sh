#!/bin/bash
# A/B sweep: bench.py (headline only) with every library under build/variants/ in turn; one line per variant.
# The variant is chosen with ZS_LIB (zlibstream_amd/_native.py): the product library is never overwritten.
for f in build/variants/*.so; do
  ZS_DEV=1 ZS_LIB="$f" timeout -k 5 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$f', d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms'].items() if v > 0.1})"
done
