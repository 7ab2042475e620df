#!/bin/bash
# A/B: run bench.py with an alternative libzsgpu.so (path in $1), print value + stage times
cp zlibstream_amd/libzsgpu.so /tmp/lib_orig.so
cp "$1" zlibstream_amd/libzsgpu.so
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['stage_ms'])"
cp /tmp/lib_orig.so zlibstream_amd/libzsgpu.so
