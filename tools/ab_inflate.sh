cp zlibstream_amd/libzsgpu.so /tmp/lib_orig.so
cp "$1" zlibstream_amd/libzsgpu.so
python tools/bench_inflate.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['bit_exact_roundtrip'], d['stage_ms'])"
cp /tmp/lib_orig.so zlibstream_amd/libzsgpu.so
