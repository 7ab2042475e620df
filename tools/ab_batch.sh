#!/bin/bash
# A/B sweep on the 256 x 1 MiB batch (tools/batch_ab.py) with every library under build/variants/
cp zlibstream_amd/libzsgpu.so /tmp/lib_orig.so
for f in build/variants/*.so; do cp "$f" zlibstream_amd/libzsgpu.so; echo "$f $(timeout -k 5 200 python tools/batch_ab.py 2>/dev/null | tail -1)"; done
cp /tmp/lib_orig.so zlibstream_amd/libzsgpu.so
