#!/bin/bash
# A/B sweep on the 256 x 1 MiB batch (tools/batch_ab.py) with every library under build/variants/ (ZS_LIB selects it:
# the product library is never overwritten)
for f in build/variants/*.so; do echo "$f $(ZS_DEV=1 ZS_LIB="$f" timeout -k 5 200 python tools/batch_ab.py 2>/dev/null | tail -1)"; done
