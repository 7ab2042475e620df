for w in 4 2 1 0.5 0.25; do echo "== wake $w"; ZS_FR_WAKE=$w ZS_DEBUG=1 timeout -k 10 100 python tools/fast_big.py 64 2>&1 | grep -v "^zs: run\|stage" | awk 'NR%3!=1' | cut -c1-700; done
