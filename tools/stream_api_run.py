"""The Stream API bench of bench.py's `secondary.stream_api` alone (the C++ mirror over english64, level 6).   python tools/stream_api_run.py [reps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from zlibstream_amd import datagen
data = datagen.english(64 << 20)
print(json.dumps(bench.secondary_stream_api(data, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 3)))
