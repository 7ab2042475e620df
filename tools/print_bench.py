"""The figures of a bench line (bench.py's JSON) one per line.   python tools/print_bench.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], d["unit"], d["ms_per_step"], "ms", "bit-identical:", d.get("bit_identical_to_cpu"), "roofline frac", d["roofline"]["frac"])
for k, v in d.get("secondary", {}).items():
    if k == "stream_api":
        ow = v.get("one_write", {})
        print(k, "deflate", ow.get("deflate_MBps"), "MB/s", ow.get("deflate_ms"), "ms; inflate", ow.get("inflate_MBps"), "MB/s", ow.get("inflate_ms"), "ms")
    elif k == "host_path":
        print(k, v.get("pageable"), v.get("pinned"))
    else:
        print(k, v.get("value"), v.get("unit"), v.get("ms_per_step"), "ms", "bit-identical:", v.get("bit_identical_to_cpu"), "cpu", (v.get("cpu_baseline") or {}).get("value"))
