"""Debug aid for levels 1-3: the first token at which the device's stream differs from the oracle's.
   python tools/fast_diff.py <corpus file | english:<bytes>> <level>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_binding
import deflate_tokens
from zlibstream_amd import Engine, datagen
name, level = sys.argv[1], int(sys.argv[2])
data = datagen.english(int(name.split(":")[1])) if name.startswith("english:") else oracle_binding.corpus(name)
eng = Engine(0); orc = oracle_binding.Oracle()
z = eng.deflate_batch([data], level=level)[0]
ref = orc.compress(data, level)
print("device %d bytes, oracle %d bytes, equal %s" % (len(z), len(ref), z == ref))
if z != ref:
    a, b = deflate_tokens.tokens(z)[0], deflate_tokens.tokens(ref)[0]
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            print("token %d: device %s oracle %s" % (i, x, y))
            print("before:", a[max(0, i - 3):i], "device next:", a[i:i + 4], "oracle next:", b[i:i + 4])
            break
    else:
        print("tokens equal up to", min(len(a), len(b)), "of", len(a), len(b))
