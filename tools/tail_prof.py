import os, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zlibstream_amd import Engine, datagen
eng = Engine()
for n in (32 << 10, 8 << 20):
    d = datagen.english(n, 3)
    for r in range(2):
        z = eng.deflate_batch([d], level=6)[0]
    print(n, len(z), flush=True)
