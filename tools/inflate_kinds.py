"""Inflate of streams that are not a chain of dynamic blocks: fixed-Huffman (Z_FIXED), stored (level 0), HuffmanOnly, Rle, many
flushes (short blocks), one enormous block.   python tools/inflate_kinds.py [MiB]"""
import sys, os, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zlibstream_amd import Engine, datagen
eng = Engine(0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
d = datagen.english(mib << 20, 7)
def comp(level, strategy=0, mem=8, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, 15, mem, strategy)
    if not flush_every:
        return c.compress(d) + c.flush()
    out = []
    for o in range(0, len(d), flush_every):
        out.append(c.compress(d[o:o + flush_every])); out.append(c.flush(zlib.Z_SYNC_FLUSH))
    out.append(c.flush())
    return b"".join(out)
for name, z in (("level 6", comp(6)), ("Z_FIXED", comp(6, zlib.Z_FIXED)), ("level 0 (stored)", comp(0)), ("Z_HUFFMAN_ONLY", comp(6, zlib.Z_HUFFMAN_ONLY)),
                ("Z_RLE", comp(6, zlib.Z_RLE)), ("level 1", comp(1)), ("level 9, memLevel 9", comp(9, 0, 9)), ("memLevel 1 (short blocks)", comp(6, 0, 1)),
                ("sync flush every 4 KiB", comp(6, 0, 8, 4096)), ("sync flush every 64 KiB", comp(6, 0, 8, 65536))):
    d_z = torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda()
    out = torch.empty(len(d), dtype=torch.uint8, device="cuda")
    a = ([d_z.data_ptr()], [len(z)], [out.data_ptr()], [len(d)])
    n = eng.inflate_batch_device(*a)[0]
    ok = n == len(d) and out.cpu().numpy().tobytes() == d
    torch.cuda.synchronize(); t = time.perf_counter()
    eng.inflate_batch_device(*a)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%-28s %10d compressed  %9.2f ms %9.1f MB/s ok %s" % (name, len(z), dt * 1e3, len(d) / dt / 1e6, ok), flush=True)
