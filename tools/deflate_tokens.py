"""Token-level view of a zlib / deflate stream (test tool): yields (position, length, distance) per symbol."""
LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class Bits:
    def __init__(self, data, pos=0):
        self.d, self.p = data, pos * 8

    def get(self, n):
        v = 0
        for i in range(n):
            v |= ((self.d[self.p >> 3] >> (self.p & 7)) & 1) << i
            self.p += 1
        return v


def table(lens):
    codes, code, out = {}, 0, {}
    cnt = [0] * 16
    for l in lens:
        cnt[l] += 1
    cnt[0] = 0
    nxt = [0] * 16
    for b in range(1, 16):
        code = (code + cnt[b - 1]) << 1
        nxt[b] = code
    for s, l in enumerate(lens):
        if l:
            out[(l, nxt[l])] = s
            nxt[l] += 1
    return out


def sym(b, t):
    code, l = 0, 0
    while True:
        code = (code << 1) | b.get(1)
        l += 1
        if (l, code) in t:
            return t[(l, code)]
        if l > 15:
            raise ValueError("bad code")


def tokens(z, raw=False):
    b = Bits(z, 0 if raw else 2)
    pos, out, blocks = 0, [], []
    while True:
        final, typ = b.get(1), b.get(2)
        blocks.append((pos, typ, b.p))
        if typ == 0:
            b.p = (b.p + 7) & ~7
            n = b.get(16)
            b.get(16)
            for _ in range(n):
                out.append((pos, 1, 0))
                b.get(8)
                pos += 1
        else:
            if typ == 1:
                lt = table([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)
                dt = table([5] * 30)
            else:
                hl, hd, hc = b.get(5) + 257, b.get(5) + 1, b.get(4) + 4
                cl = [0] * 19
                for i in range(hc):
                    cl[ORDER[i]] = b.get(3)
                ct = table(cl)
                lens = []
                while len(lens) < hl + hd:
                    s = sym(b, ct)
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + b.get(2))
                    elif s == 17:
                        lens += [0] * (3 + b.get(3))
                    else:
                        lens += [0] * (11 + b.get(7))
                lt, dt = table(lens[:hl]), table(lens[hl:])
            while True:
                s = sym(b, lt)
                if s < 256:
                    out.append((pos, 1, 0))
                    pos += 1
                elif s == 256:
                    break
                else:
                    ln = LBASE[s - 257] + b.get(LEXT[s - 257])
                    d = sym(b, dt)
                    dist = DBASE[d] + b.get(DEXT[d])
                    out.append((pos, ln, dist))
                    pos += ln
        if final:
            break
    return out, blocks
