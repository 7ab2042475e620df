// stream_api_bench.cpp -- what a caller of the drop-in Stream API sees: ZlibOutputStream / ZlibInputStream
// (include/zsgpu.hpp, the C++ mirror of ZlibOutputStream.cs / ZlibInputStream.cs) run with the reference's own
// loops -- 512-byte output chunks per Deflate call (ZlibOutputStream.cs:125-168, 213-256), 8 KiB input chunks per
// Inflate call (ZlibInputStream.cs:133-186) -- over a buffer read from a file.  Host buffers in and out, so PCIe
// and the chunk protocol are inside the timed region.  Prints one JSON line.
//
// usage: stream_api_bench <input file> <level> <reps> [write_bytes]
//   write_bytes > 0: the input is handed over in Writes of that many bytes (default: one Write of the whole buffer,
//   as DeflateCorpusBenchmark.cs:86-100 does).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <streambuf>
#include <vector>

#include "../include/zsgpu.hpp"

using namespace SixLabors::ZlibStream;

namespace {
// a std::ostream over a growing byte vector (MemoryStream)
struct VecBuf : std::streambuf {
    std::vector<char> v;
    std::streamsize xsputn(const char *s, std::streamsize n) override {
        v.insert(v.end(), s, s + n);
        return n;
    }
    int overflow(int c) override {
        if (c != EOF) v.push_back((char)c);
        return c;
    }
};
struct MemIn : std::streambuf {
    MemIn(const char *p, size_t n) { setg(const_cast<char *>(p), const_cast<char *>(p), const_cast<char *>(p) + n); }
};
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <input file> <level> <reps> [write_bytes]\n", argv[0]);
        return 2;
    }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const int level = atoi(argv[2]), reps = atoi(argv[3]);
    const long long wr = argc > 4 ? atoll(argv[4]) : 0;
    try {
        zs_ctx *ctx = GpuContext::Shared(0);
        std::vector<char> z;
        double best_def = 1e30, best_inf = 1e30;
        for (int r = 0; r < reps + 1; r++) {  // first pass untimed (workspace allocation)
            VecBuf vb;
            vb.v.reserve(data.size() / 2 + 1024);
            std::ostream os(&vb);
            const double t0 = now();
            {
                ZlibOutputStream zo(os, (CompressionLevel)level, ctx);
                if (wr <= 0) {
                    size_t off = 0;  // Write takes an int count: a buffer beyond 2 GiB - 1 would need several
                    while (off < data.size()) {
                        const size_t k = std::min<size_t>(data.size() - off, 0x7FFFFFFF);
                        zo.Write(data.data() + off, 0, (int)k);
                        off += k;
                    }
                } else {
                    for (size_t off = 0; off < data.size(); off += (size_t)wr)
                        zo.Write(data.data() + off, 0, (int)std::min<size_t>((size_t)wr, data.size() - off));
                }
                zo.Dispose();
            }
            const double dt = now() - t0;
            if (r) best_def = std::min(best_def, dt);
            z.swap(vb.v);
        }
        std::vector<uint8_t> back(data.size() + 1);
        bool same = true;
        for (int r = 0; r < reps + 1; r++) {
            MemIn mb(z.data(), z.size());
            std::istream is(&mb);
            const double t0 = now();
            ZlibInputStream zi(is, ctx);
            size_t got = 0;
            for (;;) {
                const int k = zi.Read(back.data(), 0, (int)std::min<size_t>(back.size(), 1 << 20));
                if (k <= 0) break;
                // Read fills from offset 0 of the span it is given; compare and move on (no second copy of the output)
                if (got + (size_t)k > data.size() || memcmp(back.data(), data.data() + got, (size_t)k) != 0) same = false;
                got += (size_t)k;
            }
            const double dt = now() - t0;
            if (r) best_inf = std::min(best_inf, dt);
            if (got != data.size()) same = false;
        }
        printf("{\"input_bytes\": %zu, \"compressed_bytes\": %zu, \"level\": %d, \"write_bytes\": %lld, \"deflate_chunk\": %d, "
               "\"deflate_MBps\": %.1f, \"deflate_ms\": %.2f, \"inflate_MBps\": %.1f, \"inflate_ms\": %.2f, \"roundtrip\": %s}\n",
               data.size(), z.size(), level, wr, ZlibOutputStream::BufferSize, data.size() / best_def / 1e6, best_def * 1e3,
               data.size() / best_inf / 1e6, best_inf * 1e3, same ? "true" : "false");
        return same ? 0 : 1;
    } catch (const std::exception &e) {
        fprintf(stderr, "stream_api_bench: %s\n", e.what());
        return 1;
    }
}
