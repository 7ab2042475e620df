// The reference's xunit tests (tests/ZlibStream.Tests/ZlibStreamTests.Roundtrip.cs:25-125), restated against
// the C++ host mirror (include/zsgpu.hpp) over the C ABI, with the oracle as the byte-level checker.
// Built and run by tests/test_gpu_parity.py::test_cpp_host_mirror on the GPU box.
#include <cstdio>
#include <cstring>
#include <sstream>
#include <vector>

#include "../../include/zsgpu.hpp"
#include "../../oracle/zs_oracle.h"

using namespace SixLabors::ZlibStream;

static std::vector<uint8_t> GetBuffer(int length) {  // new Random(1).NextBytes
    std::vector<uint8_t> d((size_t)length);
    zso_dotnet_random_bytes(1, d.data(), d.size());
    return d;
}
static int fails = 0;
#define CHECK(c)                                              \
    do {                                                      \
        if (!(c)) {                                           \
            printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            fails++;                                          \
        }                                                     \
    } while (0)

static std::vector<uint8_t> oracle(const std::vector<uint8_t> &d, int level, int strategy, const std::vector<size_t> &chunks) {
    std::vector<uint8_t> out(zso_compress_bound(d.size()));
    size_t n = zso_compress_stream(d.data(), d.size(), chunks.empty() ? nullptr : chunks.data(), chunks.size(), level, strategy, 0, 0,
                                   out.data(), out.size(), nullptr);
    out.resize(n);
    return out;
}

int main() {
    const int count = 2 * 4096 * 4, chunk = 2 * 4096;
    std::vector<uint8_t> expected = GetBuffer(count);
    const CompressionLevel levels[] = {CompressionLevel::NoCompression, CompressionLevel::Level1, CompressionLevel::Level2, CompressionLevel::Level3, CompressionLevel::Level4,
                                       CompressionLevel::Level5, CompressionLevel::Level6, CompressionLevel::Level7,
                                       CompressionLevel::BestCompression, CompressionLevel::DefaultCompression};
    const CompressionStrategy strategies[] = {CompressionStrategy::DefaultStrategy, CompressionStrategy::Filtered,
                                              CompressionStrategy::HuffmanOnly, CompressionStrategy::Rle, CompressionStrategy::Fixed};
    for (CompressionLevel level : levels) {
        for (CompressionStrategy strategy : strategies) {
            // EncodeDecode
            std::stringstream compressed;
            {
                ZlibOptions options;
                options.CompressionLevel_ = level;
                options.CompressionStrategy_ = strategy;
                ZlibOutputStream deflate(compressed, options);
                deflate.Write(expected.data(), 0, (int)expected.size());
            }
            std::string z = compressed.str();
            std::vector<uint8_t> ref = oracle(expected, (int)level, (int)strategy, {});
            CHECK(z.size() == ref.size() && memcmp(z.data(), ref.data(), ref.size()) == 0);
            std::vector<uint8_t> actual((size_t)count);
            ZlibInputStream inflate(compressed);
            CHECK(inflate.Read(actual.data(), 0, count) == count);
            CHECK(actual == expected);
        }
        // EncodeDecodePerChunk
        std::stringstream compressed;
        {
            ZlibOutputStream deflate(compressed, level);
            for (int i = 0; i < count; i += chunk) deflate.Write(expected.data(), i, chunk);
        }
        std::string z = compressed.str();
        std::vector<uint8_t> ref = oracle(expected, (int)level, 0, {(size_t)chunk, (size_t)chunk, (size_t)chunk, (size_t)chunk});
        CHECK(z.size() == ref.size() && memcmp(z.data(), ref.data(), ref.size()) == 0);
        std::vector<uint8_t> actual((size_t)count);
        ZlibInputStream inflate(compressed);
        for (int i = 0; i < count; i += chunk) CHECK(inflate.Read(actual.data(), i, chunk) == chunk);
        CHECK(actual == expected);
    }
    // ZlibOptions.FlushMode Partial / Sync / Full: every Write closes its block and is followed by the flush marker
    // (Deflate.cs:583-613); bytes checked against the oracle's literal 512-byte WriteCore loop
    for (FlushMode mode : {FlushMode::PartialFlush, FlushMode::SyncFlush, FlushMode::FullFlush}) {
        for (CompressionLevel level : {CompressionLevel::NoCompression, CompressionLevel::Level1, CompressionLevel::Level6}) {
            std::stringstream compressed;
            {
                ZlibOptions options;
                options.CompressionLevel_ = level;
                options.FlushMode_ = mode;
                ZlibOutputStream deflate(compressed, options);
                for (int i = 0; i < count; i += chunk) deflate.Write(expected.data(), i, chunk);
            }
            std::string z = compressed.str();
            std::vector<uint8_t> ref(zso_compress_bound(expected.size()) + 256);
            const size_t chunks[4] = {(size_t)chunk, (size_t)chunk, (size_t)chunk, (size_t)chunk};
            ref.resize(zso_compress_stream(expected.data(), expected.size(), chunks, 4, (int)level, 0, (int)mode, 0, ref.data(), ref.size(), nullptr));
            CHECK(z.size() == ref.size() && memcmp(z.data(), ref.data(), ref.size()) == 0);
            std::vector<uint8_t> actual((size_t)count);
            ZlibInputStream inflate(compressed);
            CHECK(inflate.Read(actual.data(), 0, count) == count);
            CHECK(actual == expected);
        }
    }
    // error behaviour: what is outside the device path -> ZlibStreamException("deflating: ...")
    try {
        zs_ctx *ctx = nullptr;
        CHECK(zs_ctx_create(0, &ctx) == 0);
        zs_deflate_stream *d = zs_deflate_init(ctx, 6, 0, 12, 8, 0);  // windowBits 12
        CHECK(d != nullptr);
        uint8_t outb[512];
        int32_t availIn = 10, availOut = 512;
        int rc = zs_deflate(d, expected.data(), &availIn, outb, &availOut, 0, nullptr, nullptr, nullptr);
        CHECK(rc == -2 && zs_last_message(d) != nullptr);
        zs_deflate_end(d);
        zs_ctx_destroy(ctx);
    } catch (const ZlibStreamException &e) {
        CHECK(!"unexpected exception");
    }
    // corrupt trailer -> "inflating: incorrect data check" (Inflate.cs:339)
    try {
        std::stringstream s;
        {
            ZlibOutputStream deflate(s, CompressionLevel::Level6);
            deflate.Write(expected.data(), 0, 1000);
        }
        std::string z = s.str();
        z[z.size() - 1] ^= 1;
        std::stringstream bad(z);
        ZlibInputStream inflate(bad);
        uint8_t b[16];
        inflate.Read(b, 0, 16);
        CHECK(!"expected ZlibStreamException");
    } catch (const ZlibStreamException &e) {
        CHECK(std::string(e.what()) == "inflating: incorrect data check");
    }
    printf(fails ? "FAILED %d\n" : "PASS\n", fails);
    return fails ? 1 : 0;
}
