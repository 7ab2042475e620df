// zsgpu.hpp -- C++ host-side mirror of the reference's Stream API over the C ABI (zsgpu.h).
//
// The reference is compiled managed code (C#) and no .NET toolchain exists in the build image, so the host
// side above the C ABI is written in C++: same type names, members, argument meaning and error behaviour as
//   src/ZlibStream/ZlibOutputStream.cs, ZlibInputStream.cs, ZlibOptions.cs, CompressionLevel.cs,
//   CompressionStrategy.cs, FlushMode.cs, CompressionState.cs, ZlibStreamException.cs, ThrowHelper.cs:21-23,
// with System.IO.Stream replaced by std::ostream / std::istream.  Header-only; link with libzsgpu.so.
#pragma once
#include <cstdint>
#include <istream>
#include <optional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "zsgpu.h"

namespace SixLabors {
namespace ZlibStream {

enum class CompressionLevel : int {  // CompressionLevel.cs
    DefaultCompression = -1, Level0 = 0, NoCompression = 0, Level1 = 1, BestSpeed = 1, Level2 = 2, Level3 = 3, Level4 = 4,
    Level5 = 5, Level6 = 6, Level7 = 7, Level8 = 8, Level9 = 9, BestCompression = 9
};
enum class CompressionStrategy : int { DefaultStrategy = 0, Filtered = 1, HuffmanOnly = 2, Rle = 3, Fixed = 4 };
enum class FlushMode : int { NoFlush = 0, PartialFlush = 1, SyncFlush = 2, FullFlush = 3, Finish = 4 };
enum class CompressionState : int {
    ZVERSIONERROR = -6, ZBUFERROR = -5, ZMEMERROR = -4, ZDATAERROR = -3, ZSTREAMERROR = -2, ZERRNO = -1, ZOK = 0, ZSTREAMEND = 1,
    ZNEEDDICT = 2
};

struct ZlibOptions {  // ZlibOptions.cs
    std::optional<CompressionLevel> CompressionLevel_;
    CompressionStrategy CompressionStrategy_ = CompressionStrategy::DefaultStrategy;
    FlushMode FlushMode_ = FlushMode::NoFlush;
};

class ZlibStreamException : public std::runtime_error {  // ZlibStreamException.cs
public:
    explicit ZlibStreamException(const std::string &m) : std::runtime_error(m) {}
};

// One engine context per GPU, shared by the streams of the process (zs_ctx is not thread-safe).
class GpuContext {
public:
    static zs_ctx *Shared(int device = 0) {
        static GpuContext g(device);
        return g.ctx_;
    }
private:
    explicit GpuContext(int device) {
        if (zs_ctx_create(device, &ctx_) != ZS_OK || !ctx_)
            throw ZlibStreamException("no usable MI355X / HIP device: the engine has no CPU fallback");
    }
    ~GpuContext() { zs_ctx_destroy(ctx_); }
    zs_ctx *ctx_ = nullptr;
};

// ZlibOutputStream.cs: write-only stream that deflates into BaseStream.
class ZlibOutputStream {
public:
    static constexpr int BufferSize = 512;  // ZlibOutputStream.cs: chunkBuffer

    ZlibOutputStream(std::ostream &output, CompressionLevel level, zs_ctx *ctx = nullptr)
        : ZlibOutputStream(output, ZlibOptions{level, CompressionStrategy::DefaultStrategy, FlushMode::NoFlush}, ctx) {}

    ZlibOutputStream(std::ostream &output, const ZlibOptions &options, zs_ctx *ctx = nullptr) : BaseStream(output), Options(options) {
        if (!Options.CompressionLevel_) throw std::invalid_argument("inflate mode of ZlibOutputStream is not supported");
        int level = (int)*Options.CompressionLevel_;
        // Deflate..ctor throws ArgumentOutOfRangeException (Deflate.cs:258-281)
        z_ = zs_deflate_init(ctx ? ctx : GpuContext::Shared(), level, (int)Options.CompressionStrategy_, 15, 8, ZS_HASH_CRC32C);
        if (!z_) throw std::out_of_range("level / strategy");
    }
    ZlibOutputStream(const ZlibOutputStream &) = delete;
    ZlibOutputStream &operator=(const ZlibOutputStream &) = delete;
    ~ZlibOutputStream() {
        try {
            Dispose();
        } catch (...) {
        }
    }

    std::ostream &BaseStream;
    ZlibOptions Options;
    long long TotalIn() const { return totalIn_; }
    long long TotalOut() const { return totalOut_; }
    bool CanRead() const { return false; }
    bool CanSeek() const { return false; }
    bool CanWrite() const { return true; }

    void WriteByte(uint8_t value) { Write(&value, 0, 1); }

    // WriteCore (ZlibOutputStream.cs:125-168)
    void Write(const uint8_t *buffer, int offset, int count) {
        if (!buffer && count) throw std::invalid_argument("buffer");
        if (count == 0) return;
        Loop(buffer + offset, count, (int)Options.FlushMode_);
    }
    void Write(const std::vector<uint8_t> &buffer) { Write(buffer.data(), 0, (int)buffer.size()); }

    void Flush() { BaseStream.flush(); }

    // Finish (ZlibOutputStream.cs:213-256) + Dispose (:186-211)
    void Dispose() {
        if (isDisposed_) return;
        isDisposed_ = true;
        try {
            if (!isFinished_) {
                Loop(nullptr, 0, ZS_FINISH);
                isFinished_ = true;
                Flush();
            }
        } catch (...) {
            zs_deflate_end(z_);
            z_ = nullptr;
            throw;
        }
        zs_deflate_end(z_);
        z_ = nullptr;
    }

private:
    void Loop(const uint8_t *in, int count, int flush) {
        int32_t availIn = count;
        const uint8_t *next = in;
        for (;;) {
            int32_t availOut = BufferSize;
            int32_t before = availIn;
            int state = zs_deflate(z_, next, &availIn, chunk_, &availOut, flush, &adler_, &totalIn_, &totalOut_);
            next += before - availIn;
            if (state != ZS_OK && state != ZS_STREAM_END) {
                const char *m = zs_last_message(z_);
                throw ZlibStreamException(std::string("deflating: ") + (m ? m : ""));  // ThrowHelper.cs:21-23
            }
            if (BufferSize - availOut > 0) BaseStream.write((const char *)chunk_, BufferSize - availOut);
            if (state == ZS_STREAM_END) break;
            if (!(availIn > 0 || availOut == 0)) break;
        }
    }
    zs_deflate_stream *z_ = nullptr;
    uint8_t chunk_[BufferSize];
    uint32_t adler_ = 1;
    int64_t totalIn_ = 0, totalOut_ = 0;
    bool isFinished_ = false, isDisposed_ = false;
};

// ZlibInputStream.cs: read-only stream that inflates BaseStream.  The device decodes whole streams: the
// first Read drains the base stream, inflates it on the GPU and later Reads are served from the result.
class ZlibInputStream {
public:
    explicit ZlibInputStream(std::istream &input, zs_ctx *ctx = nullptr) : BaseStream(input), ctx_(ctx ? ctx : GpuContext::Shared()) {}
    std::istream &BaseStream;
    bool CanRead() const { return true; }
    bool CanWrite() const { return false; }

    // returns the number of bytes read, 0 at the end of the stream
    int Read(uint8_t *buffer, int offset, int count) {
        if (!decoded_) Decode();
        size_t n = data_.size() - pos_;
        if (n > (size_t)count) n = (size_t)count;
        std::copy(data_.begin() + (long)pos_, data_.begin() + (long)(pos_ + n), buffer + offset);
        pos_ += n;
        return (int)n;
    }
    int ReadByte() {
        uint8_t b;
        return Read(&b, 0, 1) == 1 ? b : -1;
    }

private:
    void Decode() {
        std::vector<uint8_t> z((std::istreambuf_iterator<char>(BaseStream)), std::istreambuf_iterator<char>());
        int64_t cap = (int64_t)z.size() * 4 + 65536;
        for (;;) {
            data_.resize((size_t)cap);
            const void *in = z.data();
            void *out = data_.data();
            int64_t inLen = (int64_t)z.size(), outLen = 0;
            int status = 0;
            int rc = zs_inflate_batch(ctx_, 1, &in, &inLen, &out, &cap, &outLen, &status);
            if (rc == ZS_OK) {
                data_.resize((size_t)outLen);
                break;
            }
            std::string m = zs_ctx_last_error(ctx_);
            if (status == ZS_BUF_ERROR && m == "buffer error" && outLen >= cap && cap < (1LL << 31)) {
                cap *= 4;  // the output did not fit: retry with a larger buffer
                continue;
            }
            throw ZlibStreamException("inflating: " + m);  // ThrowHelper.cs:21-23
        }
        decoded_ = true;
    }
    zs_ctx *ctx_;
    std::vector<uint8_t> data_;
    size_t pos_ = 0;
    bool decoded_ = false;
};

}  // namespace ZlibStream
}  // namespace SixLabors
