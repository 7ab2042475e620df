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
        // ZlibStream.cs:18-29: a null level means inflate mode -- the stream inflates what is written to it
        compress_ = Options.CompressionLevel_.has_value();
        if (compress_) {
            // Deflate..ctor throws ArgumentOutOfRangeException (Deflate.cs:258-281)
            z_ = zs_deflate_init(ctx ? ctx : GpuContext::Shared(), (int)*Options.CompressionLevel_, (int)Options.CompressionStrategy_, 15, 8,
                                 ZS_HASH_CRC32C);
            if (!z_) throw std::out_of_range("level / strategy");
        } else {
            zi_ = zs_inflate_init(ctx ? ctx : GpuContext::Shared(), 15);
            if (!zi_) throw std::out_of_range("windowBits");
        }
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
            End();
            throw;
        }
        End();
    }

private:
    void Loop(const uint8_t *in, int count, int flush) {
        int32_t availIn = count;
        const uint8_t *next = in;
        for (;;) {
            int32_t availOut = BufferSize;
            int32_t before = availIn;
            int state = compress_ ? zs_deflate(z_, next, &availIn, chunk_, &availOut, flush, &adler_, &totalIn_, &totalOut_)
                                  : zs_inflate(zi_, next, &availIn, chunk_, &availOut, flush, &adler_, &totalIn_, &totalOut_);
            next += before - availIn;
            if (state != ZS_OK && state != ZS_STREAM_END) {
                const char *m = compress_ ? zs_last_message(z_) : zs_inflate_message(zi_);
                throw ZlibStreamException(std::string(compress_ ? "deflating: " : "inflating: ") + (m ? m : ""));  // ThrowHelper.cs:21-23
            }
            if (BufferSize - availOut > 0) BaseStream.write((const char *)chunk_, BufferSize - availOut);
            if (!compress_ && availIn == 0 && availOut == 0 && flush != ZS_FINISH) break;  // ZlibOutputStream.cs:155-158
            if (state == ZS_STREAM_END) break;
            if (!(availIn > 0 || availOut == 0)) break;
        }
    }
    void End() {
        if (z_) zs_deflate_end(z_);
        if (zi_) zs_inflate_end(zi_);
        z_ = nullptr, zi_ = nullptr;
    }
    zs_deflate_stream *z_ = nullptr;
    zs_inflate_stream *zi_ = nullptr;
    bool compress_ = true;
    uint8_t chunk_[BufferSize];
    uint32_t adler_ = 1;
    int64_t totalIn_ = 0, totalOut_ = 0;
    bool isFinished_ = false, isDisposed_ = false;
};

// ZlibInputStream.cs: read-only stream that inflates BaseStream.  The device decodes whole streams: the
// first Read drains the base stream, inflates it on the GPU and later Reads are served from the result.
class ZlibInputStream {
public:
    // ZlibInputStream.cs:29-76: 8 KiB chunk buffer, inflate mode
    explicit ZlibInputStream(std::istream &input, zs_ctx *ctx = nullptr)
        : BaseStream(input), ctx_(ctx ? ctx : GpuContext::Shared()), z_(zs_inflate_init(ctx_, 15)), chunk_(8192) {
        if (!z_) throw std::out_of_range("windowBits");
    }
    ~ZlibInputStream() { zs_inflate_end(z_); }
    ZlibInputStream(const ZlibInputStream &) = delete;
    ZlibInputStream &operator=(const ZlibInputStream &) = delete;
    std::istream &BaseStream;
    bool CanRead() const { return true; }
    bool CanWrite() const { return false; }
    int64_t TotalIn() const { return totalIn_; }
    int64_t TotalOut() const { return totalOut_; }

    // ReadCore (ZlibInputStream.cs:133-186): refill the chunk buffer when it is empty, call Inflate while the caller's
    // buffer has room and the state is ZOK.  Returns the number of bytes read, 0 at the end of the stream.
    int Read(uint8_t *buffer, int offset, int count) {
        if (count == 0) return 0;
        int32_t availOut = count;
        uint8_t *nextOut = buffer + offset;
        int state;
        do {
            if (availIn_ == 0 && !noMoreInput_) {
                BaseStream.read(reinterpret_cast<char *>(chunk_.data()), (std::streamsize)chunk_.size());
                availIn_ = (int32_t)BaseStream.gcount();
                nextIn_ = 0;
            }
            const int32_t inBefore = availIn_, outBefore = availOut;
            state = zs_inflate(z_, chunk_.data() + nextIn_, &availIn_, nextOut, &availOut, ZS_NO_FLUSH, &adler_, &totalIn_, &totalOut_);
            nextIn_ += inBefore - availIn_;
            nextOut += outBefore - availOut;
            if (state != ZS_OK && state != ZS_STREAM_END) {
                const char *m = zs_inflate_message(z_);
                throw ZlibStreamException(std::string("inflating: ") + (m ? m : ""));  // ThrowHelper.cs:21-23
            }
        } while (availOut > 0 && state == ZS_OK);
        return count - availOut;
    }
    int ReadByte() {
        uint8_t b;
        return Read(&b, 0, 1) == 1 ? b : -1;
    }

private:
    zs_ctx *ctx_;
    zs_inflate_stream *z_;
    std::vector<uint8_t> chunk_;
    int32_t availIn_ = 0, nextIn_ = 0;
    bool noMoreInput_ = false;
    uint32_t adler_ = 1;
    int64_t totalIn_ = 0, totalOut_ = 0;
};

}  // namespace ZlibStream
}  // namespace SixLabors
