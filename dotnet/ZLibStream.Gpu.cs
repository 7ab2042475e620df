// ZLibStream.Gpu.cs -- drop-in replacement for src/ZlibStream/ZlibStream.cs of SixLabors.ZlibStream.
//
// Same type name, namespace, accessibility and members as the reference's internal z_stream facade, so that
// ZlibOutputStream.cs (WriteCore :125-168, Finish :213-256) and ZlibInputStream.cs (ReadCore :133-186) compile and run
// against it byte for byte unchanged.  The engine objects behind the facade (the managed Deflate / Inflate classes)
// are replaced by handles into libzsgpu.so; DeflateState / InflateState are therefore opaque here.
//
// What is different from the managed facade, and why:
//   * ReadBuffer is gone: the native engine copies the caller's span during the call (same ownership rule:
//     NextIn / NextOut are only valid inside WriteCore / ReadCore's `fixed` block).
//   * DeflateParams / DeflateSetDictionary / InflateSetDictionary / InflateSync have no public caller in the reference
//     (internal-only, ZlibStream.cs:125-190); they return ZSTREAMERROR here instead of silently doing something else.
using System;
using System.Runtime.InteropServices;
#if NETCOREAPP3_0_OR_GREATER
using System.Runtime.Intrinsics.X86;
#endif

namespace SixLabors.ZlibStream
{
    /// <summary>
    /// The zlib stream class (GPU engine behind the reference's facade).
    /// </summary>
    internal sealed unsafe class ZLibStream : IDisposable
    {
        private const int MAXWBITS = 15; // 32K LZ77 window
        private const int DEFWBITS = MAXWBITS;
        private const int DEFMEMLEVEL = 8;
        private readonly GpuContext context = GpuContext.Shared;
        private IntPtr deflateHandle;
        private IntPtr inflateHandle;
        private bool isDisposed;

        public ZLibStream(ZlibOptions options)
        {
            if (options.CompressionLevel is null)
            {
                this.InflateInit();
            }
            else
            {
                this.Compress = true;
                this.DeflateInit(options);
            }
        }

        public byte* NextIn { get; set; }

        public byte* NextOut { get; set; }

        public int NextInIndex { get; set; }

        public int AvailableIn { get; set; }

        public long TotalIn { get; set; }

        public int NextOutIndex { get; set; }

        public int AvailableOut { get; set; }

        public long TotalOut { get; set; }

        public string Message { get; set; }

        public uint Adler { get; set; } = 1;

        public int DataType { get; set; }

        public bool Compress { get; }

        /// <summary>Gets the native deflate stream (zs_deflate_stream*), or zero.</summary>
        public IntPtr DeflateState => this.deflateHandle;

        /// <summary>Gets the native inflate stream (zs_inflate_stream*), or zero.</summary>
        public IntPtr InflateState => this.inflateHandle;

        public void InflateInit() => this.InflateInit(DEFWBITS);

        public void InflateInit(int windowBits)
        {
            lock (this.context.Gate)
            {
                this.inflateHandle = ZsGpu.zs_inflate_init(this.context.Handle, windowBits);
            }

            if (this.inflateHandle == IntPtr.Zero)
            {
                // Inflate..ctor rejects window sizes outside 8..15 with ZSTREAMERROR (Inflate.cs:76-96)
                throw new ArgumentOutOfRangeException(nameof(windowBits));
            }
        }

        public CompressionState Inflate(FlushMode strategy)
        {
            if (this.inflateHandle == IntPtr.Zero)
            {
                return CompressionState.ZSTREAMERROR;
            }

            int availIn = this.AvailableIn, availOut = this.AvailableOut;
            uint adler = this.Adler;
            long tin = this.TotalIn, tout = this.TotalOut;
            int state;
            lock (this.context.Gate)
            {
                state = ZsGpu.zs_inflate(this.inflateHandle, this.NextIn + this.NextInIndex, ref availIn, this.NextOut + this.NextOutIndex,
                                         ref availOut, (int)strategy, ref adler, ref tin, ref tout);
                this.Message = Marshal.PtrToStringAnsi(ZsGpu.zs_inflate_message(this.inflateHandle));
            }

            this.Advance(availIn, availOut, adler, tin, tout);
            return (CompressionState)state;
        }

        public CompressionState InflateSync() => CompressionState.ZSTREAMERROR;

        public CompressionState InflateSetDictionary(byte[] dictionary, int dictLength) => CompressionState.ZSTREAMERROR;

        public void DeflateInit(ZlibOptions options) => this.DeflateInit(options, MAXWBITS);

        public void DeflateInit(ZlibOptions options, int windowBits)
        {
            // Which UpdateHash the managed build would run on this machine decides the bytes (Deflate.Intrinsics.cs:295-307):
            // 0 = Sse42.Crc32, 1 = the multiplicative fallback.
            int hashVariant = 1;
#if NETCOREAPP3_0_OR_GREATER
            hashVariant = Sse42.IsSupported ? 0 : 1;
#endif
            int level = (int)options.CompressionLevel.GetValueOrDefault();
            lock (this.context.Gate)
            {
                this.deflateHandle = ZsGpu.zs_deflate_init(this.context.Handle, level, (int)options.CompressionStrategy, windowBits, DEFMEMLEVEL, hashVariant);
            }

            if (this.deflateHandle == IntPtr.Zero)
            {
                // Deflate..ctor: ArgumentOutOfRangeException for level / strategy / windowBits / memLevel (Deflate.cs:258-281)
                throw new ArgumentOutOfRangeException(nameof(options));
            }
        }

        public CompressionState Deflate(FlushMode flush)
        {
            if (this.deflateHandle == IntPtr.Zero)
            {
                return CompressionState.ZSTREAMERROR;
            }

            int availIn = this.AvailableIn, availOut = this.AvailableOut;
            uint adler = this.Adler;
            long tin = this.TotalIn, tout = this.TotalOut;
            int state;
            lock (this.context.Gate)
            {
                state = ZsGpu.zs_deflate(this.deflateHandle, this.NextIn + this.NextInIndex, ref availIn, this.NextOut + this.NextOutIndex,
                                         ref availOut, (int)flush, ref adler, ref tin, ref tout);
                this.Message = Marshal.PtrToStringAnsi(ZsGpu.zs_last_message(this.deflateHandle));
            }

            this.Advance(availIn, availOut, adler, tin, tout);
            return (CompressionState)state;
        }

        public CompressionState DeflateParams(CompressionLevel level, CompressionStrategy strategy) => CompressionState.ZSTREAMERROR;

        public CompressionState DeflateSetDictionary(byte[] dictionary, int dictLength) => CompressionState.ZSTREAMERROR;

        public void Dispose()
        {
            if (this.isDisposed)
            {
                return;
            }

            this.isDisposed = true;
            lock (this.context.Gate)
            {
                if (this.deflateHandle != IntPtr.Zero)
                {
                    ZsGpu.zs_deflate_end(this.deflateHandle);
                    this.deflateHandle = IntPtr.Zero;
                }

                if (this.inflateHandle != IntPtr.Zero)
                {
                    ZsGpu.zs_inflate_end(this.inflateHandle);
                    this.inflateHandle = IntPtr.Zero;
                }
            }
        }

        // the cursor fields move exactly as the managed engine moves them (ZlibStream.cs:197-222, Deflate.cs:828-854)
        private void Advance(int availIn, int availOut, uint adler, long totalIn, long totalOut)
        {
            this.NextInIndex += this.AvailableIn - availIn;
            this.AvailableIn = availIn;
            this.NextOutIndex += this.AvailableOut - availOut;
            this.AvailableOut = availOut;
            this.Adler = adler;
            this.TotalIn = totalIn;
            this.TotalOut = totalOut;
        }
    }
}
