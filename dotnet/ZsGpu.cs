// ZsGpu.cs -- P/Invoke surface of libzsgpu.so (include/zsgpu.h), the MI355X deflate / inflate engine.
//
// One declaration per C entry point; the comment on each names the managed code of SixLabors.ZlibStream it stands in
// for.  Nothing here is used by callers directly: ZLibStream.Gpu.cs (the replacement for the internal z_stream
// facade, src/ZlibStream/ZlibStream.cs) is the only consumer, and ZlibOutputStream.cs / ZlibInputStream.cs of the
// reference compile against it unchanged.
using System;
using System.Runtime.InteropServices;

namespace SixLabors.ZlibStream
{
    internal static unsafe class ZsGpu
    {
        // libzsgpu.so next to the assembly or on LD_LIBRARY_PATH (it needs libamdhip64.so from ROCm at run time)
        private const string Lib = "zsgpu";

        // ---- engine context: one per GPU, owns the reusable HBM workspace (no managed counterpart: the managed
        //      engine rents its buffers from ArrayPool, Deflate.Buffers.cs)
        [DllImport(Lib)] public static extern int zs_ctx_create(int device, out IntPtr ctx);
        [DllImport(Lib)] public static extern void zs_ctx_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr zs_ctx_last_error(IntPtr ctx);
        [DllImport(Lib)] public static extern int zs_device_count();
        // counters for tests and diagnostics ("lit_engine_bytes", "fast_rounds", "lit_fallbacks", ...; -1: no such counter)
        [DllImport(Lib)] public static extern long zs_ctx_counter(IntPtr ctx, [MarshalAs(UnmanagedType.LPStr)] string name);

        // ---- Deflate..ctor (Deflate.cs:228-310); IntPtr.Zero where the ctor throws ArgumentOutOfRangeException
        [DllImport(Lib)] public static extern IntPtr zs_deflate_init(IntPtr ctx, int level, int strategy, int windowBits, int memLevel, int hashVariant);

        // ---- Deflate.Compress (Deflate.cs:436-636) as ZLibStream.Deflate(FlushMode) calls it (ZlibStream.cs:164-167)
        [DllImport(Lib)] public static extern int zs_deflate(IntPtr s, byte* nextIn, ref int availIn, byte* nextOut, ref int availOut, int flush,
                                                             ref uint adler, ref long totalIn, ref long totalOut);

        // ---- Deflate.Dispose / ZLibStream.Message
        [DllImport(Lib)] public static extern void zs_deflate_end(IntPtr s);
        [DllImport(Lib)] public static extern IntPtr zs_last_message(IntPtr s);

        // ---- Inflate..ctor (Inflate.cs:76-96), Inflate.Decompress (Inflate.cs:103-357) as ZLibStream.Inflate(FlushMode)
        //      calls it (ZlibStream.cs:119-122), Inflate.Dispose
        [DllImport(Lib)] public static extern IntPtr zs_inflate_init(IntPtr ctx, int windowBits);
        [DllImport(Lib)] public static extern int zs_inflate(IntPtr s, byte* nextIn, ref int availIn, byte* nextOut, ref int availOut, int flush,
                                                             ref uint adler, ref long totalIn, ref long totalOut);
        [DllImport(Lib)] public static extern void zs_inflate_end(IntPtr s);
        [DllImport(Lib)] public static extern IntPtr zs_inflate_message(IntPtr s);

        // ---- throughput entry points for callers that hold many independent buffers (PNG scanline groups, tiles ...):
        //      each buffer is compressed exactly as `using (var s = new ZlibOutputStream(dst, level)) s.Write(buf)` does
        //      (DeflateCorpusBenchmark.cs:86-100); the _multi forms shard the buffers over several contexts / GPUs
        [DllImport(Lib)] public static extern long zs_deflate_bound(long n);
        [DllImport(Lib)] public static extern int zs_deflate_batch(IntPtr ctx, int n, IntPtr* input, long* inLen, IntPtr* output, long* outCap,
                                                                   long* outLen, int* status, int level, int strategy, int hashVariant);
        [DllImport(Lib)] public static extern int zs_inflate_batch(IntPtr ctx, int n, IntPtr* input, long* inLen, IntPtr* output, long* outCap,
                                                                   long* outLen, int* status);
        [DllImport(Lib)] public static extern int zs_deflate_batch_multi(IntPtr* ctxs, int nCtx, int n, IntPtr* input, long* inLen, IntPtr* output,
                                                                         long* outCap, long* outLen, int* status, int level, int strategy, int hashVariant);
        [DllImport(Lib)] public static extern int zs_inflate_batch_multi(IntPtr* ctxs, int nCtx, int n, IntPtr* input, long* inLen, IntPtr* output,
                                                                         long* outCap, long* outLen, int* status);
        // ---- device-resident forms (buffers already in HBM: an encoder whose image lives on the GPU): one stream written in
        //      several NoFlush Writes (writeEnds: the cumulative Write ends, a host array), and the multi-GPU batch over
        //      device pointers (partOf[i]: the context whose GPU holds buffer i, e.g. from zs_partition)
        [DllImport(Lib)] public static extern int zs_partition(long* sizes, int n, int nParts, int* partOf);
        [DllImport(Lib)] public static extern int zs_deflate_writes_device(IntPtr ctx, IntPtr input, long inLen, long* writeEnds, long nWrites, IntPtr output,
                                                                           long outCap, long* outLen, int level, int strategy, int hashVariant, IntPtr hipStream);
        [DllImport(Lib)] public static extern int zs_deflate_batch_multi_device(IntPtr* ctxs, int nCtx, int n, IntPtr* input, long* inLen, IntPtr* output,
                                                                                long* outCap, long* outLen, int* status, int* partOf, int level,
                                                                                int strategy, int hashVariant);
        [DllImport(Lib)] public static extern int zs_inflate_batch_multi_device(IntPtr* ctxs, int nCtx, int n, IntPtr* input, long* inLen, IntPtr* output,
                                                                                long* outCap, long* outLen, int* status, int* partOf);
        // bytes fed behind a stream's trailer before its end was seen (the engine looks for the end now and then)
        [DllImport(Lib)] public static extern long zs_inflate_surplus(IntPtr s, IntPtr* p);
    }

    /// <summary>
    /// The process-wide engine context of one GPU.  A zs_ctx is not thread-safe (like a managed Deflate instance), so
    /// calls into it are serialised by <see cref="Gate"/>; streams on different GPUs use different contexts.
    /// </summary>
    internal sealed class GpuContext : IDisposable
    {
        private static readonly object InitLock = new object();
        private static GpuContext shared;

        private GpuContext(int device)
        {
            int rc = ZsGpu.zs_ctx_create(device, out IntPtr h);
            if (rc != 0 || h == IntPtr.Zero)
            {
                // there is deliberately no managed fallback: a silent CPU path would hide a mis-deployed library
                throw new ZlibStreamException("no usable MI355X / HIP device (zs_ctx_create returned " + rc + ")");
            }

            this.Handle = h;
        }

        public IntPtr Handle { get; private set; }

        public object Gate { get; } = new object();

        public static GpuContext Shared
        {
            get
            {
                lock (InitLock)
                {
                    if (shared is null)
                    {
                        string dev = Environment.GetEnvironmentVariable("ZSGPU_DEVICE");
                        shared = new GpuContext(string.IsNullOrEmpty(dev) ? 0 : int.Parse(dev));
                    }

                    return shared;
                }
            }
        }

        public void Dispose()
        {
            if (this.Handle != IntPtr.Zero)
            {
                ZsGpu.zs_ctx_destroy(this.Handle);
                this.Handle = IntPtr.Zero;
            }
        }
    }
}
