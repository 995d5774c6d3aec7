// PgxNative.cs -- P/Invoke surface of libpgx.so (include/pgx.h), to be added to
// dotnet_src/ImageProcessing.  NOT compiled in this repository (no .NET SDK in the build image);
// it is the literal binding a maintainer adds.  See INTEGRATION.md.
using System;
using System.Runtime.InteropServices;

namespace ImageProcessing.Native;

[StructLayout(LayoutKind.Sequential)]
public struct PgxKeypoint { public int X, Y, FastScore; public float Value; }

[StructLayout(LayoutKind.Sequential)]
public struct PgxPair { public int K1, K2, Dist; }

internal static unsafe class PgxNative
{
    private const string Lib = "pgx"; // libpgx.so next to the executable or on LD_LIBRARY_PATH

    public const int Ok = 0, EDimMismatch = 1, EOobSource = 2, EEmptySet = 3, ECapacity = 4, EBadArg = 5, EHip = 6,
        ENotConfigured = 7, ERccl = 8;   // include/pgx.h: PGX_OK, PGX_E_*
    public const int DistNone = int.MaxValue;                                             // PGX_DIST_NONE
    public const int SrcRgba64 = 0, SrcRgba8 = 1;                                         // PGX_SRC_*
    public const int StageDetect = 0, StageMatchWide = 1, StageMatchRows = 2, StageMatchDone = 3;   // PGX_STAGE_*
    public const int CommIdBytes = 128;                                                   // PGX_COMM_ID_BYTES

    // Exports of include/pgx.h that this binding deliberately leaves out (tests/test_csharp_binding.py holds the list to the
    // header): the caller's-HIP-stream hook and the measurement hooks (a managed host owns no hipStream_t and reads no HIP event
    // times), the version string, the communicator query, and the two host-side helpers whose managed originals the host
    // keeps (Utils.NextGaussianPair, DeWarp.GetDistortionMatrix).
    // OMITTED: pgx_version pgx_set_stream pgx_comm_info pgx_profile_enable pgx_profile_filter pgx_profile_get pgx_profile_reset
    // OMITTED: pgx_profile_serialize pgx_match_stats pgx_debug_counters pgx_make_brief_pairs pgx_build_dewarp_map

    [DllImport(Lib)] public static extern int pgx_ctx_create(int device, out IntPtr ctx);
    [DllImport(Lib)] public static extern void pgx_ctx_destroy(IntPtr ctx);
    [DllImport(Lib)] public static extern IntPtr pgx_last_error(IntPtr ctx);
    [DllImport(Lib)] public static extern int pgx_set_dewarp_map(IntPtr ctx, int* uv, int w, int h);
    [DllImport(Lib)] public static extern int pgx_set_dewarp_coeffs(IntPtr ctx, int w, int h, double* coeffs, int ncoeffs);
    [DllImport(Lib)] public static extern int pgx_get_dewarp_map(IntPtr ctx, int* uvOut, int w, int h);
    [DllImport(Lib)] public static extern int pgx_set_brief_pairs(IntPtr ctx, int* pairs, int p);
    [DllImport(Lib)] public static extern int pgx_set_detect_params(IntPtr ctx, float threshold, int suppressionRadius);
    [DllImport(Lib)] public static extern int pgx_set_capacity(IntPtr ctx, int maxRaw, int maxKeypoints);
    [DllImport(Lib)] public static extern int pgx_set_source_format(IntPtr ctx, int format); // 0 = Rgba64, 1 = Rgba32 bytes (widened x257 on the device)
    [DllImport(Lib)] public static extern int pgx_set_match_chunk(IntPtr ctx, int imagePairsPerChunk);
    [DllImport(Lib)] public static extern int pgx_gate_match(IntPtr ctx, IntPtr other, int stage);  // the same wait inside the next matcher call, behind its init kernel
    [DllImport(Lib)] public static extern int pgx_wait_stage(IntPtr ctx, IntPtr other, int stage);   // PGX_STAGE_*: 0 detect, 1 match wide, 2 match rows, 3 match done
    [DllImport(Lib)] public static extern int pgx_dewarp(IntPtr ctx, ushort* rgba64, int w, int h, ushort* outRgba64);
    [DllImport(Lib)] public static extern int pgx_gray(IntPtr ctx, ushort* rgba64, int w, int h, float* outGray);
    [DllImport(Lib)] public static extern int pgx_fast(IntPtr ctx, float* gray, int w, int h, PgxKeypoint* o, int capacity, out int n);
    [DllImport(Lib)] public static extern int pgx_brief(IntPtr ctx, float* gray, int w, int h, PgxKeypoint* kps, int n, uint* desc);
    [DllImport(Lib)] public static extern int pgx_nms(IntPtr ctx, PgxKeypoint* kps, int n, int w, int h, int* order, out int nOut);
    [DllImport(Lib)] public static extern int pgx_match(IntPtr ctx, uint* d1, int n1, uint* d2, int n2, int words, PgxPair* o);
    // many image pairs, managed arrays, one call: descs[f] -> frame f's descriptors, pairList [m][2], lists back to back in `o`
    [DllImport(Lib)] public static extern int pgx_match_batch(IntPtr ctx, uint** descs, int* counts, int nFrames, int words,
                                                              int* pairList, int nPairs, PgxPair* o, long* outOffsets);
    [DllImport(Lib)] public static extern int pgx_detect(IntPtr ctx, ushort* rgba64, int w, int h, PgxKeypoint* kp, uint* desc,
                                                         int capacity, out int n, out int nRaw);

    // batched, device-resident entry points and the multi-GPU / pose / track-graph additions (include/pgx.h)
    [DllImport(Lib)] public static extern int pgx_detect_batch_dev(IntPtr ctx, void* dRgba64, int f, int w, int h, void* dKp, void* dDesc,
                                                                   void* dCounts, void* dNraw, int capacity);
    [DllImport(Lib)] public static extern int pgx_match_batch_dev(IntPtr ctx, void* dDesc, void* dCounts, int stride, int words,
                                                                  void* dPairlist, int m, int maxCount, void* dOut);
    [DllImport(Lib)] public static extern int pgx_check_status(IntPtr ctx);
    [DllImport(Lib)] public static extern int pgx_comm_unique_id(byte* id128);
    [DllImport(Lib)] public static extern int pgx_comm_init(IntPtr ctx, int rank, int world, byte* id128);
    [DllImport(Lib)] public static extern int pgx_comm_destroy(IntPtr ctx);
    [DllImport(Lib)] public static extern int pgx_allgather_dev(IntPtr ctx, void* dBuf, nuint bytesPerRank);
    [DllImport(Lib)] public static extern int pgx_sequence_step_dev(IntPtr ctx, void* dFramesLocal, int nLocalFrames, int frameSlots, int w, int h,
                                                                    void* dKpLocal, void* dDescAll, void* dCountsAll, void* dNrawLocal, int capacity,
                                                                    void* dPairlistLocal, int nLocalPairs, int pairSlots, void* dOutAll);
    [DllImport(Lib)] public static extern int pgx_fundamental_ransac_dev(IntPtr ctx, void* dKp, void* dMatches, void* dCounts, void* dPairlist, int m,
                                                                         int stride, int nSamples, int pairsPerSample, float threshold, int rankCheck,
                                                                         ulong seed, float* dF, int* dInliers, int* dBestSample);
    [DllImport(Lib)] public static extern int pgx_pose_dev(IntPtr ctx, void* dKp, void* dMatches, void* dCounts, void* dPairlist, int m, int stride,
                                                           float* dF, float* dRt, int* dVotes, int* dBest, float* dPoints);
    [DllImport(Lib)] public static extern int pgx_tracks_create(int* counts, int nFrames, out IntPtr tracks);
    [DllImport(Lib)] public static extern void pgx_tracks_destroy(IntPtr tracks);
    [DllImport(Lib)] public static extern int pgx_tracks_add_pair(IntPtr tracks, int frameA, int frameB, PgxPair* matches, int n, int maxDist);
    [DllImport(Lib)] public static extern int pgx_tracks_finish(IntPtr tracks, int minLen, out int nTracks, out int nNodes);
    [DllImport(Lib)] public static extern int pgx_tracks_get(IntPtr tracks, int* trackOffsets, int* nodes);
    [DllImport(Lib)] public static extern int pgx_tracks_dropped(IntPtr tracks, out int nComponents, out int nNodes);
    // the same graph built on the device over the lists where the matcher / the all-gather left them
    [DllImport(Lib)] public static extern int pgx_tracks_dev(IntPtr ctx, void* dMatches, void* dCounts, void* dPairlist, int m, int f, int stride,
                                                             void* dFrameIds, int nFrames, int maxDist, int minLen, void* dTrackOf,
                                                             void* dOffsets, void* dNodes, void* dSummary);

    /// <summary>Maps a status code back to the exception type the managed implementation throws.</summary>
    public static void Check(IntPtr ctx, int rc)
    {
        if (rc == Ok) return;
        var msg = Marshal.PtrToStringUTF8(pgx_last_error(ctx)) ?? "pgx error";
        throw rc switch
        {
            EDimMismatch or EBadArg => new ArgumentException(msg),          // DeWarp.cs:23, :48
            EOobSource => new IndexOutOfRangeException(msg),                // Matrix.cs:65, :207
            EEmptySet => new ArgumentOutOfRangeException(msg),              // KeypointMatching.cs:61
            ERccl => new System.IO.IOException(msg),                        // a collective or the RCCL library failed (multi-GPU only)
            _ => new InvalidOperationException(msg),
        };
    }
}

/// <summary>One context per GPU; registered as a DI singleton (Program.cs:40-59).</summary>
public sealed class PgxContext : IDisposable
{
    internal IntPtr Handle;
    public PgxContext(int device = 0) => PgxNative.Check(IntPtr.Zero, PgxNative.pgx_ctx_create(device, out Handle));
    public void Dispose() { if (Handle != IntPtr.Zero) { PgxNative.pgx_ctx_destroy(Handle); Handle = IntPtr.Zero; } }
}
