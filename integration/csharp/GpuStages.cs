// GpuStages.cs -- drop-in replacements for the four concrete singletons of dotnet_src/ImageProcessing
// (Program.cs:42-55 registrations).  Same public method names and exceptions as the managed classes;
// the arithmetic happens in libpgx.so.  NOT compiled here (no .NET SDK in the build image).
//
// Required host edit (SURVEY H7): Keypoint gets one more constructor that takes a precomputed descriptor,
//   public Keypoint(int imageId, Coordinate coordinate, int fastScore, BigInteger brief, Grayscale value)
// because its only constructor recomputes BRIEF from the image (Keypoint.cs:17-27).
using System;
using System.Collections.Generic;
using System.Linq;
using System.Numerics;
using System.Runtime.InteropServices;
using ImageProcessing.Abstractions;
using ImageProcessing.Options;
using Images.Abstractions.Pixels;
using LinearAlgebra;
using Microsoft.Extensions.Options;

namespace ImageProcessing.Native;

/// <summary>Replaces DeWarp.ApplyDistortionMat (DeWarp.cs:19-37).  The distortion table is handed to the library ONCE,
/// in Initialize() -- the place where DeWarpTransformStepFactory forces its Lazy&lt;Matrix&lt;Uv&gt;&gt;
/// (DeWarpTransformStepFactory.cs:23,26-31) -- never per image.</summary>
public sealed unsafe class GpuDeWarp : IInitializable
{
    private readonly PgxContext _ctx;
    private readonly DeWarp _managed;               // GetDistortionMatrix stays managed (init-time, DeWarp.cs:39-107)
    private readonly DeWarpOptions _options;
    private bool _initialized;

    public GpuDeWarp(PgxContext ctx, DeWarp managed, IOptions<DeWarpOptions> options)
    {
        _ctx = ctx; _managed = managed; _options = options.Value;
    }

    /// <summary>IInitializable.Initialize (PipelinesV3/IInitializable.cs:3-6): build the table, upload it once.</summary>
    public void Initialize()
    {
        if (_initialized) return;
        var map = _managed.GetDistortionMatrix();
        int w = map.Dimensions.Width, h = map.Dimensions.Height;
        var uv = new int[w * h * 2];                           // row-major (U, V)
        for (ushort y = 0; y < h; y++)
            for (ushort x = 0; x < w; x++)
            { var m = map[x, y]; uv[(y * w + x) * 2] = m.U; uv[(y * w + x) * 2 + 1] = m.V; }
        fixed (int* p = uv)
            PgxNative.Check(_ctx.Handle, PgxNative.pgx_set_dewarp_map(_ctx.Handle, p, w, h));
        // alternative without the managed cubic solver and the upload: the same table built on the device,
        //   fixed (double* k = _options.DistortionCoefficients) PgxNative.pgx_set_dewarp_coeffs(_ctx.Handle, w, h, k, 5);
        _initialized = true;
    }

    /// <param name="rgba64">ImageSharp's contiguous Image&lt;Rgba64&gt; pixel memory, row-major [H][W][4].</param>
    public void ApplyDistortionMat(ReadOnlySpan<ushort> rgba64, int width, int height, Span<ushort> output)
    {
        if (!_initialized) Initialize();
        fixed (ushort* src = rgba64) fixed (ushort* dst = output)
            PgxNative.Check(_ctx.Handle, PgxNative.pgx_dewarp(_ctx.Handle, src, width, height, dst)); // ArgumentException / IndexOutOfRangeException
    }
}

/// <summary>Replaces KeypointDetection.Detect + the RedundantKeypointEliminator block that follows it
/// (KeyPointDetectionTransformStepFactory.cs:33-35, RedundantKeypointEliminatorTransformStepFactory.cs:34-36).</summary>
public sealed unsafe class GpuKeypointPipeline
{
    private readonly PgxContext _ctx;
    private readonly int _words;

    public GpuKeypointPipeline(PgxContext ctx, IOptions<KeypointDetectionOptions> kd,
                               IOptions<RedundantKeypointEliminationOptions> nms,
                               IReadOnlyList<(Coordinate, Coordinate)> gaussianKeypairs)
    {
        _ctx = ctx;
        var flat = new int[gaussianKeypairs.Count * 4];
        for (var i = 0; i < gaussianKeypairs.Count; i++)
        {
            flat[4 * i] = gaussianKeypairs[i].Item1.X; flat[4 * i + 1] = gaussianKeypairs[i].Item1.Y;
            flat[4 * i + 2] = gaussianKeypairs[i].Item2.X; flat[4 * i + 3] = gaussianKeypairs[i].Item2.Y;
        }
        fixed (int* p = flat) PgxNative.Check(ctx.Handle, PgxNative.pgx_set_brief_pairs(ctx.Handle, p, gaussianKeypairs.Count));
        PgxNative.Check(ctx.Handle, PgxNative.pgx_set_detect_params(ctx.Handle, kd.Value.Threshold, nms.Value.SuppressionRadius));
        _words = (gaussianKeypairs.Count + 31) / 32;
    }

    /// <summary>dewarp -> gray -> Detect -> EliminateRedundantKeypoints for one decoded image.</summary>
    public List<Keypoint> DetectDenoised(ReadOnlySpan<ushort> rgba64, int width, int height, int capacity = 16384)
    {
        var kp = new PgxKeypoint[capacity];
        var desc = new uint[capacity * _words];
        int n, nRaw;
        fixed (ushort* src = rgba64) fixed (PgxKeypoint* k = kp) fixed (uint* d = desc)
            PgxNative.Check(_ctx.Handle, PgxNative.pgx_detect(_ctx.Handle, src, width, height, k, d, capacity, out n, out nRaw));
        var result = new List<Keypoint>(n);
        for (var i = 0; i < n; i++)
        {
            // descriptor words are the BigInteger's little-endian limbs (include/pgx.h conventions)
            var bytes = MemoryMarshal.AsBytes(desc.AsSpan(i * _words, _words));
            var brief = new BigInteger(bytes, isUnsigned: true, isBigEndian: false);
            result.Add(new Keypoint(0, new Coordinate { X = kp[i].X, Y = kp[i].Y }, kp[i].FastScore, brief,
                                    new Grayscale { K = kp[i].Value }));
        }
        return result;
    }
}

/// <summary>Replaces KeypointMatching.MatchKeypoints (KeypointMatching.cs:14-69; call site TestService.cs:96).</summary>
public sealed unsafe class GpuKeypointMatching
{
    private readonly PgxContext _ctx;
    public GpuKeypointMatching(PgxContext ctx) => _ctx = ctx;

    public List<KeypointPair> MatchKeypoints(List<Keypoint> keypoints1, List<Keypoint> keypoints2, int words = 8)
    {
        var d1 = Pack(keypoints1, words);
        var d2 = Pack(keypoints2, words);
        var pairs = new PgxPair[Math.Max(1, keypoints1.Count)];
        fixed (uint* a = d1) fixed (uint* b = d2) fixed (PgxPair* o = pairs)
            PgxNative.Check(_ctx.Handle, PgxNative.pgx_match(_ctx.Handle, a, keypoints1.Count, b, keypoints2.Count, words, o)); // ArgumentOutOfRangeException
        var result = new List<KeypointPair>(keypoints1.Count);
        for (var i = 0; i < keypoints1.Count; i++)
            result.Add(new KeypointPair { Keypoint1 = keypoints1[pairs[i].K1], Keypoint2 = keypoints2[pairs[i].K2], Distance = pairs[i].Dist });
        return result;
    }

    /// <summary>MatchKeypoints for many image pairs in ONE native call (pgx_match_batch): one upload, one enqueue, one download.
    /// One pair per call (above) leaves the GPU idle between calls; a host that has several keypoint lists should use this.</summary>
    public List<List<KeypointPair>> MatchKeypointsBatch(List<List<Keypoint>> frames, List<(int A, int B)> pairs, int words = 8)
    {
        var packed = frames.Select(f => Pack(f, words)).ToArray();
        var counts = frames.Select(f => f.Count).ToArray();
        var pl = pairs.SelectMany(p => new[] { p.A, p.B }).ToArray();
        var total = pairs.Sum(p => frames[p.A].Count);
        var o = new PgxPair[Math.Max(1, total)];
        var offs = new long[pairs.Count + 1];
        var handles = packed.Select(a => GCHandle.Alloc(a, GCHandleType.Pinned)).ToArray();
        try
        {
            var ptrs = handles.Select(h => (IntPtr)h.AddrOfPinnedObject()).ToArray();
            fixed (IntPtr* pp = ptrs) fixed (int* c = counts) fixed (int* l = pl) fixed (PgxPair* po = o) fixed (long* off = offs)
                PgxNative.Check(_ctx.Handle, PgxNative.pgx_match_batch(_ctx.Handle, (uint**)pp, c, frames.Count, words, l, pairs.Count, po, off));
        }
        finally { foreach (var h in handles) h.Free(); }
        var result = new List<List<KeypointPair>>(pairs.Count);
        for (var m = 0; m < pairs.Count; m++)
        {
            var (a, b) = pairs[m];
            var list = new List<KeypointPair>(frames[a].Count);
            for (var i = offs[m]; i < offs[m + 1]; i++)
                list.Add(new KeypointPair { Keypoint1 = frames[a][o[i].K1], Keypoint2 = frames[b][o[i].K2], Distance = o[i].Dist });
            result.Add(list);
        }
        return result;
    }

    private static uint[] Pack(List<Keypoint> kps, int words)
    {
        var flat = new uint[Math.Max(1, kps.Count * words)];
        for (var i = 0; i < kps.Count; i++)
        {
            var bytes = kps[i].BriefDescriptor.ToByteArray(isUnsigned: true, isBigEndian: false);
            Buffer.BlockCopy(bytes, 0, flat, i * words * 4, Math.Min(bytes.Length, words * 4));
        }
        return flat;
    }
}
