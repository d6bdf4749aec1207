// UrtUnityShim.cs — stand-ins with the SAME member names RayTraceMaster.cs ("RM") uses on UnityEngine.ComputeBuffer,
// ComputeShader, RenderTexture and Graphics.Blit, implemented over UrtNative (libunityraytracer_amd.so).  With this file in
// Assets/Scripts/, RM changes only type names at its declarations (RM:8, 11-12, 26-37: ComputeShader -> UrtComputeShader,
// ComputeBuffer -> UrtComputeBuffer, RenderTexture -> UrtRenderTexture) and the three Graphics.Blit calls of RM:818-819
// (-> UrtGraphics.Blit); every call inside its methods compiles unchanged against these classes.  Set UrtDevice.Devices to
// more than one ordinal and the same RM drives a device group: the frame is cut into 8-row strips over the GPUs and
// UrtGraphics.Blit(_converged, destination) performs the one frame-end gather.  SOURCE ONLY (no C# toolchain in the build
// image of this repository); the Python twin of this file, which IS exercised by tests, is unityraytracer_amd/unity_api.py.
using System;
using System.Collections.Generic;
using System.Runtime.InteropServices;
using UnityEngine;

/// The process-wide device selection: one context, or a group when several ordinals are listed.
public static class UrtDevice {
    public static int[] Devices = { 0 };
    static IntPtr ctx = IntPtr.Zero, group = IntPtr.Zero;
    internal static bool IsGroup { get { Ensure(); return group != IntPtr.Zero; } }
    internal static IntPtr Handle { get { Ensure(); return group != IntPtr.Zero ? group : ctx; } }
    static void Ensure() {
        if (ctx != IntPtr.Zero || group != IntPtr.Zero) return;
        // a negative ABI version marks an A/B / probe / diagnostic BUILD of the library (csrc/experiments.h): never the product
        if (UrtNative.urt_abi_version() < 0) throw new InvalidOperationException("libunityraytracer_amd is an experiment build (negative urt_abi_version): refused");
        if (Devices.Length > 1) UrtNative.CheckGroup(IntPtr.Zero, UrtNative.urt_group_create(Devices, Devices.Length, out group));
        else UrtNative.Check(IntPtr.Zero, UrtNative.urt_context_create(Devices[0], out ctx));
    }
    internal static void Check(int rc) { if (IsGroup) UrtNative.CheckGroup(group, rc); else UrtNative.Check(ctx, rc); }
    public static void Shutdown() {
        if (group != IntPtr.Zero) UrtNative.urt_group_destroy(group);
        if (ctx != IntPtr.Zero) UrtNative.urt_context_destroy(ctx);
        group = ctx = IntPtr.Zero;
    }
}

/// new ComputeBuffer(count, stride); .SetData(List<T>); .Release(); .count; .stride            (RM:233-252)
public sealed class UrtComputeBuffer {
    internal ulong handle;
    public int count { get; private set; }
    public int stride { get; private set; }
    public UrtComputeBuffer(int count, int stride) {
        this.count = count; this.stride = stride;
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_buffer_create(UrtDevice.Handle, count, stride, out handle)
                                          : UrtNative.urt_buffer_create(UrtDevice.Handle, count, stride, out handle));
    }
    public void SetData<T>(List<T> data) where T : struct {
        T[] a = data.ToArray();                                   // the library copies before returning (SetData semantics)
        GCHandle pin = GCHandle.Alloc(a, GCHandleType.Pinned);
        try {
            IntPtr p = pin.AddrOfPinnedObject();
            UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_buffer_set_data(UrtDevice.Handle, handle, p, a.Length)
                                              : UrtNative.urt_buffer_set_data(UrtDevice.Handle, handle, p, a.Length));
        } finally { pin.Free(); }
    }
    public void Release() {
        if (handle == 0) return;
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_buffer_release(UrtDevice.Handle, handle) : UrtNative.urt_buffer_release(UrtDevice.Handle, handle));
        handle = 0;
    }
}

/// new RenderTexture(w, h, 0, ARGBFloat, Linear) { enableRandomWrite = true }.Create(); .Release(); .width; .height   (RM:824-845)
public sealed class UrtRenderTexture {
    internal ulong handle;
    public int width { get; private set; }
    public int height { get; private set; }
    public bool enableRandomWrite;                                  // accepted: every image here is writable
    public UrtRenderTexture(int width, int height, int depth, RenderTextureFormat format, RenderTextureReadWrite readWrite) {
        this.width = width; this.height = height;                   // format is ARGBFloat / Linear at the only call site (RM:834-840)
    }
    public bool Create() {
        if (handle != 0) return true;
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_texture_create(UrtDevice.Handle, width, height, out handle)
                                          : UrtNative.urt_texture_create(UrtDevice.Handle, width, height, out handle));
        return handle != 0;
    }
    public void Release() {
        if (handle == 0) return;
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_texture_release(UrtDevice.Handle, handle) : UrtNative.urt_texture_release(UrtDevice.Handle, handle));
        handle = 0;
    }
    /// Upload RGBA32F texels, row 0 = bottom (Unity's own row order): the sky (RM:776) after Texture2D.GetPixelData<float>.
    public void SetPixels(float[] rgba) {
        Create();
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_texture_set_pixels(UrtDevice.Handle, handle, rgba) : UrtNative.urt_texture_set_pixels(UrtDevice.Handle, handle, rgba));
    }
    public void GetPixels(float[] rgba) {
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_texture_get_pixels(UrtDevice.Handle, handle, rgba) : UrtNative.urt_texture_get_pixels(UrtDevice.Handle, handle, rgba));
    }
}

/// RayTraceShader.SetMatrix/SetVector/SetFloat/SetInt/SetTexture/SetBuffer/Dispatch                (RM:255-259, 772-810)
public sealed class UrtComputeShader {
    public int FindKernel(string name) { return 0; }                // CSMain
    public void SetMatrix(string name, Matrix4x4 m) { UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_shader_set_matrix(UrtDevice.Handle, name, ref m) : UrtNative.urt_shader_set_matrix(UrtDevice.Handle, name, ref m)); }
    public void SetVector(string name, Vector4 v) { UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_shader_set_vector(UrtDevice.Handle, name, ref v) : UrtNative.urt_shader_set_vector(UrtDevice.Handle, name, ref v)); }
    public void SetFloat(string name, float v) { UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_shader_set_float(UrtDevice.Handle, name, v) : UrtNative.urt_shader_set_float(UrtDevice.Handle, name, v)); }
    public void SetInt(string name, int v) { UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_shader_set_int(UrtDevice.Handle, name, v) : UrtNative.urt_shader_set_int(UrtDevice.Handle, name, v)); }
    public void SetTexture(int kernel, string name, UrtRenderTexture t) {
        ulong h = t == null ? 0 : t.handle;
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_shader_set_texture(UrtDevice.Handle, kernel, name, h) : UrtNative.urt_shader_set_texture(UrtDevice.Handle, kernel, name, h));
    }
    public void SetBuffer(int kernel, string name, UrtComputeBuffer b) {
        ulong h = b == null ? 0 : b.handle;
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_shader_set_buffer(UrtDevice.Handle, kernel, name, h) : UrtNative.urt_shader_set_buffer(UrtDevice.Handle, kernel, name, h));
    }
    /// On a group the dispatch is partitioned: rank r traces the 8-row strips r, r+N, ... (same pixels as one full dispatch).
    public void Dispatch(int kernel, int gx, int gy, int gz) {
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_shader_dispatch(UrtDevice.Handle, kernel, gx, gy, gz) : UrtNative.urt_shader_dispatch(UrtDevice.Handle, kernel, gx, gy, gz));
    }
}

/// _additionMaterial.SetFloat("_Sample", n) + Graphics.Blit(src, dst, _additionMaterial); Graphics.Blit(src, dst)   (RM:813-819)
public sealed class UrtAdditionMaterial {
    float sample;
    public void SetFloat(string name, float v) { if (name == "_Sample") sample = v; }
    internal float Sample { get { return sample; } }
}

public static class UrtGraphics {
    public static void Blit(UrtRenderTexture src, UrtRenderTexture dst, UrtAdditionMaterial mat) {       // RM:818
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_blit_add(UrtDevice.Handle, src.handle, dst.handle, mat.Sample)
                                          : UrtNative.urt_blit_add(UrtDevice.Handle, src.handle, dst.handle, mat.Sample));
    }
    /// RM:819 "present": on one device a copy — queued behind the deferred frames and fused into the blend pass, so presenting every
    /// frame keeps the 64-frame launches (include/urt.h "Frame batching"); on a group THE frame-end gather — every rank's strips of
    /// `src` -> the full image `dst` on rank 0.
    public static void Blit(UrtRenderTexture src, UrtRenderTexture dst) {
        UrtDevice.Check(UrtDevice.IsGroup ? UrtNative.urt_group_gather(UrtDevice.Handle, src.handle, dst.handle)
                                          : UrtNative.urt_blit(UrtDevice.Handle, src.handle, dst.handle));
    }
    /// Present through Unity WITHOUT stalling the tracer: the frame that is presented is the previous call's (one frame of latency).
    /// urt_texture_read_begin snapshots `src` and sends it to a pinned host image on a copy stream while the next frames render;
    /// urt_texture_read_end hands that image out, and Texture2D.LoadRawTextureData takes it without a managed copy.
    static ulong pendingTicket = 0;
    public static void BlitPipelined(UrtRenderTexture src, RenderTexture unityDestination, ref Texture2D staging) {
        if (UrtDevice.IsGroup) throw new NotSupportedException("pipelined readback is per device: gather to rank 0's image first");
        // The image crosses the bus in the destination's OWN format, converted on the GPU (csrc/present.hip): an 8-bit back buffer of this
        // linear-colour-space project gets sRGB-encoded bytes (8.3 MB per 1080p frame instead of 33.2 MB), an HDR camera's ARGBHalf halfs.
        bool hdr = unityDestination != null && unityDestination.format == RenderTextureFormat.ARGBHalf;
        TextureFormat tf = hdr ? TextureFormat.RGBAHalf : TextureFormat.RGBA32;
        if (staging == null || staging.width != src.width || staging.height != src.height || staging.format != tf)
            staging = new Texture2D(src.width, src.height, tf, false, /* linear: */ hdr);   // RGBA32 declared sRGB: sampling decodes, the blit re-encodes
        if (pendingTicket != 0) {
            IntPtr pixels; UIntPtr bytes;
            UrtDevice.Check(UrtNative.urt_texture_read_end_format(UrtDevice.Handle, pendingTicket, out pixels, out bytes));
            staging.LoadRawTextureData(pixels, (int)bytes.ToUInt32());
            staging.Apply(false);
            Graphics.Blit(staging, unityDestination);
        }
        UrtDevice.Check(UrtNative.urt_texture_read_begin_format(UrtDevice.Handle, src.handle,
                                                                hdr ? UrtNative.URT_FORMAT_RGBA16F : UrtNative.URT_FORMAT_RGBA8_SRGB, out pendingTicket));
    }
    /// Present through Unity: read the image back (this submits and waits) and hand it to a Unity RenderTexture.
    public static void Blit(UrtRenderTexture src, RenderTexture unityDestination, ref Texture2D staging, ref float[] managed) {
        int n = src.width * src.height * 4;
        if (managed == null || managed.Length != n) managed = new float[n];
        src.GetPixels(managed);
        if (staging == null || staging.width != src.width || staging.height != src.height)
            staging = new Texture2D(src.width, src.height, TextureFormat.RGBAFloat, false, true);
        staging.SetPixelData(managed, 0);
        staging.Apply(false);
        Graphics.Blit(staging, unityDestination);
    }
}
