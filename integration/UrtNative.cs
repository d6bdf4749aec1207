// UrtNative.cs — P/Invoke declarations for libunityraytracer_amd.so (include/urt.h, ABI version 4; a negative urt_abi_version() = an experiment build: refuse it).
// Drop into Assets/Scripts/ of RemyMuj/UnityRayTracer next to RayTraceMaster.cs ("RM"); the native library goes to
// Assets/Plugins/x86_64/libunityraytracer_amd.so.  One declaration per exported entry point, in the header's order; each
// group cites the RM call site it stands in for.  SOURCE ONLY: the build image of this repository has no C#/.NET/Mono
// toolchain, so this file has not been compiled there; the same ABI is exercised by unityraytracer_amd/_lib.py (ctypes).
using System;
using System.Runtime.InteropServices;

internal static class UrtNative {
    const string Lib = "unityraytracer_amd";

    // ---- library / context ------------------------------------------------------------------------------------------
    [DllImport(Lib)] internal static extern int urt_abi_version();
    [DllImport(Lib)] internal static extern int urt_device_count(out int count);
    [DllImport(Lib)] internal static extern int urt_context_create(int device, out IntPtr ctx);
    [DllImport(Lib)] internal static extern int urt_context_destroy(IntPtr ctx);
    [DllImport(Lib)] internal static extern IntPtr urt_last_error(IntPtr ctx);                      // const char*, never NULL
    [DllImport(Lib)] internal static extern int urt_context_set_stream(IntPtr ctx, IntPtr hipStream);
    [DllImport(Lib)] internal static extern int urt_synchronize(IntPtr ctx);
    [DllImport(Lib)] internal static extern int urt_flush(IntPtr ctx);                              // submit batched frames, no wait

    // ---- ComputeBuffer                                                                     RM:233-252, 195-210 --------
    [DllImport(Lib)] internal static extern int urt_buffer_create(IntPtr ctx, int count, int stride, out ulong buffer);
    [DllImport(Lib)] internal static extern int urt_buffer_set_data(IntPtr ctx, ulong buffer, IntPtr data, int count);
    [DllImport(Lib)] internal static extern int urt_buffer_get_info(IntPtr ctx, ulong buffer, out int count, out int stride);
    [DllImport(Lib)] internal static extern int urt_buffer_release(IntPtr ctx, ulong buffer);

    // ---- RenderTexture / Texture (ARGBFloat, linear)                                        RM:824-845 -----------------
    [DllImport(Lib)] internal static extern int urt_texture_create(IntPtr ctx, int width, int height, out ulong texture);
    [DllImport(Lib)] internal static extern int urt_texture_create_external(IntPtr ctx, int width, int height, IntPtr devicePtr, out ulong texture);
    [DllImport(Lib)] internal static extern int urt_texture_set_pixels(IntPtr ctx, ulong texture, float[] rgba);
    [DllImport(Lib)] internal static extern int urt_texture_get_pixels(IntPtr ctx, ulong texture, [Out] float[] rgba);
    [DllImport(Lib)] internal static extern int urt_texture_get_info(IntPtr ctx, ulong texture, out int width, out int height, out IntPtr devicePtr);
    [DllImport(Lib)] internal static extern int urt_texture_release(IntPtr ctx, ulong texture);

    // ---- ComputeShader (kernel 0 = CSMain)                                                  RM:772-810 -----------------
    [DllImport(Lib)] internal static extern int urt_shader_set_buffer(IntPtr ctx, int kernel, string name, ulong buffer);
    [DllImport(Lib)] internal static extern int urt_shader_set_texture(IntPtr ctx, int kernel, string name, ulong texture);
    [DllImport(Lib)] internal static extern int urt_shader_set_matrix(IntPtr ctx, string name, ref UnityEngine.Matrix4x4 m);
    [DllImport(Lib)] internal static extern int urt_shader_set_vector(IntPtr ctx, string name, ref UnityEngine.Vector4 v);
    [DllImport(Lib)] internal static extern int urt_shader_set_float(IntPtr ctx, string name, float v);
    [DllImport(Lib)] internal static extern int urt_shader_set_int(IntPtr ctx, string name, int v);
    [DllImport(Lib)] internal static extern int urt_shader_dispatch(IntPtr ctx, int kernel, int gx, int gy, int gz);
    // multi-GPU form of Dispatch: only the 8-row strips firstGroupRow, +rowStride, ... (global pixel ids kept)
    [DllImport(Lib)] internal static extern int urt_shader_dispatch_rows(IntPtr ctx, int kernel, int gx, int gy, int gz, int firstGroupRow, int rowStride);

    // ---- Graphics.Blit                                                                      RM:817-819 -----------------
    [DllImport(Lib)] internal static extern int urt_blit_add(IntPtr ctx, ulong src, ulong dst, float sample);
    [DllImport(Lib)] internal static extern int urt_blit(IntPtr ctx, ulong src, ulong dst);
    // strips <-> dense device buffer for the frame-end gather of a one-process-per-GPU host
    [DllImport(Lib)] internal static extern int urt_texture_pack_rows(IntPtr ctx, ulong texture, int firstGroupRow, int rowStride, IntPtr deviceDst, out ulong bytes);
    [DllImport(Lib)] internal static extern int urt_texture_unpack_rows(IntPtr ctx, ulong texture, int firstGroupRow, int rowStride, IntPtr deviceSrc);
    [DllImport(Lib)] internal static extern int urt_texture_unpack_rows_on(IntPtr ctx, ulong texture, int firstGroupRow, int rowStride, IntPtr deviceSrc, IntPtr hipStream);
    [DllImport(Lib)] internal static extern int urt_texture_read_begin(IntPtr ctx, ulong texture, out ulong ticket);
    [DllImport(Lib)] internal static extern int urt_texture_read_end(IntPtr ctx, ulong ticket, out IntPtr rgba);   // pinned host image, width x height x 4 floats
    internal const int URT_FORMAT_RGBA32F = 0, URT_FORMAT_RGBA8_SRGB = 1, URT_FORMAT_RGBA16F = 2;
    [DllImport(Lib)] internal static extern int urt_texture_read_begin_format(IntPtr ctx, ulong texture, int format, out ulong ticket);   // converted on the GPU to the destination's format
    [DllImport(Lib)] internal static extern int urt_texture_read_end_format(IntPtr ctx, ulong ticket, out IntPtr pixels, out UIntPtr bytes);
    [DllImport(Lib)] internal static extern int urt_texture_pack_rows_rgb(IntPtr ctx, ulong texture, int firstGroupRow, int rowStride, IntPtr deviceDst, out ulong bytes);
    [DllImport(Lib)] internal static extern int urt_texture_unpack_rows_rgb(IntPtr ctx, ulong texture, int firstGroupRow, int rowStride, IntPtr deviceSrc, float alpha, IntPtr hipStream);

    // ---- measurement ----------------------------------------------------------------------------------------------------
    [StructLayout(LayoutKind.Sequential)]
    internal struct Counters {
        public ulong rays, tlasNodes, blasNodes, triTests, sphereTests, hitTri, hitSphere, hitGround, hitSky, pixels, dispatches;
        public float traceMs;
        public uint watchdogTrips;
        public ulong launches;
    }
    [DllImport(Lib)] internal static extern int urt_set_option(IntPtr ctx, string name, int value);
    [DllImport(Lib)] internal static extern int urt_get_counters(IntPtr ctx, out Counters c);
    [DllImport(Lib)] internal static extern int urt_reset_counters(IntPtr ctx);

    // ---- device groups: this (single) host thread drives N GPUs                               include/urt.h "device groups"
    [DllImport(Lib)] internal static extern int urt_group_create(int[] devices, int nDevices, out IntPtr group);
    [DllImport(Lib)] internal static extern int urt_group_destroy(IntPtr group);
    [DllImport(Lib)] internal static extern int urt_group_size(IntPtr group);
    [DllImport(Lib)] internal static extern IntPtr urt_group_context(IntPtr group, int rank);
    [DllImport(Lib)] internal static extern IntPtr urt_group_last_error(IntPtr group);
    [DllImport(Lib)] internal static extern int urt_group_buffer_create(IntPtr group, int count, int stride, out ulong buffer);
    [DllImport(Lib)] internal static extern int urt_group_buffer_set_data(IntPtr group, ulong buffer, IntPtr data, int count);
    [DllImport(Lib)] internal static extern int urt_group_buffer_release(IntPtr group, ulong buffer);
    [DllImport(Lib)] internal static extern int urt_group_texture_create(IntPtr group, int width, int height, out ulong texture);
    [DllImport(Lib)] internal static extern int urt_group_texture_set_pixels(IntPtr group, ulong texture, float[] rgba);
    [DllImport(Lib)] internal static extern int urt_group_texture_get_pixels(IntPtr group, ulong texture, [Out] float[] rgba);
    [DllImport(Lib)] internal static extern int urt_group_texture_release(IntPtr group, ulong texture);
    [DllImport(Lib)] internal static extern int urt_group_shader_set_buffer(IntPtr group, int kernel, string name, ulong buffer);
    [DllImport(Lib)] internal static extern int urt_group_shader_set_texture(IntPtr group, int kernel, string name, ulong texture);
    [DllImport(Lib)] internal static extern int urt_group_shader_set_matrix(IntPtr group, string name, ref UnityEngine.Matrix4x4 m);
    [DllImport(Lib)] internal static extern int urt_group_shader_set_vector(IntPtr group, string name, ref UnityEngine.Vector4 v);
    [DllImport(Lib)] internal static extern int urt_group_shader_set_float(IntPtr group, string name, float v);
    [DllImport(Lib)] internal static extern int urt_group_shader_set_int(IntPtr group, string name, int v);
    [DllImport(Lib)] internal static extern int urt_group_set_option(IntPtr group, string name, int value);
    [DllImport(Lib)] internal static extern int urt_group_shader_dispatch(IntPtr group, int kernel, int gx, int gy, int gz);
    [DllImport(Lib)] internal static extern int urt_group_blit_add(IntPtr group, ulong src, ulong dst, float sample);
    [DllImport(Lib)] internal static extern int urt_group_blit(IntPtr group, ulong src, ulong dst);
    [DllImport(Lib)] internal static extern int urt_group_gather(IntPtr group, ulong srcTexture, ulong dstTexture);
    [DllImport(Lib)] internal static extern int urt_group_flush(IntPtr group);
    [DllImport(Lib)] internal static extern int urt_group_synchronize(IntPtr group);
    [DllImport(Lib)] internal static extern int urt_group_get_counters(IntPtr group, out Counters c);
    [DllImport(Lib)] internal static extern int urt_group_reset_counters(IntPtr group);

    // ---- host-side helpers (no GPU): normals, object-level heaps, .hdr loader, frame writers, debug log -------------------
    [DllImport(Lib)] internal static extern int urt_host_compute_normals(IntPtr vertices, int nVertices, IntPtr indices, int nIndices, IntPtr outNormals);
    [DllImport(Lib)] internal static extern int urt_host_mesh_leaf_bounds(IntPtr meshObjects, int nMeshes, IntPtr vertices, int nVertices, IntPtr indices, int nIndices, int literal, IntPtr outLeaves);
    [DllImport(Lib)] internal static extern int urt_host_sphere_leaf_bounds(IntPtr spheres, int nSpheres, int literal, IntPtr outLeaves);
    [DllImport(Lib)] internal static extern int urt_host_object_bvh_length(int nObjects);
    [DllImport(Lib)] internal static extern int urt_host_build_object_bvh(IntPtr leaves, int nObjects, IntPtr outNodes, int capacity);
    [DllImport(Lib)] internal static extern int urt_host_load_hdr(string path, out int width, out int height, [Out] float[] rgba, UIntPtr capacityFloats);
    [DllImport(Lib)] internal static extern int urt_host_write_pfm(string path, float[] rgba, int width, int height);
    [DllImport(Lib)] internal static extern int urt_host_write_png(string path, float[] rgba, int width, int height);
    [DllImport(Lib)] internal static extern int urt_host_encode_srgb8(float[] rgba, UIntPtr nPixels, byte[] outRgba8);
    [DllImport(Lib)] internal static extern int urt_host_srgb8_first_floats(float[] out256);
    [DllImport(Lib)] internal static extern int urt_host_log(string path, int debugLevel, int level, string text);
    [DllImport(Lib)] internal static extern int urt_host_log_scene_counts(string path, int debugLevel, int nSpheres, int nMeshObjects, int nVertices, int nIndices, int nNormals);
    [DllImport(Lib)] internal static extern int urt_host_log_tree_report(string path, int debugLevel, int nMeshObjects, int meshDepth, int meshRealLength, int nSpheres, int sphereDepth, int sphereRealLength);
    [DllImport(Lib)] internal static extern int urt_host_dump_bvh(string path, IntPtr nodes, int nNodes, int depth, float[] rayStart3, float[] rayEnd3, out int lines);
    [DllImport(Lib)] internal static extern int urt_host_dump_normals(string path, IntPtr meshObjects, int nMeshes, float[] vertices, int nVertices, int[] indices, int nIndices, float[] normals, int nNormals, out int lines);
    [DllImport(Lib)] internal static extern int urt_debug_refit_stats(IntPtr ctx, out ulong refittedMeshes, out ulong incrementalPreparations);
    [DllImport(Lib)] internal static extern int urt_host_build_object_bvh_pairing(IntPtr leaves, int nObjects, IntPtr outNodes, int capacity);   // RM:459-722's pairing builder, restated
    [DllImport(Lib)] internal static extern int urt_host_resize_rgba(float[] src, int width, int height, [Out] float[] dst, int newWidth, int newHeight);
    [DllImport(Lib)] internal static extern IntPtr urt_host_last_error();          // const char* (thread-local, like the two below)
    [DllImport(Lib)] internal static extern IntPtr urt_host_io_last_error();
    [DllImport(Lib)] internal static extern IntPtr urt_host_debug_last_error();
    // ---- introspection (what tests/ use; not needed by RM) --------------------------------------------------------------
    [DllImport(Lib)] internal static extern int urt_debug_build_blas(IntPtr meshObjects, int nMeshes, float[] vertices, int nVertices, int[] indices, int nIndices, out int nNodes, out int nTris, out int maxDepth);
    [DllImport(Lib)] internal static extern int urt_debug_get_blas([Out] float[] nodes, [Out] int[] triIndex, [Out] int[] meshRoot, [Out] int[] meshFirstTri);
    [DllImport(Lib)] internal static extern int urt_debug_scene_info(IntPtr ctx, out int nNodes, out int nTris, out int maxDepth, out float prepareMs);
    [DllImport(Lib)] internal static extern int urt_debug_launch_info(IntPtr ctx, [Out] byte[] launchInfo192);   // urt_launch_info: char kernel[96] + 24 ints
    [DllImport(Lib)] internal static extern int urt_debug_read_scene_blas(IntPtr ctx, [Out] float[] nodes, [Out] int[] triIndex, [Out] int[] meshRoot);
    [DllImport(Lib)] internal static extern int urt_debug_blas_cache_stats(IntPtr ctx, out ulong reused, out ulong built);
    [DllImport(Lib)] internal static extern int urt_debug_serve_stats(IntPtr ctx, [Out] ulong[] out6);
    [DllImport(Lib)] internal static extern int urt_debug_build_walk_table(IntPtr heap, int nNodes, int nMeshes, int[] meshRoot, int[] smallFirst, float[] table, int capacityWords, out int words);

    // Unity's own API returns void and logs on error: the shim keeps that behaviour.
    internal static void Check(IntPtr ctx, int rc) {
        if (rc != 0) UnityEngine.Debug.LogError("urt: " + Marshal.PtrToStringAnsi(urt_last_error(ctx)));
    }
    internal static void CheckGroup(IntPtr group, int rc) {
        if (rc != 0) UnityEngine.Debug.LogError("urt group: " + Marshal.PtrToStringAnsi(urt_group_last_error(group)));
    }
}
