// NativeComputeShader.cs - the part of UnityEngine.ComputeShader the reference uses (Main.cs:134-145,184-271,387-420,
// ShaderHelper.cs:7-73, ComputeHelper.cs:16-23, ObjectHelper.cs:508-511,637,647,686,696) over rt_shader_* / rt_set_* / rt_dispatch.
using System;
using UnityEngine;

public sealed class NativeComputeShader
{
    readonly IntPtr h;

    /// <param name="assetName">"ReSTIR_BVH_RayTracer", "PreCalc", "PostProcessor"; the RIS project: "RIS_BVH_RayTracer", "PreCalc", "PostProcesser"</param>
    public NativeComputeShader(string assetName) { RtNative.Check(RtNative.rt_shader_open(NativeDevice.Ctx, assetName, out h)); }

    public int FindKernel(string name)
    {
        int k = RtNative.rt_find_kernel(h, name);
        RtNative.Check(k);
        return k;
    }

    public void Dispatch(int kernel, int x, int y, int z) => RtNative.Check(RtNative.rt_dispatch(h, kernel, x, y, z));

    /// <summary>Dispatch that allocates no device memory; RtStatus.NotPrepared until one frame went through Dispatch.</summary>
    public void DispatchNoAlloc(int kernel, int x, int y, int z) => RtNative.Check(RtNative.rt_dispatch_no_alloc(h, kernel, x, y, z));

    public void SetInt(string name, int v)             => RtNative.Check(RtNative.rt_set_int(h, name, v));
    public void SetFloat(string name, float v)         => RtNative.Check(RtNative.rt_set_float(h, name, v));
    public void SetBool(string name, bool v)           => RtNative.Check(RtNative.rt_set_bool(h, name, v ? 1 : 0));
    public void SetInts(string name, params int[] v)   => RtNative.Check(RtNative.rt_set_ints(h, name, v, v.Length));
    public void SetVector(string name, Vector4 v)      => RtNative.Check(RtNative.rt_set_vector(h, name, new[] { v.x, v.y, v.z, v.w }));

    public void SetMatrix(string name, Matrix4x4 m)    // Matrix4x4's indexer is column-major, as the library expects
    {
        var a = new float[16];
        for (int i = 0; i < 16; i++) a[i] = m[i];
        RtNative.Check(RtNative.rt_set_matrix(h, name, a));
    }

    public void SetBuffer(int kernel, string name, NativeComputeBuffer b)   => RtNative.Check(RtNative.rt_set_buffer(h, kernel, name, b.h));
    public void SetTexture(int kernel, string name, NativeRenderTexture t)  => RtNative.Check(RtNative.rt_set_texture(h, kernel, name, t.h));
    public void EnableKeyword(string keyword)  => RtNative.Check(RtNative.rt_enable_keyword(h, keyword));
    public void DisableKeyword(string keyword) => RtNative.Check(RtNative.rt_disable_keyword(h, keyword));

    public RtDispatchStats GetStats(int kernel)
    {
        RtNative.Check(RtNative.rt_get_stats(h, kernel, out RtDispatchStats s));
        return s;
    }

    public void Close() => RtNative.rt_shader_close(h);
}
