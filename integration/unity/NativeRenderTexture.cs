// NativeRenderTexture.cs - the render targets and input textures of the reference (Main.cs:317-378, TextureHelper.cs:118-156,
// ObjectHelper.cs:381) over rt_texture_*.  Texels are read back as fp32 whatever the format.
using System;
using System.Runtime.InteropServices;
using UnityEngine;

public sealed class NativeRenderTexture
{
    internal IntPtr h;
    public readonly int width, height;
    public readonly RtFormat format;
    int Channels => format == RtFormat.R32F ? 1 : 4;

    public NativeRenderTexture(int width, int height, RtFormat format)
    {
        this.width = width;
        this.height = height;
        this.format = format;
        RtNative.Check(RtNative.rt_texture_create(NativeDevice.Ctx, width, height, (int)format, out h));
    }

    /// <summary>TextureHelper.CreateTexture(resolution, channels) for the 3- and 4-channel targets.  The reference makes them
    /// `new RenderTexture(w, h, 24)`, 8 bits per channel: pass eightBit = true for its exact numerics, false to keep fp32.</summary>
    public static NativeRenderTexture CreateTexture(Vector2Int resolution, int channels, bool eightBit = true)
    {
        if (channels == 1) return new NativeRenderTexture(resolution.x, resolution.y, RtFormat.R32F);
        return new NativeRenderTexture(resolution.x, resolution.y, eightBit ? RtFormat.RGBA8Unorm : RtFormat.RGBA32F);
    }

    /// <summary>An input texture (TextureAtlas, EnvironmentMap) from a readable Texture2D of format RGBA32.</summary>
    public static NativeRenderTexture FromTexture2D(Texture2D src)
    {
        var t = new NativeRenderTexture(src.width, src.height, RtFormat.RGBA8Unorm);
        byte[] raw = src.GetRawTextureData();
        if (raw.Length != src.width * src.height * 4) throw new ArgumentException("expected an uncompressed RGBA32 texture without mip maps");
        var pin = GCHandle.Alloc(raw, GCHandleType.Pinned);
        try { RtNative.Check(RtNative.rt_texture_set_data(t.h, pin.AddrOfPinnedObject(), (UIntPtr)(ulong)raw.Length)); }
        finally { pin.Free(); }
        return t;
    }

    /// <summary>All texels as floats (1 or 4 per texel), row 0 first.</summary>
    public float[] GetPixels()
    {
        var data = new float[width * height * Channels];
        var pin = GCHandle.Alloc(data, GCHandleType.Pinned);
        try { RtNative.Check(RtNative.rt_texture_get_data(h, pin.AddrOfPinnedObject(), (UIntPtr)((ulong)data.Length * 4))); }
        finally { pin.Free(); }
        return data;
    }

    /// <summary>Rows [row0, row1) only.</summary>
    public float[] GetRows(int row0, int row1)
    {
        var data = new float[width * (row1 - row0) * Channels];
        var pin = GCHandle.Alloc(data, GCHandleType.Pinned);
        try { RtNative.Check(RtNative.rt_texture_get_rows(h, pin.AddrOfPinnedObject(), row0, row1)); }
        finally { pin.Free(); }
        return data;
    }

    /// <summary>For display (Graphics.Blit(texture, dest) in Main.cs:431-464): copies the texels into a Texture2D of format
    /// RGBAFloat (RFloat for the depth buffer) of the same size and uploads it.</summary>
    public void CopyTo(Texture2D dest)
    {
        dest.SetPixelData(GetPixels(), 0);
        dest.Apply(false);
    }

    public void Release()
    {
        RtNative.rt_texture_release(h);
        h = IntPtr.Zero;
    }
}
