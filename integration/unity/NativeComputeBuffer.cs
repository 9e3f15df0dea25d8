// NativeComputeBuffer.cs - the part of UnityEngine.ComputeBuffer the reference uses (ComputeHelper.cs:73-110,203-214,
// ShaderHelper.cs:7-73), over rt_buffer_*.
using System;
using System.Runtime.InteropServices;

public sealed class NativeComputeBuffer
{
    internal IntPtr h;
    public readonly int count, stride;

    public NativeComputeBuffer(int count, int stride)                     // new ComputeBuffer(count, stride)
    {
        this.count = count;
        this.stride = stride;
        RtNative.Check(RtNative.rt_buffer_create(NativeDevice.Ctx, count, stride, out h));
    }

    public bool IsValid() => h != IntPtr.Zero;

    public void SetData<T>(T[] data) where T : struct                     // ComputeBuffer.SetData(Array)
    {
        if (Marshal.SizeOf<T>() != stride) throw new ArgumentException($"element size {Marshal.SizeOf<T>()} != buffer stride {stride}");
        var pin = GCHandle.Alloc(data, GCHandleType.Pinned);
        try { RtNative.Check(RtNative.rt_buffer_set_data(h, pin.AddrOfPinnedObject(), (UIntPtr)((ulong)data.Length * (ulong)stride))); }
        finally { pin.Free(); }
    }

    public void GetData<T>(T[] data) where T : struct                     // ComputeBuffer.GetData(Array)
    {
        if (Marshal.SizeOf<T>() != stride) throw new ArgumentException($"element size {Marshal.SizeOf<T>()} != buffer stride {stride}");
        var pin = GCHandle.Alloc(data, GCHandleType.Pinned);
        try { RtNative.Check(RtNative.rt_buffer_get_data(h, pin.AddrOfPinnedObject(), (UIntPtr)((ulong)data.Length * (ulong)stride))); }
        finally { pin.Free(); }
    }

    // ComputeBuffer.GetData(Array data, int managedBufferStartIndex, int computeBufferStartIndex, int count)
    public void GetData<T>(T[] data, int managedBufferStartIndex, int computeBufferStartIndex, int count) where T : struct
    {
        if (Marshal.SizeOf<T>() != stride) throw new ArgumentException($"element size {Marshal.SizeOf<T>()} != buffer stride {stride}");
        if (managedBufferStartIndex < 0 || count < 0 || managedBufferStartIndex + count > data.Length) throw new ArgumentOutOfRangeException(nameof(count));
        var pin = GCHandle.Alloc(data, GCHandleType.Pinned);
        try
        {
            IntPtr dst = IntPtr.Add(pin.AddrOfPinnedObject(), managedBufferStartIndex * stride);
            RtNative.Check(RtNative.rt_buffer_get_range(h, dst, computeBufferStartIndex, count));
        }
        finally { pin.Free(); }
    }

    public void Release()                                                 // ComputeBuffer.Release
    {
        RtNative.rt_buffer_release(h);
        h = IntPtr.Zero;
    }
}
