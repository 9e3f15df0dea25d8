// NativeDevice.cs - the process-wide context that stands in for Unity's graphics device: one HIP device per process.
using System;

public static class NativeDevice
{
    static IntPtr ctx = IntPtr.Zero;

    public static bool HasContext => ctx != IntPtr.Zero;

    /// <summary>HIP device ordinal this process renders on; set before the first native object is created
    /// (a multi-GPU launcher sets it to the local rank).</summary>
    public static int DeviceId = 0;

    public static IntPtr Ctx
    {
        get
        {
            if (ctx == IntPtr.Zero)
            {
                int rc = RtNative.rt_create(DeviceId, out ctx);
                if (rc < 0) throw new InvalidOperationException($"rt_create({DeviceId}) failed: {(RtStatus)rc} (no usable gfx950 GPU?)");
            }
            return ctx;
        }
    }

    /// <summary>Waits for every dispatch and upload enqueued so far.</summary>
    public static void Sync() => RtNative.Check(RtNative.rt_sync(Ctx));

    /// <summary>Launches a dispatch the library still holds back for merging, without waiting: call at the end of
    /// OnRenderImage when nothing else (a readback, Sync) follows the last Dispatch of the frame.</summary>
    public static void Flush() => RtNative.Check(RtNative.rt_flush(Ctx));

    /// <summary>false: every Dispatch launches at once (no merged launches); results are identical either way.</summary>
    public static void SetFusion(bool enabled) => RtNative.Check(RtNative.rt_set_fusion(Ctx, enabled ? 1 : 0));

    /// <summary>Multi-GPU: restricts every per-pixel kernel to image rows [row0, row1); row1 &lt;= 0 restores full frames.</summary>
    public static void SetRowBand(int row0, int row1) => RtNative.Check(RtNative.rt_set_row_band(Ctx, row0, row1));

    /// <summary>Call from OnDestroy / OnApplicationQuit after every buffer, texture and shader has been released.</summary>
    public static void Shutdown()
    {
        if (ctx != IntPtr.Zero) RtNative.rt_destroy(ctx);
        ctx = IntPtr.Zero;
    }
}
