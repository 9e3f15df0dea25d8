// RtNative.cs - P/Invoke declarations for librtnative.so, one per declaration of include/rt_native.h, in the same order.
// Copy the five files of this folder to Assets/Scripts/Native/ and librtnative.so to Assets/Plugins/x86_64/.
// tests/test_unity_bindings.py checks every declaration here against the header (name, arity, argument types).
using System;
using System.Runtime.InteropServices;

public enum RtStatus { Ok = 0, InvalidArg = -1, NoDevice = -2, OutOfMemory = -3, Unbound = -4, BadScene = -5, Limit = -6, NotPrepared = -7, Launch = -8, Busy = -9 }
public enum RtFormat { R32F = 1, RGBA32F = 2, RGBA8Unorm = 3 }

[StructLayout(LayoutKind.Sequential)]
public struct RtSegment { public IntPtr ptr; public UIntPtr bytes; }

[StructLayout(LayoutKind.Sequential)]
public struct RtDispatchStats
{
    public ulong rays, pops, bvChecks, objVisits, triIters, triChecks, validNeighbours, activePixels;
}

[StructLayout(LayoutKind.Sequential)]
public struct RtKernelProfile { public ulong launches; public double milliseconds; public RtDispatchStats stats; }

internal static class RtNative
{
    const string Lib = "rtnative";

    // context
    [DllImport(Lib)] public static extern int rt_create(int deviceId, out IntPtr ctx);
    [DllImport(Lib)] public static extern void rt_destroy(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_sync(IntPtr ctx);
    [DllImport(Lib)] public static extern IntPtr rt_last_error(IntPtr ctx);
    [DllImport(Lib)] public static extern IntPtr rt_stream(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_flush(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_set_fusion(IntPtr ctx, int enabled);
    [DllImport(Lib)] public static extern int rt_set_overlap(IntPtr ctx, int mode);
    [DllImport(Lib)] public static extern int rt_set_tuning(IntPtr ctx, string name, int value);
    [DllImport(Lib)] public static extern int rt_set_row_band(IntPtr ctx, int row0, int row1);

    // shaders
    [DllImport(Lib)] public static extern int rt_shader_open(IntPtr ctx, string name, out IntPtr shader);
    [DllImport(Lib)] public static extern void rt_shader_close(IntPtr shader);
    [DllImport(Lib)] public static extern int rt_find_kernel(IntPtr shader, string name);
    [DllImport(Lib)] public static extern int rt_dispatch(IntPtr shader, int kernel, int groupsX, int groupsY, int groupsZ);
    [DllImport(Lib)] public static extern int rt_dispatch_no_alloc(IntPtr shader, int kernel, int groupsX, int groupsY, int groupsZ);
    [DllImport(Lib)] public static extern int rt_enable_keyword(IntPtr shader, string keyword);
    [DllImport(Lib)] public static extern int rt_disable_keyword(IntPtr shader, string keyword);
    [DllImport(Lib)] public static extern int rt_set_int(IntPtr shader, string name, int value);
    [DllImport(Lib)] public static extern int rt_set_float(IntPtr shader, string name, float value);
    [DllImport(Lib)] public static extern int rt_set_bool(IntPtr shader, string name, int value);
    [DllImport(Lib)] public static extern int rt_set_ints(IntPtr shader, string name, int[] values, int count);
    [DllImport(Lib)] public static extern int rt_set_vector(IntPtr shader, string name, float[] xyzw);
    [DllImport(Lib)] public static extern int rt_set_matrix(IntPtr shader, string name, float[] columnMajor16);
    [DllImport(Lib)] public static extern int rt_set_buffer(IntPtr shader, int kernel, string name, IntPtr buffer);
    [DllImport(Lib)] public static extern int rt_set_texture(IntPtr shader, int kernel, string name, IntPtr texture);

    // buffers
    [DllImport(Lib)] public static extern int rt_buffer_create(IntPtr ctx, long count, int stride, out IntPtr buffer);
    [DllImport(Lib)] public static extern int rt_buffer_set_data(IntPtr buffer, IntPtr src, UIntPtr bytes);
    [DllImport(Lib)] public static extern int rt_buffer_get_data(IntPtr buffer, IntPtr dst, UIntPtr bytes);
    [DllImport(Lib)] public static extern int rt_buffer_get_range(IntPtr buffer, IntPtr dst, long first, long count);
    [DllImport(Lib)] public static extern void rt_buffer_release(IntPtr buffer);
    [DllImport(Lib)] public static extern long rt_buffer_count(IntPtr buffer);
    [DllImport(Lib)] public static extern int rt_buffer_stride(IntPtr buffer);

    // textures
    [DllImport(Lib)] public static extern int rt_texture_create(IntPtr ctx, int width, int height, int format, out IntPtr texture);
    [DllImport(Lib)] public static extern int rt_texture_set_data(IntPtr texture, IntPtr src, UIntPtr bytes);
    [DllImport(Lib)] public static extern int rt_texture_get_data(IntPtr texture, IntPtr dst, UIntPtr bytes);
    [DllImport(Lib)] public static extern int rt_texture_get_rows(IntPtr texture, IntPtr dst, int row0, int row1);
    [DllImport(Lib)] public static extern void rt_texture_release(IntPtr texture);

    // multi-GPU row access and the transfer stream
    [DllImport(Lib)] public static extern int rt_buffer_map_rows(IntPtr buffer, int width, int row0, int row1, [Out] RtSegment[] segments4, out int count);
    [DllImport(Lib)] public static extern int rt_buffer_rows_written(IntPtr buffer, int width, int row0, int row1);
    [DllImport(Lib)] public static extern int rt_texture_map_rows(IntPtr texture, int row0, int row1, out RtSegment segment);
    [DllImport(Lib)] public static extern IntPtr rt_transfer_stream(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_stream_order(IntPtr ctx, int first);
    [DllImport(Lib)] public static extern int rt_mark(IntPtr ctx, int stream, int slot);
    [DllImport(Lib)] public static extern int rt_wait_mark(IntPtr ctx, int stream, int slot);
    [DllImport(Lib)] public static extern int rt_debug_trace(IntPtr ctx, int on);
    [DllImport(Lib)] public static extern int rt_debug_note_rows(IntPtr ctx, int stream, IntPtr buffer, IntPtr texture, int row0, int row1, int write);
    [DllImport(Lib)] public static extern long rt_debug_trace_dump(IntPtr ctx, IntPtr text, long capacity);
    [DllImport(Lib)] public static extern int rt_mark_elapsed(IntPtr ctx, int slotA, int slotB, out double milliseconds);

    // measurement
    [DllImport(Lib)] public static extern int rt_enable_stats(IntPtr ctx, int level);
    [DllImport(Lib)] public static extern int rt_get_stats(IntPtr shader, int kernel, out RtDispatchStats stats);
    [DllImport(Lib)] public static extern int rt_profile_enable(IntPtr ctx, int timing);
    [DllImport(Lib)] public static extern int rt_profile_select(IntPtr ctx, string kernelName);
    [DllImport(Lib)] public static extern int rt_profile_reset(IntPtr ctx);
    [DllImport(Lib)] public static extern int rt_profile_get(IntPtr ctx, string kernelName, out RtKernelProfile profile);

    /// <summary>Throws with the library's message when a call failed.</summary>
    public static void Check(int rc)
    {
        if (rc >= 0) return;
        IntPtr msg = NativeDevice.HasContext ? rt_last_error(NativeDevice.Ctx) : IntPtr.Zero;
        throw new InvalidOperationException($"librtnative: {(RtStatus)rc}: {(msg == IntPtr.Zero ? "(no message)" : Marshal.PtrToStringAnsi(msg))}");
    }
}
