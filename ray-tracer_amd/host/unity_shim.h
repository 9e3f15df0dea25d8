// unity_shim.h — Unity-shaped wrappers over include/rt_native.h shared by the two frame drivers.
//
// ComputeShader / ComputeBuffer / RenderTexture carry Unity's method names, ComputeHelper / Utils the reference's
// helper names (ComputeHelper.cs, Resources.cs), so the drivers read like the C# they restate: this is the same
// substitution a C# host would make with a P/Invoke shim.
#pragma once

#include <cmath>
#include <cstdint>
#include <string>

#include "../../include/rt_native.h"
#include "unity_math.h"

namespace shim {

struct int2 { int x, y; };

struct DriverError {
    std::string msg;
};

// ---- Unity-shaped wrappers over the C-ABI ---------------------------------------------------------------------
struct ComputeBuffer {
    rt_buffer h = nullptr;
    void Release() { rt_buffer_release(h); h = nullptr; }
};
struct RenderTexture {
    rt_texture h = nullptr;
    void Release() { rt_texture_release(h); h = nullptr; }
};
struct ComputeShader {
    rt_shader h = nullptr;
    rt_ctx ctx = nullptr;
    void check(int rc, const char* what) const {
        if (rc < 0) throw DriverError{std::string(what) + ": " + rt_last_error(ctx)};
    }
    void SetInt(const char* n, int v) { check(rt_set_int(h, n, v), n); }
    void SetFloat(const char* n, float v) { check(rt_set_float(h, n, v), n); }
    void SetBool(const char* n, bool v) { check(rt_set_bool(h, n, v ? 1 : 0), n); }
    void SetInts(const char* n, const int* v, int c) { check(rt_set_ints(h, n, v, c), n); }
    void SetVector(const char* n, float x, float y, float z = 0, float w = 0) {
        float v[4] = {x, y, z, w};
        check(rt_set_vector(h, n, v), n);
    }
    void SetMatrix(const char* n, const M4& m) { check(rt_set_matrix(h, n, m.m), n); }
    void SetBuffer(int k, const char* n, const ComputeBuffer& b) { check(rt_set_buffer(h, k, n, b.h), n); }
    void SetTexture(int k, const char* n, const RenderTexture& t) { check(rt_set_texture(h, k, n, t.h), n); }
    void EnableKeyword(const char* k) { check(rt_enable_keyword(h, k), k); }
    void DisableKeyword(const char* k) { check(rt_disable_keyword(h, k), k); }
    int FindKernel(const char* n) { int k = rt_find_kernel(h, n); check(k, n); return k; }
    void Dispatch(int k, int gx, int gy, int gz) { check(rt_dispatch(h, k, gx, gy, gz), "Dispatch"); }
};

// ---- ComputeHelper.cs / Resources.cs ----------------------------------------------------------------------------
namespace Utils {
inline int GetThreadGroupsNum(int threadsNum, int threadSize) { return (int)std::ceil((float)threadsNum / threadSize); }  // Resources.cs:16-20
}
namespace ComputeHelper {
inline void DispatchKernel(ComputeShader& cs, const char* kernelName, int threadsNum, int threadSize) {  // ComputeHelper.cs:13-17
    cs.Dispatch(cs.FindKernel(kernelName), Utils::GetThreadGroupsNum(threadsNum, threadSize), 1, 1);
}
inline void DispatchKernel(ComputeShader& cs, const char* kernelName, int2 threadsNum, int threadSize) {  // :20-24
    cs.Dispatch(cs.FindKernel(kernelName), Utils::GetThreadGroupsNum(threadsNum.x, threadSize),
                Utils::GetThreadGroupsNum(threadsNum.y, threadSize), 1);
}
inline ComputeBuffer CreateStructuredBuffer(rt_ctx ctx, const void* data, int64_t length, int stride) {  // :73-78
    ComputeBuffer b;
    int rc = rt_buffer_create(ctx, length > 1 ? length : 1, stride, &b.h);
    if (rc < 0) throw DriverError{std::string("CreateStructuredBuffer: ") + rt_last_error(ctx)};
    if (length > 0) {
        rc = rt_buffer_set_data(b.h, data, (size_t)length * stride);
        if (rc < 0) throw DriverError{std::string("SetData: ") + rt_last_error(ctx)};
    }
    return b;
}
inline ComputeBuffer CreateStructuredBuffer(rt_ctx ctx, int64_t count, int stride) {  // :96-100
    ComputeBuffer b;
    int rc = rt_buffer_create(ctx, count, stride, &b.h);
    if (rc < 0) throw DriverError{std::string("CreateStructuredBuffer: ") + rt_last_error(ctx)};
    return b;
}
}  // namespace ComputeHelper

inline RenderTexture CreateTexture(rt_ctx ctx, int2 res, rt_format fmt) {  // TextureHelper.cs:118-156, Main.cs:359
    RenderTexture t;
    int rc = rt_texture_create(ctx, res.x, res.y, fmt, &t.h);
    if (rc < 0) throw DriverError{std::string("CreateTexture: ") + rt_last_error(ctx)};
    return t;
}

}  // namespace shim
