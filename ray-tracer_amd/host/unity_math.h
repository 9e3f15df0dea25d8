// unity_math.h — the handful of UnityEngine.Matrix4x4 / Quaternion operations the reference's host code relies on
// (BVH_ReSTIR_DL/Assets/Scripts/Resources.cs:67-87, Main.cs:184-204), evaluated in fp32 like Unity does.
#pragma once
#include <cmath>
#include <cstring>

namespace {

struct M4 { float m[16]; };  // column-major, element (r,c) = m[c*4+r]
inline float& at(M4& a, int r, int c) { return a.m[c * 4 + r]; }
inline float at(const M4& a, int r, int c) { return a.m[c * 4 + r]; }
inline M4 identity() { M4 r; memset(&r, 0, sizeof r); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1; return r; }
inline M4 mulM(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            at(r, i, j) = at(a, i, 0) * at(b, 0, j) + at(a, i, 1) * at(b, 1, j) + at(a, i, 2) * at(b, 2, j) + at(a, i, 3) * at(b, 3, j);
    return r;
}
// Matrix4x4.Rotate(Quaternion.Euler(...)) for a rotation about one axis by `deg` degrees.
inline M4 axisRotation(int axis, float deg) {
    double h = (double)deg * 0.017453292519943295 * 0.5;
    float q[4] = {0, 0, 0, (float)cos(h)};
    q[axis] = (float)sin(h);
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float x2 = x * 2, y2 = y * 2, z2 = z * 2;
    float xx = x * x2, yy = y * y2, zz = z * z2, xy = x * y2, xz = x * z2, yz = y * z2, wx = w * x2, wy = w * y2, wz = w * z2;
    M4 r = identity();
    at(r, 0, 0) = 1.0f - (yy + zz); at(r, 1, 0) = xy + wz; at(r, 2, 0) = xz - wy;
    at(r, 0, 1) = xy - wz; at(r, 1, 1) = 1.0f - (xx + zz); at(r, 2, 1) = yz + wx;
    at(r, 0, 2) = xz + wy; at(r, 1, 2) = yz - wx; at(r, 2, 2) = 1.0f - (xx + yy);
    return r;
}
// General inverse, in double, rounded once to float (Matrix4x4.inverse is native code; only its value matters).
inline M4 inverse(const M4& a) {
    double m[16], inv[16];
    for (int i = 0; i < 16; i++) m[i] = a.m[i];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    M4 r;
    if (det == 0) { memset(&r, 0, sizeof r); return r; }
    det = 1.0 / det;
    for (int i = 0; i < 16; i++) r.m[i] = (float)(inv[i] * det);
    return r;
}
// Utils.CreateWorldToLocalMatrix (Resources.cs:67-87)
inline M4 createWorldToLocal(const float p[3], const float rot[3], const float s[3]) {
    M4 translation = identity();
    at(translation, 0, 3) = -p[0]; at(translation, 1, 3) = -p[1]; at(translation, 2, 3) = -p[2];
    M4 rx = axisRotation(0, -rot[0]), ry = axisRotation(1, -rot[1]), rz = axisRotation(2, -rot[2]);
    M4 rotation = mulM(mulM(rz, ry), rx);
    M4 scale = identity();  // Func.Inverse(scale), Resources.cs:347-354
    at(scale, 0, 0) = s[0] != 0 ? 1.0f / s[0] : 0;
    at(scale, 1, 1) = s[1] != 0 ? 1.0f / s[1] : 0;
    at(scale, 2, 2) = s[2] != 0 ? 1.0f / s[2] : 0;
    return mulM(mulM(rotation, scale), translation);
}

// Matrix4x4.TRS(position, rotation, one): Transform.localToWorldMatrix of an unparented, unscaled object.
inline M4 fromPositionRotation(const float p[3], const float q[4]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float x2 = x * 2, y2 = y * 2, z2 = z * 2;
    float xx = x * x2, yy = y * y2, zz = z * z2, xy = x * y2, xz = x * z2, yz = y * z2, wx = w * x2, wy = w * y2, wz = w * z2;
    M4 r = identity();
    at(r, 0, 0) = 1.0f - (yy + zz); at(r, 1, 0) = xy + wz; at(r, 2, 0) = xz - wy;
    at(r, 0, 1) = xy - wz; at(r, 1, 1) = 1.0f - (xx + zz); at(r, 2, 1) = yz + wx;
    at(r, 0, 2) = xz + wy; at(r, 1, 2) = yz - wx; at(r, 2, 2) = 1.0f - (xx + yy);
    at(r, 0, 3) = p[0]; at(r, 1, 3) = p[1]; at(r, 2, 3) = p[2];
    return r;
}

}  // namespace
