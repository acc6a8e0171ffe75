// pth_math.h -- the slice of glam 0.30.2 the reference's host uses on this path
// (renderer.rs:970-980,1011-1015; scene/mod.rs:557-580), restated from glam's published
// scalar algorithms.  glam is not vendored in the reference: parity unpinned.
#pragma once

#include <cmath>

namespace pth {

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Vec3 operator+(Vec3 o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-(Vec3 o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3 operator*(float s) const { return {x * s, y * s, z * s}; }
    bool operator==(Vec3 o) const { return x == o.x && y == o.y && z == o.z; }
    float dot(Vec3 o) const { return (x * o.x + y * o.y) + z * o.z; }
    Vec3 cross(Vec3 o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    Vec3 normalize() const { float inv = 1.0f / std::sqrt(dot(*this)); return *this * inv; }
};

struct Quat {
    float x = 0, y = 0, z = 0, w = 1;
    bool operator==(Quat o) const { return x == o.x && y == o.y && z == o.z && w == o.w; }
    Quat normalize() const
    {
        float inv = 1.0f / std::sqrt(((x * x + y * y) + z * z) + w * w);
        return {x * inv, y * inv, z * inv, w * inv};
    }
    // glam Quat::from_rotation_y
    static Quat from_rotation_y(float angle) { float h = angle * 0.5f; return {0.0f, std::sin(h), 0.0f, std::cos(h)}; }
    // glam Quat::mul_quat (Hamilton product, this * o)
    Quat mul(Quat o) const
    {
        return {w * o.x + x * o.w + y * o.z - z * o.y, w * o.y - x * o.z + y * o.w + z * o.x,
                w * o.z + x * o.y - y * o.x + z * o.w, w * o.w - x * o.x - y * o.y - z * o.z};
    }
    // glam Quat::mul_vec3
    Vec3 mul_vec3(Vec3 v) const
    {
        Vec3 b(x, y, z);
        float b2 = b.dot(b);
        return v * (w * w - b2) + b * (v.dot(b) * 2.0f) + b.cross(v) * (w * 2.0f);
    }
};

struct Mat4 {
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // column-major

    // glam Mat4::from_scale_rotation_translation
    static Mat4 from_scale_rotation_translation(Vec3 s, Quat q, Vec3 t)
    {
        float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
        float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2, yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
        float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
        Mat4 r;
        float ax[3] = {1.0f - (yy + zz), xy + wz, xz - wy};
        float ay[3] = {xy - wz, 1.0f - (xx + zz), yz + wx};
        float az[3] = {xz + wy, yz - wx, 1.0f - (xx + yy)};
        for (int i = 0; i < 3; ++i) { r.m[i] = ax[i] * s.x; r.m[4 + i] = ay[i] * s.y; r.m[8 + i] = az[i] * s.z; }
        r.m[3] = r.m[7] = r.m[11] = 0.0f;
        r.m[12] = t.x; r.m[13] = t.y; r.m[14] = t.z; r.m[15] = 1.0f;
        return r;
    }
    // glam Mat4::perspective_rh (depth 0..1)
    static Mat4 perspective_rh(float fov_y, float aspect, float z_near, float z_far)
    {
        float sin_fov = std::sin(0.5f * fov_y), cos_fov = std::cos(0.5f * fov_y);
        float h = cos_fov / sin_fov, w = h / aspect, r = z_far / (z_near - z_far);
        Mat4 p;
        for (float &v : p.m) v = 0.0f;
        p.m[0] = w; p.m[5] = h; p.m[10] = r; p.m[11] = -1.0f; p.m[14] = r * z_near;
        return p;
    }
    // glam Mat4::look_to_rh
    static Mat4 look_to_rh(Vec3 eye, Vec3 dir, Vec3 up)
    {
        Vec3 f = dir.normalize();
        Vec3 s = f.cross(up).normalize();
        Vec3 u = s.cross(f);
        Mat4 v;
        v.m[0] = s.x; v.m[1] = u.x; v.m[2] = -f.x; v.m[3] = 0;
        v.m[4] = s.y; v.m[5] = u.y; v.m[6] = -f.y; v.m[7] = 0;
        v.m[8] = s.z; v.m[9] = u.z; v.m[10] = -f.z; v.m[11] = 0;
        v.m[12] = -eye.dot(s); v.m[13] = -eye.dot(u); v.m[14] = eye.dot(f); v.m[15] = 1.0f;
        return v;
    }
    Mat4 operator*(const Mat4 &b) const
    {
        Mat4 o;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r)
                o.m[c * 4 + r] = ((m[r] * b.m[c * 4] + m[4 + r] * b.m[c * 4 + 1]) + m[8 + r] * b.m[c * 4 + 2]) + m[12 + r] * b.m[c * 4 + 3];
        return o;
    }
    // glam Mat4::inverse (cofactor expansion after GLM)
    Mat4 inverse() const
    {
        float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3], m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
        float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11], m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
        float c00 = m22 * m33 - m32 * m23, c02 = m12 * m33 - m32 * m13, c03 = m12 * m23 - m22 * m13;
        float c04 = m21 * m33 - m31 * m23, c06 = m11 * m33 - m31 * m13, c07 = m11 * m23 - m21 * m13;
        float c08 = m21 * m32 - m31 * m22, c10 = m11 * m32 - m31 * m12, c11 = m11 * m22 - m21 * m12;
        float c12 = m20 * m33 - m30 * m23, c14 = m10 * m33 - m30 * m13, c15 = m10 * m23 - m20 * m13;
        float c16 = m20 * m32 - m30 * m22, c18 = m10 * m32 - m30 * m12, c19 = m10 * m22 - m20 * m12;
        float c20 = m20 * m31 - m30 * m21, c22 = m10 * m31 - m30 * m11, c23 = m10 * m21 - m20 * m11;
        float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
        float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
        float v0[4] = {m10, m00, m00, m00}, v1[4] = {m11, m01, m01, m01}, v2[4] = {m12, m02, m02, m02}, v3[4] = {m13, m03, m03, m03};
        const float sa[4] = {1.0f, -1.0f, 1.0f, -1.0f}, sb[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
        float inv[16];
        for (int i = 0; i < 4; ++i) {
            inv[i] = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i]) * sa[i];
            inv[4 + i] = ((v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i]) * sb[i];
            inv[8 + i] = ((v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i]) * sa[i];
            inv[12 + i] = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb[i];
        }
        float dot1 = ((m00 * inv[0] + m01 * inv[4]) + m02 * inv[8]) + m03 * inv[12];
        float rcp = 1.0f / dot1;
        Mat4 o;
        for (int i = 0; i < 16; ++i) o.m[i] = inv[i] * rcp;
        return o;
    }
};

}  // namespace pth
