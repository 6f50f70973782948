// Arithmetic of the hot path, written once for host and device and for float and double.
// Every function keeps the reference's operation order; the library is compiled with
// -ffp-contract=off so no multiply-add is fused on either side.
#pragma once
#include <math.h>
#include <stdint.h>

#include "rt_flat.h"

#ifdef __HIPCC__
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtm {

template <class R> struct V3 { R x, y, z; };

template <class R> RT_HD R rt_sqrt(R v);
template <> RT_HD float rt_sqrt<float>(float v) { return sqrtf(v); }
template <> RT_HD double rt_sqrt<double>(double v) { return sqrt(v); }
template <class R> RT_HD R rt_min(R a, R b);
template <> RT_HD float rt_min<float>(float a, float b) { return __builtin_fminf(a, b); }
template <> RT_HD double rt_min<double>(double a, double b) { return __builtin_fmin(a, b); }
template <class R> RT_HD R rt_max(R a, R b);
template <> RT_HD float rt_max<float>(float a, float b) { return __builtin_fmaxf(a, b); }
template <> RT_HD double rt_max<double>(double a, double b) { return __builtin_fmax(a, b); }
template <class R> RT_HD R rt_abs(R v);
template <> RT_HD float rt_abs<float>(float v) { return fabsf(v); }
template <> RT_HD double rt_abs<double>(double v) { return fabs(v); }

// Vect.h:43-75
template <class R> RT_HD V3<R> vnorm(V3<R> v) { R m = rt_sqrt<R>(v.x * v.x + v.y * v.y + v.z * v.z); return {v.x / m, v.y / m, v.z / m}; }
template <class R> RT_HD V3<R> vneg(V3<R> v) { return {-v.x, -v.y, -v.z}; }
template <class R> RT_HD R vdot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> RT_HD V3<R> vcross(V3<R> a, V3<R> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <class R> RT_HD V3<R> vadd(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> RT_HD V3<R> vsub(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> RT_HD V3<R> vmul(V3<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }
template <class R> RT_HD R vsqd(V3<R> a, V3<R> b) { return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z); }
template <class R> RT_HD V3<R> ld3(const R* p) { return {p[0], p[1], p[2]}; }

// BBox::Intersect (BVH.cpp:5-24), literal form: the ternaries define what happens when a
// direction component is 0 and 0 * inf produces a NaN.  Returns t0, or -1 on a miss.
template <class R>
RT_HD R slab_literal(const R* lo, const R* hi, const R* o, const R* d, R current_best)
{
    R t0 = 0, t1 = current_best;
    for (int i = 0; i < 3; i++) {
        R inv_ray = R(1) / d[i];
        R far_t = (lo[i] - o[i]) * inv_ray;
        R near_t = (hi[i] - o[i]) * inv_ray;
        if (near_t > far_t) { R tmp = near_t; near_t = far_t; far_t = tmp; }
        t0 = near_t > t0 ? near_t : t0;
        t1 = far_t > t1 ? t1 : far_t;
        if (t0 > t1) return R(-1);
    }
    return t0;
}

// Same test for rays whose reciprocal direction and origin are finite: no NaN can arise, so the
// ternaries reduce to min/max and the per-axis early exit to one final comparison (t0 never
// decreases, t1 never increases).  inv[i] = 1 / d[i] is the value the literal form computes.
template <class R>
RT_HD R slab_finite(const R* lo, const R* hi, const R* o, const R* inv, R current_best)
{
    R t0 = 0, t1 = current_best;
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int i = 0; i < 3; i++) {
        R a = (lo[i] - o[i]) * inv[i];
        R b = (hi[i] - o[i]) * inv[i];
        // no NaN can reach here, so min/max instructions give what the ternaries give (the sign of
        // a zero is never observed: the results are only compared)
        R n = rt_min<R>(a, b);
        R f = rt_max<R>(a, b);
        t0 = rt_max<R>(n, t0);
        t1 = rt_min<R>(f, t1);
    }
    return t0 > t1 ? R(-1) : t0;
}

// Triangle::findIntersection (Triangle.h:260-279) with e1 = B - A, e2 = C - A precomputed.
template <class R>
RT_HD R tri_intersect(const R* A, const R* e1, const R* e2, const R* o, const R* d)
{
    R qx = d[1] * e2[2] - d[2] * e2[1];
    R qy = d[2] * e2[0] - d[0] * e2[2];
    R qz = d[0] * e2[1] - d[1] * e2[0];
    R a = e1[0] * qx + e1[1] * qy + e1[2] * qz;
    if (rt_abs<R>(a) < R(0.00001)) return R(-1);
    R f = R(1) / a;
    R sx = o[0] - A[0], sy = o[1] - A[1], sz = o[2] - A[2];
    R u = f * (sx * qx + sy * qy + sz * qz);
    if (u < R(0)) return R(-1);
    R rx = sy * e1[2] - sz * e1[1];
    R ry = sz * e1[0] - sx * e1[2];
    R rz = sx * e1[1] - sy * e1[0];
    R v = f * (d[0] * rx + d[1] * ry + d[2] * rz);
    if (v < R(0) || u + v > R(1)) return R(-1);
    return f * (rx * e2[0] + ry * e2[1] + rz * e2[2]);
}

// The same test without branches, for SIMD execution: everything is computed, the reference's
// rejections are applied at the end as the exact negations of its conditions (so NaNs behave the
// same: a NaN never rejects, it flows into the returned t, which the caller's `t >= 0` drops).
template <class R>
RT_HD R tri_intersect_flat(const R* A, const R* e1, const R* e2, const R* o, const R* d)
{
    R qx = d[1] * e2[2] - d[2] * e2[1];
    R qy = d[2] * e2[0] - d[0] * e2[2];
    R qz = d[0] * e2[1] - d[1] * e2[0];
    R a = e1[0] * qx + e1[1] * qy + e1[2] * qz;
    R f = R(1) / a;
    R sx = o[0] - A[0], sy = o[1] - A[1], sz = o[2] - A[2];
    R u = f * (sx * qx + sy * qy + sz * qz);
    R rx = sy * e1[2] - sz * e1[1];
    R ry = sz * e1[0] - sx * e1[2];
    R rz = sx * e1[1] - sy * e1[0];
    R v = f * (d[0] * rx + d[1] * ry + d[2] * rz);
    R t = f * (rx * e2[0] + ry * e2[1] + rz * e2[2]);
    bool reject = (rt_abs<R>(a) < R(0.00001)) || (u < R(0)) || (v < R(0) || u + v > R(1));
    return reject ? R(-1) : t;
}

// The same test arranged for SIMD execution: the reciprocal (a correctly rounded division, by far the
// most expensive step) is postponed until the signs of the three numerators have been looked at.
// u = f * x with f = 1 / a is negative exactly when a and x have opposite signs, provided the
// product cannot underflow to zero; the guards (|a| < BIG, |x| > TINY) ensure |f * x| stays far
// inside the normal range, and any NaN makes a guard false and falls through to the full
// computation.  The same holds for v = f * y and t = f * z.  A lane that leaves early returns -1
// where the reference returns either -1 or a negative t; both are dropped by the caller's
// `t >= 0`.  Lanes that reach the end compute u, v and t exactly as tri_intersect does.
template <class R> struct SignGuard;
template <> struct SignGuard<float> { static constexpr float BIG = 1e15f, TINY = 1e-15f; };
template <> struct SignGuard<double> { static constexpr double BIG = 1e150, TINY = 1e-150; };

template <class R>
RT_HD R tri_intersect_staged(const R* A, const R* e1, const R* e2, const R* o, const R* d)
{
    R qx = d[1] * e2[2] - d[2] * e2[1];
    R qy = d[2] * e2[0] - d[0] * e2[2];
    R qz = d[0] * e2[1] - d[1] * e2[0];
    R a = e1[0] * qx + e1[1] * qy + e1[2] * qz;
    R abs_a = rt_abs<R>(a);
    if (abs_a < R(0.00001)) return R(-1);
    const bool guarded = abs_a < SignGuard<R>::BIG;
    const bool a_neg = a < R(0);
    R sx = o[0] - A[0], sy = o[1] - A[1], sz = o[2] - A[2];
    R x = sx * qx + sy * qy + sz * qz;
    if (guarded && rt_abs<R>(x) > SignGuard<R>::TINY && (x < R(0)) != a_neg) return R(-1);   // u < 0
    R rx = sy * e1[2] - sz * e1[1];
    R ry = sz * e1[0] - sx * e1[2];
    R rz = sx * e1[1] - sy * e1[0];
    R y = d[0] * rx + d[1] * ry + d[2] * rz;
    if (guarded && rt_abs<R>(y) > SignGuard<R>::TINY && (y < R(0)) != a_neg) return R(-1);   // v < 0
    R z = rx * e2[0] + ry * e2[1] + rz * e2[2];
    if (guarded && rt_abs<R>(z) > SignGuard<R>::TINY && (z < R(0)) != a_neg) return R(-1);   // t < 0
    R f = R(1) / a;
    R u = f * x;
    if (u < R(0)) return R(-1);
    R v = f * y;
    if (v < R(0) || u + v > R(1)) return R(-1);
    return f * z;
}

// Sphere::findIntersection (Sphere.h:44-90): origin pre-offset by 0.1 * dir, and the second root
// exactly as written there.
template <class R>
RT_HD R sphere_intersect(const R* c, R radius, const R* o, const R* d)
{
    R ox = o[0] + R(0.1) * d[0], oy = o[1] + R(0.1) * d[1], oz = o[2] + R(0.1) * d[2];
    R b = (R(2) * (ox - c[0]) * d[0]) + (R(2) * (oy - c[1]) * d[1]) + (R(2) * (oz - c[2]) * d[2]);
    R cc = (ox - c[0]) * (ox - c[0]) + (oy - c[1]) * (oy - c[1]) + (oz - c[2]) * (oz - c[2]) - radius * radius;
    R disc = b * b - R(4) * cc;
    if (disc > R(0)) {
        R root_1 = ((R(-1) * b - rt_sqrt<R>(disc)) / R(2)) - R(0.000001);
        if (root_1 > R(0)) return root_1;
        return ((R(-1) * b + rt_sqrt<R>(disc) - b) / R(2)) - R(0.000001);
    }
    return R(-1);
}

// Plane::findIntersection (Plane.h:26-39)
template <class R>
RT_HD R plane_intersect(const R* n, R distance, const R* o, const R* d)
{
    R a = d[0] * n[0] + d[1] * n[1] + d[2] * n[2];
    if (a == R(0)) return R(-1);
    R px = o[0] + (-(n[0] * distance)), py = o[1] + (-(n[1] * distance)), pz = o[2] + (-(n[2] * distance));
    R b = n[0] * px + n[1] * py + n[2] * pz;
    return R(-1) * b / a;
}

template <class R>
RT_HD R prim_intersect(const rtf::Prim<R>& p, const R* o, const R* d)
{
    if (p.kind == rtf::PRIM_TRI) return tri_intersect<R>(p.a, p.e1, p.e2, o, d);
    if (p.kind == rtf::PRIM_SPHERE) return sphere_intersect<R>(p.a, p.e1[0], o, d);
    return plane_intersect<R>(p.a, p.e1[0], o, d);
}

// ---------------------------------------------------------------- shading arithmetic (double)
typedef V3<double> D3;

// Triangle::getTriangleNormal, Triangle.h:223-227
RT_HD D3 tri_normal(const rtf::ObjShade& p)
{
    D3 CA = {p.C[0] - p.A[0], p.C[1] - p.A[1], p.C[2] - p.A[2]};
    D3 BA = {p.B[0] - p.A[0], p.B[1] - p.A[1], p.B[2] - p.A[2]};
    return vnorm(vcross(CA, BA));
}

// Triangle::getProjectPosition, Triangle.h:157-162
RT_HD D3 tri_project(const rtf::ObjShade& p, D3 above)
{
    D3 A = ld3(p.A);
    D3 line_ap = vadd(above, vneg(A));
    D3 n = tri_normal(p);
    return vadd(vadd(vneg(vmul(n, vdot(line_ap, n))), line_ap), A);
}

// The barycentric cascade of Triangle.h:77-97 and :193-213, exact `!= 0` tests and the third
// branch as written there.
RT_HD void tri_bary(const rtf::ObjShade& p, D3 position, double& b, double& c)
{
    double AP_x = position.x - p.A[0], AP_y = position.y - p.A[1], AP_z = position.z - p.A[2];
    double AB_x = p.B[0] - p.A[0], AB_y = p.B[1] - p.A[1], AB_z = p.B[2] - p.A[2];
    double AC_x = p.C[0] - p.A[0], AC_y = p.C[1] - p.A[1], AC_z = p.C[2] - p.A[2];
    if (AB_y * AC_x - AC_y * AB_x != 0) c = (AP_x * AB_y - AP_y * AB_x) / (AB_y * AC_x - AC_y * AB_x);
    else if (AC_y * AB_z - AC_z * AB_y != 0) c = (AP_y * AB_z - AP_z * AB_y) / (AC_y * AB_z - AC_z * AB_y);
    else c = (AP_x * AB_z - AP_z - AB_x) / (AC_x * AB_z - AC_z * AB_x);
    if (AB_x != 0) b = (AP_x - c * AC_x) / AB_x;
    else if (AB_y != 0) b = (AP_y - c * AC_y) / AB_y;
    else b = (AP_z - c * AC_z) / AB_z;
}

// Triangle::getNormalAt / getTriangleSmoothNormal (Triangle.h:164-240), Sphere::getNormalAt
// (Sphere.h:33-36), Plane::getNormalAt (Plane.h:22-24)
RT_HD D3 obj_normal(const rtf::ObjShade& p, D3 pt)
{
    if (p.kind == rtf::PRIM_TRI) {
        if (!p.has_normal) return tri_normal(p);
        D3 position = tri_project(p, pt);
        D3 NA = ld3(p.NA);
        D3 NAB = vadd(ld3(p.NB), vneg(NA));
        D3 NAC = vadd(ld3(p.NC), vneg(NA));
        double b, c;
        tri_bary(p, position, b, c);
        D3 r = vnorm(vadd(NA, vadd(vmul(NAB, b), vmul(NAC, c))));
        if (vdot(r, tri_normal(p)) < 0) r = vneg(r);
        return r;
    }
    if (p.kind == rtf::PRIM_SPHERE) return vnorm(vadd(pt, vneg(ld3(p.A))));
    return ld3(p.A);
}

// FresnelEquation, main.cpp:111-123 (single-precision arithmetic as there)
RT_HD double fresnel(float cosThetaI, float n1, float n2, int type)
{
    float sin2ThetaT = (n1 / n2) * (n1 / n2) * (1 - cosThetaI * cosThetaI);
    float cosThetaT = sqrtf(1 - sin2ThetaT);
    float Rs = (n1 * cosThetaI - n2 * cosThetaT) / (n1 * cosThetaI + n2 * cosThetaT);
    float Rp = (n2 * cosThetaI - n1 * cosThetaT) / (n2 * cosThetaI + n1 * cosThetaT);
    if (type == 0) return (Rs * Rs + Rp * Rp) / 2;
    return 1 - (Rs * Rs + Rp * Rp) / 2;
}

// Triangle::getTangentAt (Triangle.h:242-256), Sphere::getTangentAt (Sphere.h:40-42)
RT_HD D3 obj_tangent(const rtf::ObjShade& p, D3 pt)
{
    if (p.kind == rtf::PRIM_TRI) {
        if (!p.has_normal) return vnorm(D3{p.C[0] - p.A[0], p.C[1] - p.A[1], p.C[2] - p.A[2]});
        D3 n = obj_normal(p, pt);
        if (n.y == 0 && n.z == 0) return D3{0, 1, 0};
        return vnorm(vcross(n, D3{1, 0, 0}));
    }
    return D3{1, 0, 0};
}

// TextureMap::GetColor (TextureMap.h:127-182): wrap by +-1 steps, bilinear, u indexes rows.
RT_HD void tex_lookup(const double* texels, int x_res, int y_res, double x, double y, double* rgb)
{
    double dot_x = x, dot_y = y;
    if (dot_x > 0) { while (dot_x - 1 > 0) dot_x -= 1; }
    else if (dot_x < 0) { do { dot_x += 1; } while (dot_x < 0); }
    if (dot_y > 0) { while (dot_y - 1 > 0) dot_y -= 1; }
    else if (dot_y < 0) { do { dot_y += 1; } while (dot_y < 0); }
    dot_x = dot_x * (x_res - 1);
    dot_y = dot_y * (y_res - 1);
    int floor_x = int(dot_x);
    int ceil_x = (floor_x + 1) >= x_res ? x_res - 1 : floor_x + 1;
    int floor_y = int(dot_y);
    int ceil_y = (floor_y + 1) >= y_res ? y_res - 1 : floor_y + 1;
    double s = dot_x - floor_x, t = dot_y - floor_y;
    for (int ch = 0; ch < 3; ch++) {
        double lu = texels[((int64_t)floor_x * y_res + floor_y) * 3 + ch];
        double ld = texels[((int64_t)ceil_x * y_res + floor_y) * 3 + ch];
        double ru = texels[((int64_t)floor_x * y_res + ceil_y) * 3 + ch];
        double rd = texels[((int64_t)ceil_x * y_res + ceil_y) * 3 + ch];
        rgb[ch] = lu * (1 - s) * (1 - t) + ld * s * (1 - t) + ru * (1 - s) * t + rd * s * t;
    }
}

// Triangle::getColor(pos) / getTextureColor (Triangle.h:29-104); out = r, g, b, special
RT_HD void obj_color(const rtf::ObjShade& p, const rtf::TexDesc* maps, const double* texel_pool, D3 above, double* out)
{
    if (p.kind == rtf::PRIM_TRI && p.has_texture && p.map >= 0) {
        D3 position = tri_project(p, above);
        double b, c;
        tri_bary(p, position, b, c);
        double ABx = p.tB[0] + p.tA[0] * (-1), ABy = p.tB[1] + (-1) * p.tA[1];
        double ACx = p.tC[0] + p.tA[0] * (-1), ACy = p.tC[1] + (-1) * p.tA[1];
        double u = (ABx * b + ACx * c) + p.tA[0];
        double v = (ABy * b + ACy * c) + p.tA[1];
        const rtf::TexDesc& m = maps[p.map];
        tex_lookup(texel_pool + m.offset, m.x_res, m.y_res, u, v, out);
        out[3] = 0;
        return;
    }
    out[0] = p.color[0]; out[1] = p.color[1]; out[2] = p.color[2]; out[3] = p.color[3];
}

// ---- sampling extras of the shade loop (main.cpp:139-156 glossy rays, :792-810 lens jitter).  The
// reference draws from one std::mt19937 shared by all threads; here every draw is a pure function of
// (where on the path, which draw, seed), so a frame does not depend on scheduling.
RT_HD double counter_uniform(uint32_t a, uint32_t b, uint32_t seed)     // uniform in (0, 1)
{
    uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ seed * 0xC2B2AE3Du;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return ((double)x + 0.5) * (1.0 / 4294967296.0);
}
RT_HD uint32_t float_bits(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
// identifies a point of a path whatever the order in which paths are processed
RT_HD uint32_t path_key(D3 p) { return float_bits((float)p.x) * 0x9E3779B1u ^ float_bits((float)p.y) * 0x85EBCA77u ^ float_bits((float)p.z) * 0xC2B2AE3Du; }

// the material switch of main.cpp:177: special * 100 has the fractional part .7654
RT_HD bool glossy_special(double special)
{
    double f = special * 100.0 - floor(special * 100.0);
    return f - 0.7654 < 0.0001 && f - 0.7653 > 0;
}
constexpr int GLOSSY_SAMPLE = 16;      // main.cpp:32

// computeGlossyRay, main.cpp:139-156 (a, b: the two uniform draws)
RT_HD D3 glossy_ray(D3 reflection_direction, D3 normal, double glossiness, double a, double b)
{
    double theta = acos(pow(1 - a, glossiness));
    double phi = 2 * 3.1415926 * b;
    double x = sin(theta) * cos(phi);
    double z = sin(theta) * sin(phi);
    D3 u = vcross(reflection_direction, normal);
    D3 v = vcross(reflection_direction, u);
    return vnorm(vadd(vmul(u, x), vadd(vmul(v, z), reflection_direction)));
}

// computeReflectionDir, main.cpp:125-136
RT_HD D3 reflect_dir(D3 normal, D3 ray_in)
{
    double dot1 = vdot(normal, vneg(ray_in));
    if (dot1 <= 0.02) return ray_in;
    D3 scalar1 = vmul(normal, dot1);
    D3 add1 = vadd(scalar1, ray_in);
    D3 scalar2 = vmul(add1, 2.0);
    D3 add2 = vadd(vneg(ray_in), scalar2);
    return vnorm(add2);
}

// Camera-ray direction of main.cpp:753-790 for pixel (x, y), AA sample aax (both offsets use aax,
// main.cpp:775-785), depth of field off.
RT_HD D3 camera_ray_dir(const double* camDir, const double* camRight, const double* camDown, int width, int height, int aadepth,
                        int x, int y, int aax)
{
    double aspectRatio = (double)width / (double)height;
    double xamnt, yamnt;
    double off = aadepth == 1 ? 0.5 : (double)aax / ((double)aadepth - 1);
    if (width > height) {
        xamnt = ((x + off) / width) * aspectRatio - (((width - height) / (double)height) / 2);
        yamnt = ((height - y) + off) / height;
    } else if (height > width) {
        xamnt = (x + off) / width;
        yamnt = (((height - y) + off) / height) / aspectRatio - (((height - width) / (double)width) / 2);
    } else {
        xamnt = (x + off) / width;
        yamnt = ((height - y) + off) / height;
    }
    D3 r = vmul(ld3(camRight), xamnt - 0.5);
    D3 dn = vmul(ld3(camDown), yamnt - 0.5);
    return vnorm(vadd(ld3(camDir), vadd(r, dn)));
}

}  // namespace rtm
