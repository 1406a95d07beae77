// raytracing_amd/csrc/rt_device.h -- device-side scene layout, RNG, traversal, shading and kernels.
//
// Hand-written HIP for gfx950 (wave64).  Everything is templated on the scalar R:
//   R = double : RT_MODE_REFERENCE_FP64, compiled with -ffp-contract=off (rt_kernels_fp64.hip); evaluation
//                order follows the reference line by line so closest hits are bit-identical to its g++ build.
//   R = float  : RT_MODE_FAST_FP32 (rt_kernels_fp32.hip), FMA contraction allowed.
// Reference citations are paths under the reference's src/cpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_algebra/vec3.hpp"
#include "rtcore.h"

namespace rt {

// ------------------------------------------------------------------ math overloads
// sin and cos of one argument in [0, 2 pi] -- the only sines and cosines a path takes outside the checker texture are those of
// phi = 2 pi r1 in random_cosine (util.hpp:98-107).  The device library's sincos is built for every double: ~150 vector instructions with
// the large-argument reduction inlined beside the path taken; this one is ~55, branch-free: n = rint(x 2 / pi) <= 4, the remainder as
// a double-double by a Cody-Waite reduction against pi / 2 in 33-bit pieces (n times a piece is exact), then the kernel polynomials
// of fdlibm (k_sin.c, k_cos.c, e_rem_pio2.c: (C) 1993 Sun Microsystems, freely distributable; the constants are theirs, the code
// is written out here without their branches).  Measured against 80-bit sinl / cosl on 30 M arguments 2 pi k / 2^32: <= 0.85 ulp;
// carries the host libm's bits on 96.7 % of them (the device library: 96.8 %), the rest 1 ulp away (tools/libm_agreement.py).
__device__ inline void sincos_turn(double x, double& s, double& c)
{
    const double PIO2_1 = 1.57079632673412561417e+00, PIO2_2 = 6.07710050630396597660e-11, PIO2_2T = 2.02226624879595063154e-21, INVPIO2 = 6.36619772367581382433e-01;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double fn = __builtin_rint(x * INVPIO2);
    const int n = (int) fn;
    const double t = x - fn * PIO2_1, w = fn * PIO2_2, r = t - w;
    const double w2 = fn * PIO2_2T - ((t - r) - w);
    const double y0 = r - w2, y1 = (r - y0) - w2;                       // x - n pi / 2 = y0 + y1
    const double z = y0 * y0, v = z * y0;
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double ax = fabs(y0);
    // k_cos.c: 1 - (z / 2 - (z rc - y0 y1)) below 0.3; above, a piece qx ~ |y0| / 4 (its high word, exponent - 2) is moved between the terms
    const double qbits = __longlong_as_double((long long) ((((unsigned long long) __double_as_longlong(ax) >> 32) - 0x00200000ull) << 32));
    const double qx = ax < 0.3 ? 0.0 : (ax > 0.78125 ? 0.28125 : qbits);
    const double kc = (1.0 - qx) - ((0.5 * z - qx) - (z * rc - y0 * y1));
    const double ss = (n & 1) ? kc : ks, cc = (n & 1) ? ks : kc;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}
__device__ inline float  rt_sin(float v) { return sinf(v); }
__device__ inline double rt_sin(double v) { if (v >= 0.0 && v <= 6.3) { double s, c; sincos_turn(v, s, c); return s; } return sin(v); }
__device__ inline float  rt_cos(float v) { return cosf(v); }
__device__ inline double rt_cos(double v) { if (v >= 0.0 && v <= 6.3) { double s, c; sincos_turn(v, s, c); return c; } return cos(v); }
__device__ inline float  rt_log(float v) { return logf(v); }
__device__ inline double rt_log(double v) { return log(v); }
__device__ inline float  rt_acos(float v) { return acosf(v); }
__device__ inline double rt_acos(double v) { return acos(v); }
__device__ inline float  rt_atan2(float a, float b) { return atan2f(a, b); }
__device__ inline double rt_atan2(double a, double b) { return atan2(a, b); }
__device__ inline float  rt_pow(float a, float b) { return powf(a, b); }
__device__ inline double rt_pow(double a, double b) { return pow(a, b); }
__device__ inline float  rt_abs(float v) { return fabsf(v); }
__device__ inline double rt_abs(double v) { return fabs(v); }
__device__ inline float  rt_min(float a, float b) { return fminf(a, b); }
__device__ inline double rt_min(double a, double b) { return fmin(a, b); }
__device__ inline bool rt_finite(float v) { return isfinite(v); }
__device__ inline bool rt_finite(double v) { return isfinite(v); }

template <typename R> struct consts;
template <> struct consts<double> {
    static constexpr double pi = 3.141592653589;   // util.hpp:13 (truncated on purpose)
    static constexpr double eps = 1e-7;            // util.hpp:14
    static constexpr double tmin = 1e-3;           // ray.hpp:12
};
template <> struct consts<float> {
    static constexpr float pi = 3.141592653589f;
    static constexpr float eps = 1e-7f;
    static constexpr float tmin = 1e-3f;
};
template <typename R> __device__ inline R rt_inf() { return R(__builtin_huge_val()); }

// ------------------------------------------------------------------ device scene layout (HBM; small scenes are staged into LDS)
// Reference-order traversal program: the reference tree (bvh.hpp) written out in pre-order.  An entry is either a
// bvh node (prim < 0: test the box, on a miss jump to `skip`) or a leaf reference (prim >= 0: intersect it).
// Walking it front to back with one running t_max IS bvh::intersect's recursion (bvh.hpp:58-68): left subtree,
// then right subtree with t_max shrunk; a span-1 node's leaf appears twice, as it is tested twice.
template <typename R> struct alignas(8) Op {
    R mn[3], mx[3];
    int32_t skip;
    int32_t prim;
};
// hot primitive record (intersection only).  g[]: sphere c,r | msphere c0,c1,start,end,r | plane a0,a1,b0,b1,k | triangle p0,e1,e2
// 48 B in fp32: three 16-byte loads fetch any primitive.
template <typename R> struct alignas(16) Prim {
    R g[9];
    int32_t kf;                     // kind | flags << 8 | (instance transform + 1) << 16
    int32_t mat, id;                // id: primitive id (index into tris[]); records may be stored in any order
    __host__ __device__ int kind() const { return kf & 0xff; }
    __host__ __device__ int flags() const { return (kf >> 8) & 0xff; }
    __host__ __device__ int xform() const { return (int) (((uint32_t) kf) >> 16) - 1; }   // instance transform (Scene::xforms), -1: none
};
// rotatey / translate around an instance (transform.hpp:9-121): sin, cos as the rotatey constructor computes them, translate target
template <typename R> struct Xform { R sn, cs, t[3]; int32_t flags; };
// Fast-mode BVH node (64 B, one cache line): both children's boxes live in the parent, so one fetch decides both.
// Child code c >= 0: inner node c; c < 0: leaf, ~c = first_record * 8 + (count - 1) into Scene::fprims (<= 8 records).
struct alignas(16) BNode {
    float lb[6];        // left  child box: min xyz, max xyz (rounded outward from fp64)
    float rb[6];        // right child box
    int32_t left, right;
    int32_t pad[2];
};
// Quantised four-wide node (64 B): the binary tree collapsed (rt_bvh_build.h widen()), so a traversal makes half as many dependent
// round trips to memory and LDS (mesh workload: 8.8 node visits per ray instead of 16.6, 18.1 box tests instead of 17.2).
// Per axis a: o[a] = a float origin <= all children's minima whose low byte is the (signed) exponent e of a power-of-two step -- the
// kernel uses the whole dword as the origin, exponent bits included, and the builder accounts for them; lo[a] / hi[a]: byte k = child
// k's low / high plane in steps from the origin (lo rounded down, hi up, after padding by the fp32 decode error, so the decoded boxes
// contain the real ones).  lo <= hi byte-wise, so which plane a ray meets first follows from the sign of its direction alone: the
// near planes are lo[a] for a ray going up the axis, else hi[a] -- no min / max per plane pair.  child[k]: child code as in BNode; an
// unused slot names the dummy leaf (rtcore.hip: one record no ray can hit), so a visit tests nothing per child but its box.
// Slot order is visiting order: the builder sorts the children by their centres along one axis; a ray going up that axis takes the
// slots forwards, a ray going down it backwards (no sort by entry distance in the kernel).  n: that axis times 0x01010101 (a v_perm_b32
// selector, rt_wave.h node_step); pad[0]: children in use (2..4; 1 for a single-leaf tree).
struct alignas(16) WNode {
    uint32_t o[3];
    uint32_t lo[3];
    uint32_t hi[3];
    int32_t child[4];
    uint32_t n, pad[2];
};
#define RT_Q_SLOTS_MAX 256          // rt_queue.h: path slots per wave (list entries are one byte)
#define RT_BVH_SENTINEL ((int32_t) 0x80000000)
#define RT_BVH_STACK_MAX 64
#ifndef RT_WAVES_FP32
#define RT_WAVES_FP32 4   // waves per SIMD the fp32 render kernels are compiled for (<= 128 VGPRs)
#endif

// per-primitive side record, by primitive id.  Triangles: vertex normals and uvs (cold: read once per hit).  Boxes: n[0..2] is the
// translate target (read by the box test).
template <typename R> struct TriShade { R n[9]; R uv[6]; R area; };      // area: light_tris only -- length(cross(e1, e2)) / 2 (polygon.hpp:146), a constant of the scene, set by rt_scene_create
template <typename R> struct Mat { int32_t kind, tex; R d[16]; };
template <typename R> struct Tex { int32_t kind, a, b, ch, color_based; uint32_t offset; R rgb[3]; R pad; };
// what a primitive test may need besides the hot record: side records by primitive id, instance transforms
template <typename R> struct Side {
    const TriShade<R>* tris; const Xform<R>* xforms;
    __device__ const TriShade<R>& operator[](int i) const { return tris[i]; }
};
template <typename R> struct Camera { vec3<R> pos, left_bottom, horizontal, vertical, u, v; R lens_radius, s0, s1; };

template <typename R> struct Scene {
    const Op<R>* ops; int32_t n_ops;
    const Prim<R>* prims; int32_t n_prims;
    const TriShade<R>* tris;          // n_prims entries, or null when the scene has neither triangles nor boxes
    const Mat<R>* mats;
    const Tex<R>* texs;
    const uint8_t* pixels;
    const Prim<R>* lights; const TriShade<R>* light_tris; int32_t n_lights;
    R n_lights_r, inv_n_lights;       // R(n_lights) and R(1) / R(n_lights) (group.hpp:107, :117), taken once by rt_scene_create: left to the kernels they are loop invariants held in vector registers for the whole launch
    int32_t need_uv;                  // some material reads uv (image texture)
    const BNode* bnodes; const Prim<R>* fprims; int32_t n_bnodes; int32_t bvh_depth;   // fast mode: SAH tree built by the library, records in leaf order
    const WNode* wnodes; int32_t wide_stack;   // the same tree collapsed to quantised four-wide nodes; stack entries a traversal of it can need
    int32_t sorted_visit;             // wide nodes: visit the hit children by entry distance (a five-exchange network) instead of in slot order -- chosen per
                                      // tree by rtcore.hip (rt_bvh_build.h visit_order_census): a sparse soup needs it, a surface mesh does not
    const Xform<R>* xforms;           // instance transforms (null: none)
    const uint32_t* ref_rank;         // by primitive id: index of the LAST entry of `ops` that names it, i.e. when the reference's walk visits it last.
                                      // Among surfaces at the same t (bit for bit) the reference ends up with the one it visits last (its range test
                                      // lets t == t_max pass: polygon.hpp:113, primitive.hpp:42); the fast modes break such ties by this rank (take_hit)
    Camera<R> cam;
    __host__ __device__ Side<R> side() const { Side<R> s; s.tris = tris; s.xforms = xforms; return s; }
};

// `from`: id of the primitive the origin lies on (-1: none).  Only the fp32 modes read it, see prim_t.
template <typename R> struct Ray { vec3<R> o, d; R time = R(0); int32_t from = -1; };   // every member has a value from construction: no indeterminate lanes

// ------------------------------------------------------------------ counter-based RNG
// PCG32 XSH-RR over a state hashed from (seed, pixel, sample): nothing is stored between samples or launches,
// so any (pixel, sample) can be rendered on any GPU in any order.  Bit-identical to oracle/rt_oracle.c.
__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t s;
    __host__ __device__ void seed(uint64_t seed, uint32_t pixel, uint32_t sample)
    {
        uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ULL * (1ULL + pixel));
        s = mix64(h ^ (0xD1B54A32D192ED03ULL * (1ULL + sample)));
    }
    __host__ __device__ uint32_t next()
    {
        uint64_t old = s;
        s = old * 6364136223846793005ULL + 1442695040888963407ULL;
        uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t) (old >> 59);
        return (xs >> rot) | (xs << ((32 - rot) & 31));
    }
};
template <typename R> __device__ inline R uniform(Rng& g);
template <> __device__ inline double uniform<double>(Rng& g) { return g.next() * (1.0 / 4294967296.0); }        // util.hpp:30-33
template <> __device__ inline float uniform<float>(Rng& g) { return (g.next() >> 8) * (1.0f / 16777216.0f); }
template <typename R> __device__ inline R uniform(Rng& g, R mn, R mx) { return mn + (mx - mn) * uniform<R>(g); } // util.hpp:36-39

// Draw order inside these helpers is the reference's as compiled by g++ (constructor arguments evaluate right to
// left); it is pinned by the golden fixtures through the oracle.
template <typename R> __device__ inline vec3<R> random_sphere(Rng& g)          // util.hpp:60-70
{
    for (;;) {
        R z = uniform<R>(g, R(-1), R(1)), y = uniform<R>(g, R(-1), R(1)), x = uniform<R>(g, R(-1), R(1));
        vec3<R> p(x, y, z);
        R l = length(p);
        if (l * l >= R(1)) continue;
        return normalize(p);
    }
}
template <typename R> __device__ inline vec3<R> random_disk(Rng& g)            // util.hpp:85-95
{
    for (;;) {
        R y = uniform<R>(g, R(-1), R(1)), x = uniform<R>(g, R(-1), R(1));
        vec3<R> p(x, y, R(0));
        R l = length(p);
        if (l * l >= R(1)) continue;
        return p;
    }
}
template <typename R> __device__ inline vec3<R> random_cosine(Rng& g)          // util.hpp:98-107
{
    R r1 = uniform<R>(g), r2 = uniform<R>(g);
    R z = rt_sqrt(R(1) - r2);
    R phi = R(2) * consts<R>::pi * r1;                    // in [0, 2 pi): r1 = k / 2^32
    R sn, cs;
    sincos_turn(phi, sn, cs);
    R x = cs * rt_sqrt(r2);
    R y = sn * rt_sqrt(r2);
    return vec3<R>(x, y, z);
}
// fp32: phi = 2*pi*r1 with r1 in [0,1): the hardware sine/cosine take their argument in revolutions (v_sin_f32 / v_cos_f32),
// so no range reduction and no 2*pi multiply are needed.
template <> __device__ inline vec3<float> random_cosine<float>(Rng& g)
{
    float r1 = uniform<float>(g), r2 = uniform<float>(g);
    float z = sqrtf(1.0f - r2), s = sqrtf(r2);
    return vec3<float>(__builtin_amdgcn_cosf(r1) * s, __builtin_amdgcn_sinf(r1) * s, z);
}
template <typename R> __device__ inline vec3<R> reflect(const vec3<R>& in, const vec3<R>& n) { return in - (R(2) * dot(in, n)) * n; } // util.hpp:154-157
template <typename R> __device__ inline vec3<R> refract(const vec3<R>& in, const vec3<R>& n, R ratio)                                  // util.hpp:160-166
{
    R cos_theta = rt_min(-dot(in, n), R(1));
    vec3<R> perp = ratio * (in + cos_theta * n);
    R pl = length(perp);
    vec3<R> par = (-rt_sqrt(rt_abs(R(1) - pl * pl))) * n;
    return perp + par;
}

// ------------------------------------------------------------------ intersection
template <typename R> __device__ inline bool box_hit(const Op<R>& b, const Ray<R>& r, R t_min, R t_max)   // aabb.hpp:23-41
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        R inv = R(1) / r.d[i];
        R t0 = (b.mn[i] - r.o[i]) * inv;
        R t1 = (b.mx[i] - r.o[i]) * inv;
        if (inv < R(0)) { R s = t0; t0 = t1; t1 = s; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return false;
    }
    return true;
}

template <typename R> __device__ inline vec3<R> msphere_center(const Prim<R>& p, R time)                  // primitive.hpp:98-101
{
    vec3<R> c0(p.g[0], p.g[1], p.g[2]), c1(p.g[3], p.g[4], p.g[5]);
    return c0 + ((time - p.g[6]) / (p.g[7] - p.g[6])) * (c1 - c0);
}

template <typename R> __device__ inline bool sphere_t(const vec3<R>& center, R radius, const Ray<R>& r, R t_min, R t_max, R& t)
{                                                                                                         // primitive.hpp:31-48
    vec3<R> oc = r.o - center;
    R a = dot(r.d, r.d);
    R b = R(2) * dot(oc, r.d);
    R c = dot(oc, oc) - radius * radius;
    R disc = b * b - R(4) * a * c;
    if (disc < R(0)) return false;
    R sq = rt_sqrt(disc);
    R root = (-b - sq) / (a * R(2));
    if (root < t_min || root > t_max) {
        root = (-b + sq) / (a * R(2));
        if (root < t_min || root > t_max) return false;
    }
    t = root;
    return true;
}

// fp32 fast mode: `oc.oc - r*r` cancels catastrophically for a big sphere hit from its own surface (the r = 1000
// ground sphere: absolute error ~0.06 in fp32 => self-intersections past t_min = 1e-3).  The three cancelling
// quantities b, c and the discriminant are formed in fp64 (gfx950 runs vector fp64 at half the fp32 rate, and
// leaf tests are ~1/7 of the box tests), everything else, including the sqrt and the roots, stays fp32; the
// roots use the cancellation-free form q = -(b/2 + sign(b) sqrt(disc/4)), t0 = q/a, t1 = c/q.
template <> __device__ inline bool sphere_t<float>(const vec3<float>& center, float radius, const Ray<float>& r, float t_min, float t_max, float& t)
{
    double ox = (double) r.o.x - (double) center.x, oy = (double) r.o.y - (double) center.y, oz = (double) r.o.z - (double) center.z;
    double dx = r.d.x, dy = r.d.y, dz = r.d.z;
    double a = dx * dx + dy * dy + dz * dz;
    double hb = ox * dx + oy * dy + oz * dz;                       // b / 2
    double c = ox * ox + oy * oy + oz * oz - (double) radius * (double) radius;
    double qd = hb * hb - a * c;                                    // disc / 4
    if (qd < 0.0) return false;
    float sq = sqrtf((float) qd);
    float hbf = (float) hb, af = (float) a, cf = (float) c;
    float q = hbf < 0.0f ? sq - hbf : -(hbf + sq);                  // -(b/2 + sign(b) sqrt(disc/4))
    float r0 = q / af, r1 = cf / q;
    float lo = fminf(r0, r1), hi = fmaxf(r0, r1);
    if (q == 0.0f) { lo = hi = 0.0f; }
    float root = lo;
    if (root < t_min || root > t_max) {
        root = hi;
        if (root < t_min || root > t_max) return false;
    }
    t = root;
    return true;
}

__device__ inline void plane_axes(int kind, int& ax, int& bx, int& kx)
{
    if (kind == RT_PRIM_PLANEXY) { ax = 0; bx = 1; kx = 2; }
    else if (kind == RT_PRIM_PLANEXZ) { ax = 0; bx = 2; kx = 1; }
    else { ax = 1; bx = 2; kx = 0; }
}
// components of v along the rect's two in-plane axes (a, b) and its normal axis (k).  Written as selects on the kind: indexing a
// vec3 with a runtime index makes the compiler keep the whole ray in scratch memory.
template <typename R> __device__ inline void plane_split(int kind, const vec3<R>& v, R& a, R& b, R& k)
{
    const bool xy = kind == RT_PRIM_PLANEXY, xz = kind == RT_PRIM_PLANEXZ;
    a = (xy || xz) ? v.x : v.y;
    b = xy ? v.y : v.z;
    k = xy ? v.z : (xz ? v.y : v.x);
}

template <typename R> __device__ inline bool plane_t(const Prim<R>& p, const Ray<R>& r, R t_min, R t_max, R& t, R& a, R& b)
{                                                                                                         // primitive.hpp:176-187 / 227-238 / 296-307
    R oa, ob, ok, da, db, dk;
    plane_split(p.kind(), r.o, oa, ob, ok);
    plane_split(p.kind(), r.d, da, db, dk);
    R tt = (p.g[4] - ok) / dk;
    if (tt < t_min || tt > t_max) return false;
    a = oa + tt * da;
    b = ob + tt * db;
    if (a < p.g[0] || a > p.g[1] || b < p.g[2] || b > p.g[3]) return false;
    t = tt;
    return true;
}

template <typename R> __device__ inline bool tri_t(const Prim<R>& p, const Ray<R>& r, R t_min, R t_max, R& t, R& u, R& v)
{                                                                                                         // polygon.hpp:93-115
    vec3<R> p0(p.g[0], p.g[1], p.g[2]), e1(p.g[3], p.g[4], p.g[5]), e2(p.g[6], p.g[7], p.g[8]);
    vec3<R> q = cross(r.d, e2);
    R a = dot(e1, q);
    if (rt_abs(a) < consts<R>::eps) return false;
    R f = R(1) / a;
    vec3<R> s = r.o - p0;
    u = f * dot(s, q);
    if (u < R(0)) return false;
    vec3<R> k = cross(s, e1);
    v = f * dot(r.d, k);
    if (v < R(0) || u + v > R(1)) return false;
    R tt = f * dot(e2, k);
    if (tt < t_min || tt > t_max) return false;
    t = tt;
    return true;
}

// box under rotatey / translate (primitive.hpp:399-458, transform.hpp:9-121): the ray goes into the box's frame, the six rects are
// tested in the constructor's order through group::intersect's shrinking t_max (group.hpp:68-83), and the record comes back
// out through both wrappers' set_face_normal calls.  h == nullptr: distance only.
template <typename R> struct Hit;
template <typename R, bool kSelf = true> __device__ inline bool box_t(const Prim<R>& p, const TriShade<R>& aux, const Ray<R>& r, R t_min, R t_max, R& t, Hit<R>* h);

// A ray that starts ON a sphere has the roots t = 0 and t = -2 (oc.d)/(d.d) exactly.  In fp64 the reference's t = 0
// root comes out ~1e-16 and is cut by t_min = 1e-3; in fp32 the origin is only within ~1e-7 of the surface, and at
// grazing angles c/q grows past t_min: the ray would re-hit its own sphere from the inside (measured: +3.6 % rays,
// paths trapped in the glass spheres).  So the fp32 modes take the non-zero root algebraically for the sphere the ray
// starts on, and skip planar primitives the ray starts on (their only root is the t = 0 one).  fp64 mode: untouched.
__device__ inline bool sphere_from_surface(const vec3<float>& center, const Ray<float>& r, float t_min, float t_max, float& t)
{
    vec3<float> oc = r.o - center;
    float a = dot(r.d, r.d), hb = dot(oc, r.d);
    float root = -2.0f * hb / a;
    if (!(root >= t_min && root <= t_max)) return false;
    t = root;
    return true;
}

// volume::intersect (primitive.hpp:346-385) around a sphere boundary: g[0..3] centre, radius, g[4] = -1/density.  The two crossings
// are the boundary's own intersect() over (-inf, inf) and over (t1 + 0.0001, inf); the free-flight distance is drawn INSIDE the
// test (:373), so the position in the path's stream depends on which volumes the walk reaches, in which order.
template <typename R> __device__ inline bool volume_crossings(const Prim<R>& p, const Ray<R>& r, R& t1, R& t2)
{
    const vec3<R> c(p.g[0], p.g[1], p.g[2]);
    if (!sphere_t(c, p.g[3], r, -rt_inf<R>(), rt_inf<R>(), t1)) return false;                    // :349
    return sphere_t(c, p.g[3], r, t1 + R(0.0001), rt_inf<R>(), t2);                              // :353
}
// fp32: t1 + 0.0001 is t1 itself once t1 > ~1000 (the radius-5000 haze sphere), so both roots are taken from one solve, with the
// cancelling terms in fp64 as in sphere_t<float>
template <> __device__ inline bool volume_crossings<float>(const Prim<float>& p, const Ray<float>& r, float& t1, float& t2)
{
    const double ox = (double) r.o.x - (double) p.g[0], oy = (double) r.o.y - (double) p.g[1], oz = (double) r.o.z - (double) p.g[2];
    const double dx = r.d.x, dy = r.d.y, dz = r.d.z;
    const double a = dx * dx + dy * dy + dz * dz;
    const double hb = ox * dx + oy * dy + oz * dz;
    const double c = ox * ox + oy * oy + oz * oz - (double) p.g[3] * (double) p.g[3];
    const double qd = hb * hb - a * c;
    if (!(qd >= 0.0) || !(a > 0.0)) return false;
    const float sq = sqrtf((float) qd), hbf = (float) hb, af = (float) a, cf = (float) c;
    const float q = hbf < 0.0f ? sq - hbf : -(hbf + sq);
    float r0 = q / af, r1 = q != 0.0f ? cf / q : r0;
    t1 = fminf(r0, r1); t2 = fmaxf(r0, r1);
    return t2 >= t1 + 0.0001f;
}
// ... around a box boundary (RT_PRIM_FLAG_BOUNDARY_BOX; the box optionally inside rotatey and translate, which box_t applies): the same
// two calls of the boundary's intersect(), group::intersect over the six rects each time (primitive.hpp:439-442)
template <typename R> __device__ inline bool volume_crossings_box(const Prim<R>& p, const TriShade<R>& aux, const Ray<R>& r, R& t1, R& t2)
{
    if (!box_t<R, false>(p, aux, r, -rt_inf<R>(), rt_inf<R>(), t1, nullptr)) return false;        // :349
    return box_t<R, false>(p, aux, r, t1 + R(0.0001), rt_inf<R>(), t2, nullptr);                  // :353
}
// fp32: entry and exit of the slabs in the box's frame from one pass (t1 + 0.0001 is t1 itself for distant boxes, as with the sphere)
template <> __device__ inline bool volume_crossings_box<float>(const Prim<float>& p, const TriShade<float>& aux, const Ray<float>& r, float& t1, float& t2)
{
    const int fl = p.flags();
    vec3<float> o = r.o, d = r.d;
    if (fl & RT_PRIM_FLAG_TRANSLATE) o = o - vec3<float>(aux.n[0], aux.n[1], aux.n[2]);
    if (fl & RT_PRIM_FLAG_ROTATEY) {
        const float sn = p.g[6], cs = p.g[7];
        const float ox = cs * o.x - sn * o.z, oz = sn * o.x + cs * o.z, dx = cs * d.x - sn * d.z, dz = sn * d.x + cs * d.z;
        o.x = ox; o.z = oz; d.x = dx; d.z = dz;
    }
    const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    float t_in = -rt_inf<float>(), t_out = rt_inf<float>();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = p.g[k], hi = p.g[3 + k];
        if (dd[k] == 0.0f) { if (oo[k] < lo || oo[k] > hi) return false; continue; }
        const float inv = 1.0f / dd[k], a = (lo - oo[k]) * inv, b = (hi - oo[k]) * inv;
        t_in = fmaxf(t_in, fminf(a, b)); t_out = fminf(t_out, fmaxf(a, b));
    }
    t1 = t_in; t2 = t_out;
    return t2 >= t1 + 0.0001f;
}
template <typename R> __device__ inline bool volume_t(const Prim<R>& p, Side<R> aux, const Ray<R>& r, R t_min, R t_max, R& t, Rng* g)
{
    R t1, t2;
    if (!g) return false;                                              // no stream (first-hit entry points): a volume is never hit
    const bool boxed = (p.flags() & RT_PRIM_FLAG_BOUNDARY_BOX) != 0;
    if (boxed ? !volume_crossings_box(p, aux[p.id], r, t1, t2) : !volume_crossings(p, r, t1, t2)) return false;
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return false;
    if (t1 < R(0)) t1 = R(0);
    const R ray_length = length(r.d);
    const R inner_length = (t2 - t1) * ray_length;
    const R neg_inv_d = boxed ? p.g[8] : p.g[4];
    const R intersect_distance = neg_inv_d * rt_log(uniform<R>(*g));
    if (intersect_distance > inner_length) return false;
    t = t1 + intersect_distance / ray_length;
    return true;
}

// ray into an instance's frame: translate::intersect (transform.hpp:18-20), then rotatey::intersect (:83-92)
template <typename R> __device__ inline void xform_rays(const Xform<R>& x, const Ray<R>& r, Ray<R>& tr, Ray<R>& rr)
{
    tr = r;
    if (x.flags & RT_PRIM_FLAG_TRANSLATE) tr.o = r.o - vec3<R>(x.t[0], x.t[1], x.t[2]);
    rr = tr;
    if (x.flags & RT_PRIM_FLAG_ROTATEY) {
        rr.o.x = x.cs * tr.o.x - x.sn * tr.o.z;
        rr.o.z = x.sn * tr.o.x + x.cs * tr.o.z;
        rr.d.x = x.cs * tr.d.x - x.sn * tr.d.z;
        rr.d.z = x.sn * tr.d.x + x.cs * tr.d.z;
    }
}

// ray into the frame at the inner end of the transform chain that starts at record xi (rtcore.h RT_XFORM_FLAG_CHAIN: wrappers that do not
// fit one translate(rotatey(.)) record, instances inside instances), outermost wrapper first as the reference's intersect() calls nest
template <typename R> __device__ inline void chain_ray(const Xform<R>* __restrict__ xs, int xi, const Ray<R>& r, Ray<R>& out)
{
    Ray<R> tr;
    xform_rays(xs[xi], r, tr, out);
    while (xs[xi].flags & RT_XFORM_FLAG_CHAIN) {
        xi++;
        const Ray<R> cur = out;
        xform_rays(xs[xi], cur, tr, out);
    }
}

// Scene classes the fast kernels are compiled for (kFull): 0 spheres, rects and triangles; 1 also boxes / volumes; 2 also instances;
// 3 an OBJ-loader scene (triangles only, every material a phong, no checker; 4: unused); 5, 6, 7: the classes 0, 1, 2 of scenes in the
// style of "ray tracing in one weekend / the next week" -- no triangle, no phong, no image texture, rect lights only (5 also without
// fog, which only a volume carries).  Every kind a variant could meet costs it registers whether the scene holds one or not.
constexpr bool class_is_obj(int kFull) { return kFull == 3 || kFull == 4; }
constexpr bool class_no_tri(int kFull) { return kFull == 5 || kFull == 6 || kFull == 7; }
constexpr bool class_boxes(int kFull) { return kFull == 1 || kFull == 2 || kFull == 6 || kFull == 7; }      // boxes and volumes
constexpr bool class_instances(int kFull) { return kFull == 2 || kFull == 7; }
constexpr int class_mat(int kFull) { return class_is_obj(kFull) ? 1 : (kFull == 5 ? 2 : (class_no_tri(kFull) ? 3 : 0)); }   // kMat of the mat_* functions
constexpr int class_tex(int kFull) { return class_no_tri(kFull) ? 2 : (class_is_obj(kFull) ? 1 : 0); }                                                 // kTex of tex_color
constexpr int class_light(int kFull) { return class_no_tri(kFull) ? 1 : 0; }                                               // kLight of the light functions

template <typename R, int kFull = 2> __device__ inline bool prim_t_local(const Prim<R>& p, Side<R> aux, const Ray<R>& r, R t_min, R t_max, R& t, Rng* g)
{
    R a, b;
    if constexpr (class_is_obj(kFull)) {                                 // a scene of triangles only (what the OBJ loader builds)
        if (sizeof(R) == sizeof(float) && p.id == r.from) return false;
        return tri_t(p, r, t_min, t_max, t, a, b);
    }
    if constexpr (sizeof(R) == sizeof(float)) {
        if (p.id == r.from && p.kind() != RT_PRIM_BOX && p.kind() != RT_PRIM_VOLUME) {   // (a box is tested again, without the side the ray starts on: see box_t;
                                                                                          //  a path scattered inside a medium is still inside it)
            if (p.kind() != RT_PRIM_SPHERE && p.kind() != RT_PRIM_MSPHERE) return false;
            return sphere_from_surface(p.kind() == RT_PRIM_SPHERE ? vec3<float>(p.g[0], p.g[1], p.g[2]) : msphere_center(p, r.time), r, t_min, t_max, t);
        }
    }
    switch (p.kind()) {
    case RT_PRIM_SPHERE:                // one body for both sphere kinds: a wave whose leaves hold both would otherwise run the solve twice
    case RT_PRIM_MSPHERE: {
        const bool moving = p.kind() == RT_PRIM_MSPHERE;
        return sphere_t(moving ? msphere_center(p, r.time) : vec3<R>(p.g[0], p.g[1], p.g[2]), moving ? p.g[8] : p.g[3], r, t_min, t_max, t);
    }
    case RT_PRIM_TRIANGLE: if constexpr (!class_no_tri(kFull)) return tri_t(p, r, t_min, t_max, t, a, b); else return false;
    case RT_PRIM_BOX: if constexpr (class_boxes(kFull)) return box_t<R>(p, aux[p.id], r, t_min, t_max, t, nullptr); else return false;
    case RT_PRIM_VOLUME: if constexpr (class_boxes(kFull)) return volume_t<R>(p, aux, r, t_min, t_max, t, g); else return false;
    default: return plane_t(p, r, t_min, t_max, t, a, b);
    }
}

// A leaf test of the fast modes passed with distance t: does the lane take it?  Nearer than the closest hit so far: yes.  At the SAME
// distance (two coincident surfaces): the reference keeps whichever its own walk visits last, whatever order this tree visits them in.
template <typename R> __device__ inline bool take_hit(const Scene<R>& sc, R t, R closest, int32_t cur_hit, int32_t new_id)
{
    if (t != closest || cur_hit < 0) return true;
    return sc.ref_rank[new_id] > sc.ref_rank[sc.fprims[cur_hit].id];
}

// the primitive test behind every leaf; members of an instance are tested in the instance's frame (t is the same in both)
// kFull: the scene class (class_* above).  Carrying the box, volume and instance code costs the sphere / triangle workloads 5-10 %, the
// instance check costs the Cornell box 15 %; the triangle / phong / image-texture code costs the sphere scenes 7-18 %.
template <typename R, int kFull = 2> __device__ inline bool prim_t(const Prim<R>& p, Side<R> aux, const Ray<R>& r, R t_min, R t_max, R& t, Rng* g = nullptr)
{
    const int xi = class_instances(kFull) ? p.xform() : -1;
    if (class_instances(kFull) && xi >= 0) {
        Ray<R> rr;
        chain_ray(aux.xforms, xi, r, rr);
        return prim_t_local<R, kFull>(p, aux, rr, t_min, t_max, t, g);
    }
    return prim_t_local<R, kFull>(p, aux, r, t_min, t_max, t, g);
}

// hit record (intersection.hpp:13-27)
template <typename R> struct Hit {
    vec3<R> pos, normal;
    R u = R(0), v = R(0), t = R(0);
    int32_t mat = 0;
    int32_t id = -1;   // primitive id of the hit
    bool front = false;
};

template <typename R> __device__ inline void set_face_normal(Hit<R>& h, const Ray<R>& r, const vec3<R>& n)  // intersection.hpp:22-26
{
    h.front = dot(r.d, n) < R(0);
    h.normal = h.front ? n : -n;
}

// Completes the record of a hit already known to lie at t on primitive p: the statements after the range test in
// each intersect() (primitive.hpp:49-55, 123-129, 189-198; polygon.hpp:117-123; flip transform.hpp:134).
template <typename R, int kFull = 2> __device__ inline void fill_record_local(const Prim<R>& p, Side<R> tris, const Ray<R>& r, R t,
                                                                                 bool need_uv, Hit<R>& h)
{
    h.t = t;
    h.mat = p.mat;
    h.id = p.id;
    h.u = h.v = R(0);
    switch (class_is_obj(kFull) ? RT_PRIM_TRIANGLE : p.kind()) {
    case RT_PRIM_SPHERE:
    case RT_PRIM_MSPHERE: {
        vec3<R> c = p.kind() == RT_PRIM_SPHERE ? vec3<R>(p.g[0], p.g[1], p.g[2]) : msphere_center(p, r.time);
        R radius = p.kind() == RT_PRIM_SPHERE ? p.g[3] : p.g[8];
        h.pos = r.o + t * r.d;
        vec3<R> n = (h.pos - c) / radius;
        set_face_normal(h, r, n);
        if (!class_no_tri(kFull) && need_uv) {                              // primitive.hpp:67-74 (only an image texture reads uv)
            R theta = rt_acos(-n.y);
            R phi = rt_atan2(-n.z, n.x) + consts<R>::pi;
            h.u = phi / (R(2) * consts<R>::pi);
            h.v = theta / consts<R>::pi;
        }
        break;
    }
    case RT_PRIM_TRIANGLE: if constexpr (!class_no_tri(kFull)) {
        R tt, u, v;
        tri_t(p, r, -rt_inf<R>(), rt_inf<R>(), tt, u, v);                   // same arithmetic as the accepted test
        h.pos = r.o + t * r.d;
        const TriShade<R>& s = tris[p.id];
        R w = R(1) - u - v;
        h.u = (u * s.uv[2] + v * s.uv[4]) + w * s.uv[0];                    // polygon.hpp:174-177
        h.v = (u * s.uv[3] + v * s.uv[5]) + w * s.uv[1];
        vec3<R> n0(s.n[0], s.n[1], s.n[2]), n1(s.n[3], s.n[4], s.n[5]), n2(s.n[6], s.n[7], s.n[8]);
        vec3<R> n = (u * n1 + v * n2) + w * n0;                             // polygon.hpp:179-182 (not re-normalised)
        set_face_normal(h, r, n);
        break;
    } else break;
    case RT_PRIM_BOX: {
        if constexpr (class_boxes(kFull)) {
            R tt;
            box_t<R>(p, tris[p.id], r, consts<R>::tmin, rt_inf<R>(), tt, &h);   // the closest side again: same arithmetic as the accepted test
        }
        break;
    }
    case RT_PRIM_VOLUME:                                                    // primitive.hpp:379-383 (uv_coord is left as it was: 0 here)
        h.pos = r.o + t * r.d;
        h.normal = vec3<R>(R(1), R(0), R(0));
        h.front = false;
        break;
    default: {
        R oa, ob, ok, da, db, dk;
        plane_split(p.kind(), r.o, oa, ob, ok);
        plane_split(p.kind(), r.d, da, db, dk);
        R a = oa + t * da;
        R b = ob + t * db;
        h.u = (a - p.g[0]) / (p.g[1] - p.g[0]);
        h.v = (b - p.g[2]) / (p.g[3] - p.g[2]);
        vec3<R> n(p.kind() == RT_PRIM_PLANEYZ ? R(1) : R(0), p.kind() == RT_PRIM_PLANEXZ ? R(1) : R(0), p.kind() == RT_PRIM_PLANEXY ? R(1) : R(0));
        set_face_normal(h, r, n);
        h.pos = r.o + t * r.d;
        break;
    }
    }
    if (p.flags() & RT_PRIM_FLAG_FLIP) h.front = !h.front;
}

template <typename R, int kFull = 2> __device__ inline void fill_record(const Prim<R>& p, Side<R> tris, const Ray<R>& r, R t, bool need_uv, Hit<R>& h)
{
    const int xi = class_instances(kFull) ? p.xform() : -1;
    if (!class_instances(kFull) || xi < 0) { fill_record_local<R, kFull>(p, tris, r, t, need_uv, h); return; }
    // member of an instance: the record is made in the innermost frame and comes back out through every wrapper's set_face_normal, innermost
    // first -- each with the ray as THAT wrapper saw it (transform.hpp:25-27, 99-110), so the rays are rebuilt from the outside for every level
    // (chains are one record long for translate(rotatey(.)); longer ones are rare and short)
    const Xform<R>* __restrict__ xs = tris.xforms;
    int n = 1;
    while (xs[xi + n - 1].flags & RT_XFORM_FLAG_CHAIN) n++;
    for (int level = n - 1; level >= 0; level--) {
        Ray<R> cur = r, tr, rr;
        for (int k = 0; k < level; k++) { Ray<R> a; const Ray<R> c = cur; xform_rays(xs[xi + k], c, a, cur); }
        xform_rays(xs[xi + level], cur, tr, rr);
        if (level == n - 1) fill_record_local<R, kFull>(p, tris, rr, t, need_uv, h);
        const Xform<R>& x = xs[xi + level];
        if (x.flags & RT_PRIM_FLAG_ROTATEY) {                                   // transform.hpp:99-110
            vec3<R> position = h.pos, normal = h.normal;
            position.x = x.cs * h.pos.x + x.sn * h.pos.z;
            position.z = -x.sn * h.pos.x + x.cs * h.pos.z;
            normal.x = x.cs * h.normal.x + x.sn * h.normal.z;
            normal.z = -x.sn * h.normal.x + x.cs * h.normal.z;
            h.pos = position;
            set_face_normal(h, rr, normal);
        }
        if (x.flags & RT_PRIM_FLAG_TRANSLATE) {                                 // transform.hpp:25-27
            h.pos = h.pos + vec3<R>(x.t[0], x.t[1], x.t[2]);
            set_face_normal(h, tr, h.normal);
        }
    }
    if (p.flags() & RT_PRIM_FLAG_FLIP_OUTER) h.front = !h.front;                // a flip outside the whole chain (transform.hpp:134)
}

template <typename R, bool kSelf> __device__ inline bool box_t(const Prim<R>& p, const TriShade<R>& aux, const Ray<R>& r, R t_min, R t_max, R& t, Hit<R>* h)
{
    const int fl = p.flags();
    const vec3<R> target(aux.n[0], aux.n[1], aux.n[2]);
    const R sn = p.g[6], cs = p.g[7];
    Ray<R> tr = r;                                                       // translate::intersect, transform.hpp:18-28
    if (fl & RT_PRIM_FLAG_TRANSLATE) tr.o = r.o - target;
    Ray<R> rr = tr;                                                      // rotatey::intersect, transform.hpp:81-110
    if (fl & RT_PRIM_FLAG_ROTATEY) {
        rr.o.x = cs * tr.o.x - sn * tr.o.z;
        rr.o.z = sn * tr.o.x + cs * tr.o.z;
        rr.d.x = cs * tr.d.x - sn * tr.d.z;
        rr.d.z = sn * tr.d.x + cs * tr.d.z;
    }
    // the six rects of the box constructor (primitive.hpp:409-431), in its order: xy at max z, min z; xz at max y, min y; yz at max x, min x
    auto side = [&](int i, Prim<R>& s) {
        const int pair = i >> 1;
        s.kf = pair == 0 ? RT_PRIM_PLANEXY : (pair == 1 ? RT_PRIM_PLANEXZ : RT_PRIM_PLANEYZ);
        s.g[0] = pair == 2 ? p.g[1] : p.g[0]; s.g[1] = pair == 2 ? p.g[4] : p.g[3];
        s.g[2] = pair == 0 ? p.g[1] : p.g[2]; s.g[3] = pair == 0 ? p.g[4] : p.g[5];
        const R hi = pair == 0 ? p.g[5] : (pair == 1 ? p.g[4] : p.g[3]), lo = pair == 0 ? p.g[2] : (pair == 1 ? p.g[1] : p.g[0]);
        s.g[4] = (i & 1) ? lo : hi;
    };
    // fp32: a ray that starts on this box must not find the side it starts on (its t = 0 root drifts past t_min at grazing
    // angles), but it must find the others: the wrappers' set_face_normal calls (object-space ray against the world-space
    // normal) can turn the shading normal into the box, and the reference then traces the path inside it.
    const bool self = kSelf && sizeof(R) == sizeof(float) && p.id == r.from;     // (kSelf false: the box is a volume's boundary)
    R closest = t_max;
    int hit_side = -1;
    Prim<R> s;
#ifndef RT_BOX_LOOP
    if constexpr (sizeof(R) == sizeof(float)) {
        // fp32: the closest of the six rects is the slab entry, or, from inside, the slab exit -- three reciprocals instead of six
        // rect tests (the 400 ground boxes of all_feature_test() made this the hottest leaf code of that scene)
        const R o[3] = { rr.o.x, rr.o.y, rr.o.z }, d[3] = { rr.d.x, rr.d.y, rr.d.z };
        R t_in = -rt_inf<R>(), t_out = rt_inf<R>();
        int side_in = -1, side_out = -1;
        bool start_in = false, start_out = false;                        // the ray starts on a side and heads into / out of the box
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const R lo = p.g[k], hi = p.g[3 + k], inv = R(1) / d[k];
            const R t_lo = (lo - o[k]) * inv, t_hi = (hi - o[k]) * inv;
            const bool up = d[k] >= R(0);                                // enters through lo, leaves through hi
            R tn = up ? t_lo : t_hi, tf = up ? t_hi : t_lo;
            if (d[k] == R(0)) { const bool in = o[k] >= lo && o[k] <= hi; tn = in ? -rt_inf<R>() : rt_inf<R>(); tf = in ? rt_inf<R>() : -rt_inf<R>(); }
            if (self) {
                const R scale_lo = rt_abs(o[k]) > rt_abs(lo) ? rt_abs(o[k]) : rt_abs(lo), scale_hi = rt_abs(o[k]) > rt_abs(hi) ? rt_abs(o[k]) : rt_abs(hi);
                const bool on_lo = rt_abs(lo - o[k]) <= R(2e-6) * (scale_lo > R(1) ? scale_lo : R(1));
                const bool on_hi = rt_abs(hi - o[k]) <= R(2e-6) * (scale_hi > R(1) ? scale_hi : R(1));
                if ((on_lo && up) || (on_hi && !up)) { start_in = true; tn = -rt_inf<R>(); }      // the side it starts on is not a candidate
                if ((on_lo && !up) || (on_hi && up)) start_out = true;
            }
            const int pair = 2 - k;                                      // rect pairs in constructor order: z, y, x; even index = the max side
            if (tn > t_in) { t_in = tn; side_in = pair * 2 + (up ? 1 : 0); }
            if (tf < t_out) { t_out = tf; side_out = pair * 2 + (up ? 0 : 1); }
        }
        if (start_out && !start_in) return false;                        // leaves the box at once: the other rects lie behind their bounds
        if (!(t_in <= t_out)) return false;
        if (t_in >= t_min && t_in <= closest && side_in >= 0) { closest = t_in; hit_side = side_in; }
        else if (t_out >= t_min && t_out <= closest) { closest = t_out; hit_side = side_out; }
        if (hit_side < 0) return false;
    }
    else
#endif
    {
#pragma unroll 1
        for (int i = 0; i < 6; i++) {                                    // group::intersect, group.hpp:68-83
            side(i, s);
            R tt, a, b;
            if (plane_t(s, rr, t_min, closest, tt, a, b)) { closest = tt; hit_side = i; }
        }
        if (hit_side < 0) return false;
    }
    t = closest;
    if (h) {
        side(hit_side, s);
        R oa, ob, ok, da, db, dk;
        plane_split(s.kind(), rr.o, oa, ob, ok);
        plane_split(s.kind(), rr.d, da, db, dk);
        h->t = t;
        h->u = ((oa + t * da) - s.g[0]) / (s.g[1] - s.g[0]);
        h->v = ((ob + t * db) - s.g[2]) / (s.g[3] - s.g[2]);
        const int pair = hit_side >> 1;
        set_face_normal(*h, rr, vec3<R>(pair == 2 ? R(1) : R(0), pair == 1 ? R(1) : R(0), pair == 0 ? R(1) : R(0)));
        h->pos = rr.o + t * rr.d;
        if (fl & RT_PRIM_FLAG_ROTATEY) {                                 // transform.hpp:99-110
            vec3<R> position = h->pos, normal = h->normal;
            position.x = cs * h->pos.x + sn * h->pos.z;
            position.z = -sn * h->pos.x + cs * h->pos.z;
            normal.x = cs * h->normal.x + sn * h->normal.z;
            normal.z = -sn * h->normal.x + cs * h->normal.z;
            h->pos = position;
            set_face_normal(*h, rr, normal);
        }
        if (fl & RT_PRIM_FLAG_TRANSLATE) {                               // transform.hpp:25-27
            h->pos = h->pos + target;
            set_face_normal(*h, tr, h->normal);
        }
    }
    return true;
}

struct VisitStats { uint32_t n_box, n_leaf; uint64_t hash; uint32_t w_node, w_leaf, w_bounce; uint32_t n_zero = 0; };   // n_zero: rays traced with an all-zero direction (phong's unset wo)   // w_*: wave-level loop iterations (counted by the first active lane)
__device__ inline bool first_active_lane() { return (int) __lane_id() == __ffsll((unsigned long long) __ballot(1)) - 1; }

// scene.objects.intersect(r, rec) (group.hpp:68-83 -> bvh.hpp:58-68) as a walk over the pre-order program.
template <typename R, int kStats>   // 0: none, 1: count box/leaf tests, 2: counts + visit-order hash
__device__ inline int closest_hit(const Op<R>* __restrict__ ops, int n_ops, const Prim<R>* __restrict__ prims, Side<R> aux,
                                  const Ray<R>& r, R& t_hit, VisitStats* st, Rng* g = nullptr)
{
    R closest = rt_inf<R>();
    const R t_min = consts<R>::tmin;
    int hit = -1;
    int i = 0;
    while (i < n_ops) {
        const Op<R> op = ops[i];
        if (kStats == 2) st->hash = (st->hash ^ (uint64_t) (uint32_t) i) * 0x100000001B3ULL;
        if (op.prim < 0) {                                                  // -1: world space; -2 - x: a node of instance x, box in its frame
            if (kStats) st->n_box++;
            bool inside;
            if (op.prim == -1) inside = box_hit(op, r, t_min, closest);
            else { Ray<R> rr; chain_ray(aux.xforms, -2 - op.prim, r, rr); inside = box_hit(op, rr, t_min, closest); }
            i = inside ? i + 1 : op.skip;
        }
        else {
            if (kStats) st->n_leaf++;
            R t;
            if (prim_t(prims[op.prim], aux, r, t_min, closest, t, g)) { closest = t; hit = op.prim; }
            i++;
        }
    }
    t_hit = closest;
    return hit;
}


// ------------------------------------------------------------------ fast mode: ordered stack traversal of the library's own SAH tree
// Same closest hit as the reference's walk (the closest hit does not depend on visit order), far fewer tests.
// "while-while" form: every lane first descends inner nodes until it holds a leaf, then the wave tests leaves
// together, so box-test and primitive-test instructions are not interleaved per lane.  The per-lane stack of
// postponed far children lives in LDS, one dword per level, lane-interleaved (entry `level*256 + tid`) so a
// wave's pushes/pops hit 64 consecutive banks.
__device__ inline void slab2(const float* b, const vec3<float>& o, const vec3<float>& idir, float t_min, float t_max, float& t_near, bool& hit)
{
    float x0 = (b[0] - o.x) * idir.x, x1 = (b[3] - o.x) * idir.x;
    float y0 = (b[1] - o.y) * idir.y, y1 = (b[4] - o.y) * idir.y;
    float z0 = (b[2] - o.z) * idir.z, z1 = (b[5] - o.z) * idir.z;
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_min));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t_max));
    t_near = tn;
    hit = tn <= tf * 1.0000004f;     // two-ulp slack keeps the culling conservative under fp32 rounding
}

template <int kStats>
__device__ inline int closest_hit_bvh(const BNode* __restrict__ nodes, const Prim<float>* __restrict__ prims, Side<float> aux,
                                      const Ray<float>& r, float& t_hit, uint32_t* stack, VisitStats* st, Rng* g = nullptr)
{
    const float t_min = consts<float>::tmin;
    float closest = rt_inf<float>();
    int hit = -1;
    // a zero direction component would make 0 * inf = NaN in the slab test: nudge it (box tests only)
    auto safe = [](float d) { return fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d; };
    const vec3<float> idir(1.0f / safe(r.d.x), 1.0f / safe(r.d.y), 1.0f / safe(r.d.z));
    int sp = 0;
    stack[0] = (uint32_t) RT_BVH_SENTINEL;
    int32_t cur = 0;
    while (cur != RT_BVH_SENTINEL) {
        while (cur >= 0) {
            const BNode n = nodes[cur];
            if (kStats && first_active_lane()) st->w_node++;
            float tl, tr; bool hl, hr;
            slab2(n.lb, r.o, idir, t_min, closest, tl, hl);
            slab2(n.rb, r.o, idir, t_min, closest, tr, hr);
            if (kStats) st->n_box += 2;
            if (hl && hr) {
                bool left_first = tl <= tr;
                int32_t far = left_first ? n.right : n.left;
                cur = left_first ? n.left : n.right;
                sp++;
                stack[sp * 256] = (uint32_t) far;
            }
            else if (hl) cur = n.left;
            else if (hr) cur = n.right;
            else { cur = (int32_t) stack[sp * 256]; sp--; }
        }
        if (cur == RT_BVH_SENTINEL) break;
        const int code = ~cur, first = code >> 3, count = (code & 7) + 1;
        for (int k = 0; k < count; k++) {
            float t;
            if (kStats && first_active_lane()) st->w_leaf++;
            if (kStats) st->n_leaf++;
            if (prim_t(prims[first + k], aux, r, t_min, closest, t, g)) { closest = t; hit = first + k; }
        }
        cur = (int32_t) stack[sp * 256]; sp--;
    }
    t_hit = closest;
    return hit;
}

// ------------------------------------------------------------------ textures and materials
// checker::value (texture.hpp:65-74) takes `odd` where sin(10 x) sin(10 y) sin(10 z) < 0: only the SIGN of the product is used.
// fp64: three library sines were ~400 of a shade pass's vector instructions on every scene with the checkered ground.  The sign of sin(t)
// is the parity of floor(t / pi): k = rint(|t| / pi), then the sign of r = |t| - k pi decides between the intervals k - 1 and k.  r is
// taken against pi = P1 + P2 (106 bits): k P1 = p + e exactly (fma), |t| - p is exact where it matters (Sterbenz: near a multiple of pi),
// and what is left out (k times the bits of pi below P2, the roundings of k P2 and of the last sum: < 1e-23 for |t| < 2^30) is far below the
// 4.7e-19 no double comes closer to a multiple of pi / 2 than (the bound behind fdlibm's argument reduction) -- so the computed sign IS
// the sign of the true sine, which is what any libm's sin(t) carries (its error is < 1 ulp of a value that is never that close to 0).
// Zero (t = +-0: sin = +-0, product not < 0), NaN / infinity (sin = NaN, `NaN < 0` false) and |t| >= 2^30 go the library's way.
// sign of sin(t) for 0 < |t| < 2^30, branch-free: true where sin(t) < 0
__device__ inline bool sine_negative(double t)
{
    const double P1 = 0x1.921fb54442d18p+1, P2 = 0x1.1a62633145c07p-53, INV_PI = 0x1.45f306dc9c883p-2;
    const double a = fabs(t);
    const double k = __builtin_rint(a * INV_PI);
    const double p = k * P1, e = __builtin_fma(k, P1, -p);
    const double r = __builtin_fma(-k, P2, (a - p) - e);
    const bool odd = ((int) k & 1) != 0;
    return (r >= 0.0 ? odd : !odd) != (t < 0.0);
}
__device__ inline int sine_sign(double t)                     // -1 / 0 / +1: sin(t) negative / zero / positive; 2: not a number (test hook, libm_kernel)
{
    if (!(fabs(t) < 1073741824.0)) { const double s = rt_sin(t); return s < 0.0 ? -1 : s > 0.0 ? 1 : s == 0.0 ? 0 : 2; }
    return t == 0.0 ? 0 : sine_negative(t) ? -1 : 1;
}
// the cold side of checker_odd, out of line: inlined under its branch, three copies of the library's sine cost every kernel variant
// scalar-register spills around the block (one copy in a loop: vector spills instead)
__device__ __attribute__((noinline)) inline bool checker_odd_library(double tx, double ty, double tz) { return rt_sin(tx) * rt_sin(ty) * rt_sin(tz) < 0.0; }
template <typename R> __device__ inline bool checker_odd(const vec3<R>& pos)
{
    const R tx = R(10) * pos.x, ty = R(10) * pos.y, tz = R(10) * pos.z;                                           // texture.hpp:67
    // one branch for all three: the library's sines where an argument is huge or not a number (never, in a scene of ordinary size)
    if (!(rt_abs(tx) < R(1073741824.0) && rt_abs(ty) < R(1073741824.0) && rt_abs(tz) < R(1073741824.0)))          // (a NaN compares false)
        return checker_odd_library(tx, ty, tz);
    const bool zero = tx == R(0) || ty == R(0) || tz == R(0);                                                     // a factor +-0: the product is not < 0
    return !zero && (sine_negative(tx) != sine_negative(ty)) != sine_negative(tz);
}
template <typename R> __device__ inline R checker_sine(const vec3<R>& pos);
// fp32: only the sign is used.  The hardware sine takes revolutions and is exact enough after taking the fraction
// (v_fract_f32 + v_sin_f32 instead of three range-reduced sinf calls, ~300 instructions).
template <> __device__ inline float checker_sine<float>(const vec3<float>& pos)
{
    const float k = 10.0f / (2.0f * consts<float>::pi);
    auto s = [](float rev) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev)); };
    return s(k * pos.x) * s(k * pos.y) * s(k * pos.z);
}
template <> __device__ inline bool checker_odd<float>(const vec3<float>& pos) { return checker_sine<float>(pos) < 0.0f; }
// image_texture::value (texture.hpp:95-124).  `pixels`: the texel bytes; the 256 entries of R in front of them hold pow(b / 255, 2.2) for
// every byte b, computed by rt_scene_create with the host's pow (rtcore.hip build_device): a colour texel is three table reads
template <typename R> __device__ inline vec3<R> tex_image(const uint8_t* __restrict__ pixels, const Tex<R>& t, R u, R v)
{
    if (t.a <= 0) return vec3<R>(R(0), R(1), R(1));
    R uu = u < R(0) ? R(0) : (u > R(1) ? R(1) : u);
    R vv = R(1) - (v < R(0) ? R(0) : (v > R(1) ? R(1) : v));
    int x = (int) (uu * t.a), y = (int) (vv * t.b);
    if (x >= t.a) x = t.a - 1;
    if (y >= t.b) y = t.b - 1;
    const uint8_t* b = pixels + t.offset + ((size_t) y * t.a * t.ch + (size_t) x * t.ch);
    if (t.color_based) {
        const R* __restrict__ decoded = reinterpret_cast<const R*>(pixels) - 256;
        return vec3<R>(decoded[b[0]], decoded[b[1]], decoded[b[2]]);
    }
    const R scale = R(1) / R(255);
    return scale * vec3<R>(R(b[0]), R(b[1]), R(b[2]));
}
// kTex = 2: the scene holds no image texture; 1: it holds no checker (OBJ-loader scenes: rtcore.hip scene_kinds)
template <typename R, int kTex = 0> __device__ inline vec3<R> tex_color(const Scene<R>& sc, int ti, const vec3<R>& pos, R u, R v)
{
    for (int level = 0; level < 8; level++) {
        const Tex<R>& t = sc.texs[ti];
        if (t.kind == RT_TEX_CONSTANT) return vec3<R>(t.rgb[0], t.rgb[1], t.rgb[2]);        // texture.hpp:32-35
        if (kTex != 1 && t.kind == RT_TEX_CHECKER) {                                         // texture.hpp:65-74
            ti = checker_odd(pos) ? t.a : t.b;
            continue;
        }
        if (kTex != 2 && t.kind == RT_TEX_IMAGE) return tex_image<R>(sc.pixels, t, u, v);
        break;
    }
    return vec3<R>();
}

template <typename R> __device__ inline bool nz3(const R* k) { return k[0] > R(0) || k[1] != R(0) || k[2] != R(0); }   // material.hpp:298-302

template <typename R> __device__ inline R reflectance(R cos_theta, R ior)                                              // material.hpp:128-133
{
    R r = (R(1) - ior) / (R(1) + ior);
    r = r * r;
    return r + (R(1) - r) * rt_pow(R(1) - cos_theta, R(5));
}

// fp64: (1 - cos)^5 without the library's pow() (~200 vector instructions on every shade pass in which ONE lane stands on glass).  x^5
// is taken in double-double -- x^2 = p1 + e1 exactly (fma), x^4 and x^5 carry their rounding errors along -- and rounded once at the
// end, so the result is the correctly rounded x^5 (but where the exact value lies within 2^-100 of a rounding boundary): what a correctly
// rounded pow(x, 5) returns, and what glibc's returns for all but a sliver of arguments (tests: equal to the host's pow on >= 99.9 % of
// random arguments, never more than 1 ulp away)
__device__ inline double pow5(double x)
{
    const double p1 = x * x, e1 = __builtin_fma(x, x, -p1);
    const double p2 = p1 * p1, e2 = __builtin_fma(p1, p1, -p2) + 2.0 * (p1 * e1);
    const double p3 = p2 * x, e3 = __builtin_fma(p2, x, -p3) + e2 * x;
    return p3 - p3 == 0.0 ? p3 + e3 : p3;              // p3 not finite (overflow, x = inf, NaN): the error terms are inf - inf; p3 is pow()'s answer
}
template <> __device__ inline double reflectance<double>(double cos_theta, double ior)
{
    double r = (1.0 - ior) / (1.0 + ior);
    r = r * r;
    return r + (1.0 - r) * pow5(1.0 - cos_theta);
}

// fp32: (1 - cos)^5 by multiplication instead of powf (~150 instructions)
template <> __device__ inline float reflectance<float>(float cos_theta, float ior)
{
    float r = (1.0f - ior) / (1.0f + ior);
    r = r * r;
    const float x = 1.0f - cos_theta, x2 = x * x;
    return r + (1.0f - r) * (x2 * x2 * x);
}

// `next`: normalize(wi.d), taken once for the metal and the dielectric lanes of a shade pass (mat_scatter)
template <typename R> __device__ inline vec3<R> glass_dir(Rng& g, const vec3<R>& next, const Hit<R>& h, R ior)         // material.hpp:109-122
{
    R ratio = h.front ? (R(1) / ior) : ior;
    R cos_theta = rt_min(dot(-next, h.normal), R(1));
    R sin_theta = rt_sqrt(R(1) - cos_theta * cos_theta);
    bool not_refract = ratio * sin_theta > R(1);
    if (not_refract || reflectance(cos_theta, ratio) > uniform<R>(g)) return reflect(next, h.normal);
    return refract(next, h.normal, ratio);
}

template <typename R> __device__ inline vec3<R> cosine_sample(Rng& g, const vec3<R>& normal)                           // material.hpp:43-52
{
    vec3<R> rc = random_cosine<R>(g);
    vec3<R> a2 = normalize(normal);
    vec3<R> x = (rt_abs(a2.x) > R(0.9)) ? vec3<R>(R(0), R(1), R(0)) : vec3<R>(R(1), R(0), R(0));
    vec3<R> a1 = normalize(cross(a2, x));
    vec3<R> a0 = cross(a2, a1);
    return (rc.x * a0 + rc.y * a1) + rc.z * a2;
}

// `texc`: the material's texture evaluated at the hit (tex_color draws no random numbers, so it is evaluated once, before the
// per-material branches, for every lane whose material has one: one copy of the texture code, executed by all of them together)
// kMat = 1: the caller knows every material of the scene is a phong (what the OBJ loader builds, loader.hpp:160); 2: none is a phong or a fog; 3: none is a phong
template <typename R, int kMat = 0> __device__ inline vec3<R> mat_emit(const Mat<R>& m, const Hit<R>& h, const vec3<R>& texc)
{
    if (kMat != 1 && m.kind == RT_MAT_EMISSIVE) return h.front ? texc : vec3<R>();                                                   // material.hpp:146-154
    if (kMat == 1 || (kMat != 2 && kMat != 3 && m.kind == RT_MAT_PHONG)) return (nz3(m.d + 7) && h.front) ? vec3<R>(m.d[7], m.d[8], m.d[9]) : vec3<R>();        // material.hpp:201-209
    return vec3<R>();
}

// material::shading.  wo keeps its default (origin 0, direction 0, time 0: `ray wo;` raytracing.cpp:41) when the
// material does not set it (phong's specular and diffuse branches): `dir_set` stays false.
// The per-material branches produce only the direction; the scattered ray is assembled once, after they have
// rejoined (mat_shading below).  Assigning the whole ray inside every branch made the compiler's output fragile:
// with some flag combinations lanes on the dielectric branch came out with o.z and time of the scattered ray holding
// temporaries of the other branches (found by comparing per-pixel ray counts with the fp64 mode; guarded by
// tests/test_gpu_parity.py::test_fast_ray_counts_match_fp64_per_pixel).
// `fixed` (RT_FLAG_COMPAT_FIXED, no counterpart in the reference): phong sets its direction (mirror / cosine lobe), and a specular
// direction below the surface is absorbed (returns false) instead of continuing with a pdf of 0.
template <typename R, int kMat = 0> __device__ inline bool mat_scatter(Rng& g, const Mat<R>& m, const Ray<R>& wi, const Hit<R>& h, const vec3<R>& texc,
                                                         vec3<R>& dir, bool& dir_set, vec3<R>& att, bool& spec, bool fixed = false)
{
    dir_set = true;
    // the incoming direction's unit vector, once for the metal and the dielectric lanes of the pass (a wave runs both branches whenever it
    // holds both kinds: one root and one division instead of two of each)
    vec3<R> unit_in;
    if (kMat != 1 && (m.kind == RT_MAT_METAL || m.kind == RT_MAT_DIELECTRIC)) unit_in = normalize(wi.d);
    switch (kMat == 1 ? RT_MAT_PHONG : m.kind) {
    case RT_MAT_LAMBERTIAN:                                                                 // material.hpp:29-35
        spec = false;
        dir = cosine_sample(g, h.normal);
        att = texc;
        return true;
    case RT_MAT_METAL: {                                                                    // material.hpp:74-82
        vec3<R> next = reflect(unit_in, h.normal);
        vec3<R> fuzz = m.d[0] * random_sphere<R>(g);
        dir = next + fuzz;
        att = texc;
        spec = true;
        if (fixed && !(dot(dir, h.normal) > R(0))) return false;
        return true;
    }
    case RT_MAT_DIELECTRIC:                                                                 // material.hpp:105-125
        spec = true;
        att = vec3<R>(R(1), R(1), R(1));
        dir = glass_dir(g, unit_in, h, m.d[0]);
        return true;
    case RT_MAT_FOG:                                                                        // material.hpp:174-180; pdf() stays the base class's 0 (:14)
        if constexpr (kMat == 2) return false;
        spec = false;
        dir = random_sphere<R>(g);
        att = texc;
        return true;
    case RT_MAT_PHONG: if constexpr (kMat == 2 || kMat == 3) return false; else {            // material.hpp:211-255
        if (nz3(m.d + 7)) return false;
        vec3<R> base = (m.tex >= 0) ? texc : vec3<R>(m.d[1], m.d[2], m.d[3]);
        if (nz3(m.d + 4) && uniform<R>(g) < R(0.5)) {
            att = vec3<R>(m.d[4], m.d[5], m.d[6]);
            spec = true;
            dir_set = fixed;
            if (fixed) { dir = reflect(normalize(wi.d), h.normal); if (!(dot(dir, h.normal) > R(0))) return false; }
            return true;
        }
        if (nz3(m.d + 10)) {
            spec = true;
            att = vec3<R>(m.d[10], m.d[11], m.d[12]);
            dir = glass_dir(g, normalize(wi.d), h, m.d[14]);
            return true;
        }
        spec = false;
        att = base;
        dir_set = fixed;
        if (fixed) dir = cosine_sample(g, h.normal);
        return true;
    }
    default: return false;                                                                  // emissive material.hpp:156-159
    }
}

#ifdef RT_OLD_SCATTER   // reproducer only (never built by the Makefile): the pre-fix shape, whole ray assigned inside every branch.  With it,
                        // `tools/ray_count_parity.py` shows 2.7 % / 11 % of the pixels off in the register-resident fp32 kernels (hipcc 7.2, gfx950),
                        // also with every Ray / Hit member initialised; optimisation barriers around the call or -ffp-contract=on hide it
template <typename R, int kMat = 0> __device__ inline bool mat_shading(Rng& g, const Mat<R>& m, const Ray<R>& wi, const Hit<R>& h, const vec3<R>& texc,
                                                         Ray<R>& wo, vec3<R>& att, bool& spec)
{
    switch (kMat == 1 ? RT_MAT_PHONG : m.kind) {
    case RT_MAT_LAMBERTIAN:
        spec = false;
        wo.o = h.pos; wo.from = h.id; wo.d = cosine_sample(g, h.normal); wo.time = wi.time;
        att = texc;
        return true;
    case RT_MAT_METAL: {
        vec3<R> next = reflect(normalize(wi.d), h.normal);
        vec3<R> fuzz = m.d[0] * random_sphere<R>(g);
        wo.o = h.pos; wo.from = h.id; wo.d = next + fuzz; wo.time = wi.time;
        att = texc;
        spec = true;
        return true;
    }
    case RT_MAT_DIELECTRIC:
        spec = true;
        att = vec3<R>(R(1), R(1), R(1));
        wo.o = h.pos; wo.from = h.id; wo.d = glass_dir(g, normalize(wi.d), h, m.d[0]); wo.time = wi.time;
        return true;
    case RT_MAT_PHONG: {
        if (nz3(m.d + 7)) return false;
        vec3<R> base = (m.tex >= 0) ? texc : vec3<R>(m.d[1], m.d[2], m.d[3]);
        if (nz3(m.d + 4) && uniform<R>(g) < R(0.5)) { att = vec3<R>(m.d[4], m.d[5], m.d[6]); spec = true; return true; }
        if (nz3(m.d + 10)) {
            spec = true;
            att = vec3<R>(m.d[10], m.d[11], m.d[12]);
            wo.o = h.pos; wo.from = h.id; wo.d = glass_dir(g, normalize(wi.d), h, m.d[14]); wo.time = wi.time;
            return true;
        }
        spec = false;
        att = base;
        return true;
    }
    default: return false;
    }
}
#else
template <typename R, int kMat = 0> __device__ inline bool mat_shading(Rng& g, const Mat<R>& m, const Ray<R>& wi, const Hit<R>& h, const vec3<R>& texc,
                                                                       Ray<R>& wo, vec3<R>& att, bool& spec, bool fixed = false)
{
    vec3<R> dir;
    bool dir_set = false;
    if (!mat_scatter<R, kMat>(g, m, wi, h, texc, dir, dir_set, att, spec, fixed)) return false;
    if (dir_set) { wo.o = h.pos; wo.from = h.id; wo.d = dir; wo.time = wi.time; }
    return true;
}
#endif

// normalize(a) for a caller that already holds length(a): the same operations (vec3.hpp: zero stays zero, else (1 / l) * a)
template <typename R> __device__ inline vec3<R> normalize_by(const vec3<R>& a, R l) { return l == R(0) ? vec3<R>() : a / l; }
template <> __device__ inline vec3<float> normalize_by<float>(const vec3<float>& a, float) { return normalize(a); }       // fp32: one v_rsq_f32

// wo_len = length(wo.d): the light pdf of the same direction needs it too (light_pdf), so the shade pass takes the root once
template <typename R, int kMat = 0> __device__ inline R mat_pdf(Rng& g, const Mat<R>& m, const Hit<R>& h, const Ray<R>& wo, R wo_len, bool fixed = false)
{
    switch (kMat == 1 ? RT_MAT_PHONG : m.kind) {
    case RT_MAT_LAMBERTIAN: {                                                               // material.hpp:37-41
        R c = dot(h.normal, normalize_by(wo.d, wo_len));
        return c > R(0) ? c / consts<R>::pi : R(0);
    }
    case RT_MAT_METAL: {                                                                    // material.hpp:84-88
        R c = dot(h.normal, normalize_by(wo.d, wo_len));
        return c > consts<R>::eps ? R(1) : R(0);
    }
    case RT_MAT_PHONG: if constexpr (kMat == 2 || kMat == 3) return R(0); else {             // material.hpp:257-266
        R c = dot(h.normal, normalize_by(wo.d, wo_len));
        if (!fixed && nz3(m.d + 4) && uniform<R>(g) < R(0.5)) return c > consts<R>::eps ? R(1) : R(0);   // fixed: the diffuse lobe's pdf, no second roll
        return c > R(0) ? c / consts<R>::pi : R(0);
    }
    default: return R(0);                                                                   // material.hpp:14
    }
}
template <typename R, int kMat = 0> __device__ inline R mat_pdf(Rng& g, const Mat<R>& m, const Hit<R>& h, const Ray<R>& wo, bool fixed = false)
{
    return mat_pdf<R, kMat>(g, m, h, wo, length(wo.d), fixed);
}

// ------------------------------------------------------------------ lights (group.hpp:104-119 over planexz / triangle)
// kLight = 1: every light of the scene is a rect (no triangle lights)
// len = length(value), taken once by the caller for all lights
template <typename R, int kLight = 0> __device__ inline R light_pdf(const Scene<R>& sc, int li, const vec3<R>& origin, const vec3<R>& value, R len)
{
    const Prim<R>& p = sc.lights[li];
    Ray<R> r; r.o = origin; r.d = value; r.time = R(0);                                     // ray(origin, value): time 0, t_min 1e-3, t_max inf
    R t, a, b;
    vec3<R> normal;
    R area;
    if (p.kind() == RT_PRIM_PLANEXZ) {                                                        // primitive.hpp:258-269
        if (!plane_t(p, r, consts<R>::tmin, rt_inf<R>(), t, a, b)) return R(0);
        vec3<R> n(R(0), R(1), R(0));
        normal = dot(r.d, n) < R(0) ? n : -n;
        area = (p.g[1] - p.g[0]) * (p.g[3] - p.g[2]);
    }
    else if (kLight != 1 && p.kind() == RT_PRIM_TRIANGLE) {                                   // polygon.hpp:140-151
        R u, v;
        if (!tri_t(p, r, consts<R>::tmin, rt_inf<R>(), t, u, v)) return R(0);
        const TriShade<R>& s = sc.light_tris[li];
        R w = R(1) - u - v;
        vec3<R> n0(s.n[0], s.n[1], s.n[2]), n1(s.n[3], s.n[4], s.n[5]), n2(s.n[6], s.n[7], s.n[8]);
        vec3<R> n = (u * n1 + v * n2) + w * n0;
        normal = dot(r.d, n) < R(0) ? n : -n;
        area = s.area;                                                                      // length(cross(e1, e2)) / 2, polygon.hpp:146
    }
    else return R(0);                                                                       // object.hpp:16
    R dist2 = t * t * len * len;
    R cosine = rt_abs(dot(value, normal) / len);
    return dist2 / (cosine * area);
}

template <typename R, int kLight = 0> __device__ inline vec3<R> light_rand(const Scene<R>& sc, Rng& g, const vec3<R>& origin)
{
    int idx = (int) uniform<R>(g, R(0), sc.n_lights_r);                                     // group.hpp:117, util.hpp:42-45
    if (idx >= sc.n_lights) idx = sc.n_lights - 1;                                          // fp32 only: guards u*n rounding up to n
    const Prim<R>& p = sc.lights[idx];
    if (p.kind() == RT_PRIM_PLANEXZ) {                                                        // primitive.hpp:270-274
        R z = uniform<R>(g, p.g[2], p.g[3]);
        R x = uniform<R>(g, p.g[0], p.g[1]);
        return vec3<R>(x, p.g[4], z) - origin;
    }
    if (kLight != 1 && p.kind() == RT_PRIM_TRIANGLE) {                                        // polygon.hpp:153-167
        vec3<R> p0(p.g[0], p.g[1], p.g[2]), e1(p.g[3], p.g[4], p.g[5]), e2(p.g[6], p.g[7], p.g[8]);
        R u = uniform<R>(g), v = uniform<R>(g);
        vec3<R> q = (u + v > R(1)) ? (p0 + (R(1) - u) * e1) + (R(1) - v) * e2 : (p0 + u * e1) + v * e2;
        return q - origin;
    }
    return vec3<R>(R(1), R(0), R(0));                                                       // object.hpp:17
}

template <typename R, int kLight = 0> __device__ inline R lights_pdf(const Scene<R>& sc, const vec3<R>& origin, const vec3<R>& value, R len)
{
    R weight = sc.inv_n_lights;                                                             // 1.0 / objects.size(), group.hpp:107
    R avg = R(0);
    for (int i = 0; i < sc.n_lights; i++) avg += weight * light_pdf<R, kLight>(sc, i, origin, value, len);
    return avg;
}
template <typename R, int kLight = 0> __device__ inline R lights_pdf(const Scene<R>& sc, const vec3<R>& origin, const vec3<R>& value)
{
    return lights_pdf<R, kLight>(sc, origin, value, length(value));
}

// ------------------------------------------------------------------ camera (camera.hpp:62-68)
template <typename R> __device__ inline Ray<R> camera_emit(const Camera<R>& c, Rng& g, R x, R y)
{
    vec3<R> defocus = c.lens_radius * random_disk<R>(g);
    vec3<R> offset = defocus.x * c.u + defocus.y * c.v;
    Ray<R> r;
    r.d = (((c.left_bottom + x * c.horizontal) + y * c.vertical) - c.pos) - offset;
    r.o = c.pos + offset;
    r.time = uniform<R>(g, c.s0, c.s1);
    return r;
}

// ------------------------------------------------------------------ launch parameters shared with the host side
struct RenderArgs {
    int32_t width, height, y_begin, y_end, spp_begin, spp_end, max_depth, flags;
    double wm1, hm1;                           // double(width - 1), double(height - 1): the divisors of raytracing.cpp:102-103, converted once on the host
    uint64_t seed;
    uint32_t lds_ops_bytes, lds_prims_bytes;   // 0: read the traversal program from global memory
    uint32_t bvh_stack_levels;                 // > 0: fast mode, LDS stack of this many dwords per lane
    uint32_t lds_tile_offset;                  // fp32 modes: byte offset of the tile accumulation buffer (768 floats + counter) in LDS
    // rt_wave.h phase thresholds, in 1/256ths: stay in the traversal phase while traversing lanes >= trav_keep/256 of the
    // lanes holding paths; stay in the node loop while lanes on inner nodes >= node_keep/256 of the traversing lanes;
    // regenerate when at least regen_min lanes have no path (or none has one)
    uint32_t trav_keep, node_keep, regen_min;
    uint32_t quantised;                        // fast mode walks Scene::wnodes (quantised, four-wide) instead of Scene::bnodes
    uint32_t kinds;                            // fast kernel variant: 0 spheres / rects / triangles only, 1 also boxes or volumes, 2 also instances, 3 triangles + phong only
    uint32_t xcd_tiles;                        // != 0: rt_wave.h deals 8 x 8-tile super-tiles to the 8 XCDs (grid padded to whole rounds of 8 super-tiles)
    // rt_queue.h (decoupled traversal / shading): slots per wave (0: the rt_wave.h kernel), pools allocated, lanes that must wait before a
    // hand-off is run, known hits from which a starved wave shades a partial pass
    uint32_t q_slots, q_pools, q_handoff_min, q_shade_starved;
    // rt_wave.h, cost-ordered launch: a scheduling unit is a super-tile (xcd_tiles) or a tile.  unit_order (or null): launch slot ->
    // unit, the units sorted by the rays they cost in an earlier render of the same frame geometry, dearest first, so that the
    // workgroups still running when the queue has emptied are the cheap ones (the launch's tail).  unit_cost (or null): this launch
    // adds every workgroup's ray count to its unit's entry
    const uint32_t* unit_order;
    uint32_t* unit_cost;
};

// ------------------------------------------------------------------ render kernel
// One lane owns one pixel and walks its samples [spp_begin, spp_end).  The bounce loop is flattened: every
// iteration each live lane extends its path by ONE segment; a lane whose path ended regenerates the next camera
// sample in place (path regeneration), so a wave only idles at the very end of its pixels' sample ranges instead
// of after every sample.  Waves are 8x8 pixel tiles; a 256-thread workgroup is a 16x16 tile.
//
// Iterative form of trace() (raytracing.cpp:30-70): L = sum_i beta_i * emission_i with
// beta_{i+1} = beta_i (*) (1/pdf_i) * (mat_pdf_i * attenuation_i).  The reference's recursion multiplies the same
// factors on the way back up, so a non-finite 1/pdf or weight anywhere poisons the whole sample, even when the
// path later misses (0 * inf); `poisoned` carries that.
// kTrav: 0 reference-order program in global memory, 1 the same staged in LDS, 2 fast-mode SAH tree (fp32 only)
template <typename R, int kTrav, bool kCount>
__global__ void __launch_bounds__(256, (sizeof(R) == 4 ? RT_WAVES_FP32 : 1)) render_kernel(Scene<R> sc, RenderArgs a, R* __restrict__ accum, unsigned long long* __restrict__ counters)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const Op<R>* ops = sc.ops;
    const Prim<R>* prims = sc.prims;
    if (kTrav == 1) {
        // stage the whole traversal program into LDS: 16-byte copies, all 256 threads
        const uint4* src0 = reinterpret_cast<const uint4*>(sc.ops);
        uint4* dst0 = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = threadIdx.x; i < a.lds_ops_bytes / 16; i += blockDim.x) dst0[i] = src0[i];
        const uint4* src1 = reinterpret_cast<const uint4*>(sc.prims);
        uint4* dst1 = reinterpret_cast<uint4*>(smem + a.lds_ops_bytes);
        for (uint32_t i = threadIdx.x; i < a.lds_prims_bytes / 16; i += blockDim.x) dst1[i] = src1[i];
        __syncthreads();
        ops = reinterpret_cast<const Op<R>*>(smem);
        prims = reinterpret_cast<const Prim<R>*>(smem + a.lds_ops_bytes);
    }

    const int tiles_x = (a.width + 15) >> 4;
    const int tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
    const int lane = threadIdx.x & 63;
    // pixel of slot `local` (0..255) of this workgroup's 16x16 tile: waves are 8x8 sub-tiles
    auto pixel_of = [&](int local, int& jj, int& ii) {
        const int wv = local >> 6, ln = local & 63;
        jj = tile_x * 16 + (wv & 1) * 8 + (ln & 7);
        ii = a.y_begin + tile_y * 16 + (wv >> 1) * 8 + (ln >> 3);
    };

    // Work distribution.  fp64 reference mode: a lane owns one pixel and walks its samples (deterministic sums, bit-identical
    // under spp / row splits).  fp32 modes: the WORKGROUP owns the tile; lanes pull (sample, pixel) items from an LDS
    // counter -- item = sample * 256 + slot, so a wave's simultaneous fetches are neighbouring pixels of one sample -- and
    // add finished samples into an LDS tile buffer.  A pixel whose paths are 20x longer than its neighbours' (a glass
    // sphere's silhouette) then no longer pins 63 idle lanes to its wave: measured bounce-level lane utilisation 0.59 -> ~1.
    constexpr bool kDynamic = sizeof(R) == sizeof(float);
    float* tile_sum = reinterpret_cast<float*>(smem + a.lds_tile_offset);
    unsigned int* item_counter = reinterpret_cast<unsigned int*>(tile_sum + 768);
    if (kDynamic) {
        tile_sum[threadIdx.x * 3] = 0.f; tile_sum[threadIdx.x * 3 + 1] = 0.f; tile_sum[threadIdx.x * 3 + 2] = 0.f;
        if (threadIdx.x == 0) *item_counter = 0u;
        __syncthreads();
    }
    const unsigned int n_items = 256u * (unsigned int) (a.spp_end - a.spp_begin);

    unsigned long long n_rays = 0, n_rejected = 0;
    VisitStats vs; vs.n_box = 0; vs.n_leaf = 0; vs.hash = 0; vs.w_node = vs.w_leaf = vs.w_bounce = 0;
    vec3<R> sum;
    int i, j, local = threadIdx.x;
    pixel_of(local, j, i);
    const bool own_in_image = j < a.width && i < a.y_end;
    // lane-owned pixels (fp64 mode): RT_FLAG_ACCUMULATE continues the running sum in the buffer sample by sample, so a frame
    // resumed from a checkpoint adds in the same order as an uninterrupted one and equals it bit for bit
    const bool resume = !kDynamic && own_in_image && (a.flags & RT_FLAG_ACCUMULATE);
    if (resume) { const R* in = accum + ((size_t) i * a.width + j) * 3; sum = vec3<R>(in[0], in[1], in[2]); }

    if (kDynamic || own_in_image) {
        uint32_t pixel = (uint32_t) (i * a.width + j);
        int k = a.spp_begin;
        bool alive = false, poisoned = false;
        int depth = 0, path_rays = 0;
        Rng g; g.s = 0;
        Ray<R> ray;
        vec3<R> beta, L;

        for (;;) {
            if (!alive) {
                if (kDynamic) {
                    // one LDS atomic per wave: the first fetching lane reserves a run of items for all fetching lanes
                    const unsigned long long need = __ballot(1);
                    const int leader = __ffsll(need) - 1;
                    unsigned int base = 0;
                    if (lane == leader) base = atomicAdd(item_counter, (unsigned int) __popcll(need));
                    base = __shfl(base, leader, 64);
                    const unsigned int item = base + (unsigned int) __popcll(need & ((1ull << lane) - 1ull));
                    if (item >= n_items) break;
                    local = (int) (item & 255u);
                    k = a.spp_begin + (int) (item >> 8);
                    pixel_of(local, j, i);
                    if (!(j < a.width && i < a.y_end)) continue;     // slot outside the image (edge tiles)
                    pixel = (uint32_t) (i * a.width + j);
                }
                else if (k >= a.spp_end) break;
                g.seed(a.seed, pixel, (uint32_t) k);
                k++;
                R x = (R(j) + uniform<R>(g)) / R(a.width - 1);                             // raytracing.cpp:102-103
                R y = (R(i) + uniform<R>(g)) / R(a.height - 1);
                ray = camera_emit(sc.cam, g, x, y);
                beta = vec3<R>(R(1), R(1), R(1));
                L = vec3<R>();
                depth = a.max_depth;
                poisoned = false;
                path_rays = 0;
                alive = depth > 0;
                if (!alive) continue;
            }

            // ---- one path segment: trace() body for the current depth
            n_rays++;
            path_rays++;
            if (kCount && first_active_lane()) vs.w_bounce++;
            if (kCount && ray.d.x == R(0) && ray.d.y == R(0) && ray.d.z == R(0)) vs.n_zero++;
            R t;
            int prim;
            if constexpr (kTrav == 2) prim = closest_hit_bvh<kCount ? 1 : 0>(sc.bnodes, sc.fprims, sc.side(), ray, t, reinterpret_cast<uint32_t*>(smem) + threadIdx.x, &vs, &g);
            else prim = closest_hit<R, kCount ? 1 : 0>(ops, sc.n_ops, prims, sc.side(), ray, t, &vs, &g);
            bool done = true;
            if (prim >= 0) {
                Hit<R> h;
                fill_record(kTrav == 2 ? sc.fprims[prim] : prims[prim], sc.side(), ray, t, sc.need_uv != 0, h);
                const Mat<R>& m = sc.mats[h.mat];
                const vec3<R> texc = m.tex >= 0 ? tex_color(sc, m.tex, h.pos, h.u, h.v) : vec3<R>();
                vec3<R> emission = mat_emit(m, h, texc);
                Ray<R> wo; wo.time = R(0);                                                  // `ray wo;` raytracing.cpp:41
                vec3<R> att;
                bool spec = false;
                L += beta * emission;
                const bool fixed = (a.flags & RT_FLAG_COMPAT_FIXED) != 0;
                if (mat_shading(g, m, ray, h, texc, wo, att, spec, fixed)) {
                    R pdf, mpdf;
                    bool dropped = false;
                    if (fixed) {                                                            // no counterpart in the reference: see rtcore.h RT_FLAG_COMPAT_FIXED
                        if (spec) beta = beta * att;                                        // a delta lobe weighs its attenuation
                        else {
                            if (uniform<R>(g) < R(0.5)) { wo.o = h.pos; wo.from = h.id; wo.d = light_rand(sc, g, h.pos); wo.time = ray.time; }
                            mpdf = mat_pdf(g, m, h, wo, true);
                            pdf = R(0.5) * (lights_pdf(sc, h.pos, wo.d) + mpdf);
                            if (!(pdf > R(0)) || !(mpdf > R(0))) dropped = true;            // a direction of zero probability contributes nothing
                            else beta = beta * ((R(1) / pdf) * (mpdf * att));
                        }
                    }
                    else {
                    if (spec) {                                                             // raytracing.cpp:51-54
                        mpdf = mat_pdf(g, m, h, wo);
                        pdf = mpdf;
                    }
                    else {                                                                  // raytracing.cpp:55-61
                        if (uniform<R>(g) < R(0.5)) { wo.o = h.pos; wo.from = h.id; wo.d = light_rand(sc, g, h.pos); wo.time = ray.time; }
                        mpdf = mat_pdf(g, m, h, wo);
                        pdf = R(0.5) * (lights_pdf(sc, h.pos, wo.d) + mpdf);
                    }
                    R inv = R(1) / pdf;                                                     // raytracing.cpp:69, vec3.hpp:121-123
                    vec3<R> w = mpdf * att;
                    if (!rt_finite(inv) || !rt_finite(w.x) || !rt_finite(w.y) || !rt_finite(w.z)) poisoned = true;
                    beta = beta * (inv * w);
                    }
                    if (!dropped) {
                        ray = wo;
                        depth--;
                        done = depth <= 0;
                        if (poisoned && (a.flags & RT_FLAG_EARLY_EXIT)) done = true;
                    }
                }
            }
            if (done) {
                alive = false;
                bool bad = poisoned || !rt_finite(L.x) || !rt_finite(L.y) || !rt_finite(L.z);  // raytracing.cpp:108-110
                if (bad) n_rejected++;
                if (a.flags & RT_FLAG_DEBUG_RAY_COUNTS) { L = vec3<R>(R(path_rays), bad ? R(1) : R(0), R(0)); bad = false; }
                if (!bad) {
                    if (kDynamic) {
                        float* dst = tile_sum + local * 3;
                        __hip_atomic_fetch_add(dst, (float) L.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(dst + 1, (float) L.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(dst + 2, (float) L.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    else sum += L;
                }
            }
        }
    }

    if (kDynamic) {
        __syncthreads();
        sum = vec3<R>(R(tile_sum[threadIdx.x * 3]), R(tile_sum[threadIdx.x * 3 + 1]), R(tile_sum[threadIdx.x * 3 + 2]));
        pixel_of((int) threadIdx.x, j, i);
    }
    if (own_in_image) {
        R* out = accum + ((size_t) i * a.width + j) * 3;
        if ((a.flags & RT_FLAG_ACCUMULATE) && !resume) { out[0] += sum.x; out[1] += sum.y; out[2] += sum.z; }
        else { out[0] = sum.x; out[1] = sum.y; out[2] = sum.z; }
    }

    // wave-level reduction of the counters, one atomic pair per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_rays += __shfl_down(n_rays, off, 64);
        n_rejected += __shfl_down(n_rejected, off, 64);
    }
    if (lane == 0) {
        atomicAdd(&counters[0], n_rays);
        if (n_rejected) atomicAdd(&counters[1], n_rejected);
    }
    if (kCount) {   // diagnostic launches only: per-lane 32-bit counts are ample for one launch's per-pixel work
        unsigned long long nb = vs.n_box, nl = vs.n_leaf, wn = vs.w_node, wl = vs.w_leaf, wb = vs.w_bounce, nz = vs.n_zero;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            nb += __shfl_down(nb, off, 64); nl += __shfl_down(nl, off, 64); nz += __shfl_down(nz, off, 64);
            wn += __shfl_down(wn, off, 64); wl += __shfl_down(wl, off, 64); wb += __shfl_down(wb, off, 64);
        }
        if (lane == 0) { atomicAdd(&counters[2], nb); atomicAdd(&counters[3], nl); atomicAdd(&counters[4], wn); atomicAdd(&counters[5], wl); atomicAdd(&counters[6], wb); atomicAdd(&counters[7], nz); }
    }
}

// ------------------------------------------------------------------ parity kernel: explicit rays, with visit statistics
template <typename R>
__global__ void __launch_bounds__(256) first_hit_kernel(Scene<R> sc, const double* __restrict__ rays, int n, int32_t* prim, double* t_out,
                                                        int32_t* front, uint32_t* n_box, uint32_t* n_leaf, unsigned long long* hash)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const double* p = rays + (size_t) idx * 7;
    Ray<R> r;
    r.o = vec3<R>(R(p[0]), R(p[1]), R(p[2]));
    r.d = vec3<R>(R(p[3]), R(p[4]), R(p[5]));
    r.time = R(p[6]);
    VisitStats st; st.n_box = 0; st.n_leaf = 0; st.hash = 0xcbf29ce484222325ULL; st.w_node = st.w_leaf = st.w_bounce = 0;
    R t;
    int hit = closest_hit<R, 2>(sc.ops, sc.n_ops, sc.prims, sc.side(), r, t, &st);
    prim[idx] = hit;
    t_out[idx] = hit >= 0 ? (double) t : 0.0;
    int f = -1;
    if (hit >= 0) {
        Hit<R> h;
        fill_record(sc.prims[hit], sc.side(), r, t, false, h);
        f = h.front ? 1 : 0;
    }
    front[idx] = f;
    n_box[idx] = st.n_box; n_leaf[idx] = st.n_leaf; hash[idx] = st.hash;
}


#ifdef RT_WITH_FIRST_HIT        // (the fp64 translation unit)
// ------------------------------------------------------------------ debug kernels (test tooling: tools/first_divergence.py)
// The path of ONE (pixel, sample) stream through the reference-order walk with the render kernel's own functions (render_kernel above,
// R = double, kTrav = 0), one lane: per segment the ray (origin, direction, time), the distance of its hit (0: a miss) and the
// primitive (-1).  -ffp-contract=off makes the arithmetic independent of the surrounding code, so these are the rays the frame traced.
template <typename R>
__global__ void path_kernel(Scene<R> sc, RenderArgs a, int i, int j, int k, double* __restrict__ rows9, int cap, int* __restrict__ n_out)
{
    if (blockIdx.x || threadIdx.x) return;
    Rng g; g.seed(a.seed, (uint32_t) (i * a.width + j), (uint32_t) k);
    R x = (R(j) + uniform<R>(g)) / R(a.width - 1);
    R y = (R(i) + uniform<R>(g)) / R(a.height - 1);
    Ray<R> ray = camera_emit(sc.cam, g, x, y);
    VisitStats vs;
    const bool fixed = (a.flags & RT_FLAG_COMPAT_FIXED) != 0;
    int n = 0;
    for (int depth = a.max_depth; depth > 0; depth--) {
        R t = R(0);
        const int prim = closest_hit<R, 0>(sc.ops, sc.n_ops, sc.prims, sc.side(), ray, t, &vs, &g);
        if (n < cap) {
            double* w = rows9 + (size_t) n * 9;
            w[0] = ray.o.x; w[1] = ray.o.y; w[2] = ray.o.z; w[3] = ray.d.x; w[4] = ray.d.y; w[5] = ray.d.z; w[6] = ray.time; w[7] = prim >= 0 ? (double) t : 0.0; w[8] = prim;
        }
        n++;
        if (prim < 0) break;
        Hit<R> h;
        fill_record(sc.prims[prim], sc.side(), ray, t, sc.need_uv != 0, h);
        const Mat<R>& m = sc.mats[h.mat];
        const vec3<R> texc = m.tex >= 0 ? tex_color(sc, m.tex, h.pos, h.u, h.v) : vec3<R>();
        Ray<R> wo; wo.time = R(0);
        vec3<R> att;
        bool spec = false;
        if (!mat_shading(g, m, ray, h, texc, wo, att, spec, fixed)) break;
        if (!(fixed && spec)) {
            if (!spec && uniform<R>(g) < R(0.5)) { wo.o = h.pos; wo.from = h.id; wo.d = light_rand(sc, g, h.pos); wo.time = ray.time; }
            const R mpdf = mat_pdf(g, m, h, wo, fixed && !spec);
            if (fixed && !spec) { const R pdf = R(0.5) * (lights_pdf(sc, h.pos, wo.d) + mpdf); if (!(pdf > R(0)) || !(mpdf > R(0))) break; }
        }
        ray = wo;
    }
    *n_out = n;
}

// the device library's results for the libm calls of a path (operations as oracle/rt_oracle.c numbers them: 1 sin, 2 cos, 3 acos,
// 4 atan2(x, y), 5 pow(x, y), 6 log), through the same rt_* overloads the kernels call; 7: sine_sign(x), the checker texture's sign of sin(x); 8: pow5(x), the dielectric's (1 - cos)^5
__global__ void libm_kernel(const int32_t* __restrict__ op, const double* __restrict__ x, const double* __restrict__ y, int n, double* __restrict__ out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    double r = 0.0;
    switch (op[idx]) {
    case 1: r = rt_sin(x[idx]); break;
    case 2: r = rt_cos(x[idx]); break;
    case 3: r = rt_acos(x[idx]); break;
    case 4: r = rt_atan2(x[idx], y[idx]); break;
    case 5: r = y[idx] == 5.0 ? pow5(x[idx]) : rt_pow(x[idx], y[idx]); break;      // (the one fifth power of a path is the dielectric's: what the kernels compute for it)
    case 6: r = rt_log(x[idx]); break;
    case 7: r = (double) sine_sign(x[idx]); break;
    case 8: r = pow5(x[idx]); break;
    default: break;
    }
    out[idx] = r;
}
#endif

#ifdef RT_WITH_FAST_BVH
// fast-mode closest hit of explicit rays (test hook): primitive id and t only (defined in the fp32 translation unit only)
__global__ void __launch_bounds__(256) first_hit_fast_kernel(Scene<float> sc, const double* __restrict__ rays, int n, int32_t* prim, float* t_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const double* p = rays + (size_t) idx * 7;
    Ray<float> r;
    r.o = vec3<float>((float) p[0], (float) p[1], (float) p[2]);
    r.d = vec3<float>((float) p[3], (float) p[4], (float) p[5]);
    r.time = (float) p[6];
    float t;
    int hit = closest_hit_bvh<0>(sc.bnodes, sc.fprims, sc.side(), r, t, reinterpret_cast<uint32_t*>(smem) + threadIdx.x, nullptr);
    prim[idx] = hit >= 0 ? sc.fprims[hit].id : -1;
    t_out[idx] = hit >= 0 ? t : 0.0f;
}
#endif

}  // namespace rt
