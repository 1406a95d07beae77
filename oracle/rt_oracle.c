/* oracle/rt_oracle.c -- TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * Plain-C fp64 restatement of the reference CPU renderer's hot path.  Each
 * function cites the reference file:line it follows (paths under
 * /root/reference/).  Arithmetic is written in the reference's evaluation
 * order (e.g. vec/T == (1/T)*vec, vec3.hpp:121-123) so that, fed the same
 * rand() stream, results are bit-identical to the g++ -O3 build of the
 * reference; build with -ffp-contract=off.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against
 * tests/golden/ npz files (made by oracle/gen_golden.py from the real reference).
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ constants (util.hpp:12-14) */
static const double ORC_PI = 3.141592653589;      /* truncated on purpose, util.hpp:13 */
static const double ORC_EPS = 1e-7;
#define ORC_INF (1.0 / 0.0)

/* ------------------------------------------------------------------ vec3 (include/algebra/vec/vec3.hpp) */
typedef struct { double x, y, z; } v3;
typedef struct { double u, v; } v2;

static inline v3 V(double x, double y, double z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 smul(double s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }              /* vec3.hpp:109-115 */
static inline v3 sdiv(v3 a, double s) { return smul(1 / s, a); }                             /* vec3.hpp:121-123 */
static inline double vdot(v3 a, v3 b) { double r1 = a.x * b.x, r2 = a.y * b.y, r3 = a.z * b.z; return r1 + r2 + r3; } /* :67-72 */
static inline v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* libm calls of the per-path code go through these, so that orc_debug_libm can list them (operation, arguments, result, and the path
 * segment during whose shading they were made): tools/first_divergence.py evaluates the same arguments on the GPU and shows where
 * glibc's and the device library's last bits part.  Logging is off (one predictable branch) except inside orc_debug_libm. */
static double* g_lm_out = 0; static int g_lm_n = 0, g_lm_cap = 0;
static int g_lm_segment = 0;
static inline double lm_log(int op, double x, double y, double r)
{
    if (g_lm_out && g_lm_n < g_lm_cap) { double* w = g_lm_out + 5 * g_lm_n++; w[0] = g_lm_segment; w[1] = op; w[2] = x; w[3] = y; w[4] = r; }
    return r;
}
static inline double LM_sin(double x) { return lm_log(1, x, 0, sin(x)); }
static inline double LM_cos(double x) { return lm_log(2, x, 0, cos(x)); }
static inline double LM_acos(double x) { return lm_log(3, x, 0, acos(x)); }
static inline double LM_atan2(double y, double x) { return lm_log(4, y, x, atan2(y, x)); }
static inline double LM_pow(double x, double y) { return lm_log(5, x, y, pow(x, y)); }
static inline double LM_log(double x) { return lm_log(6, x, 0, log(x)); }

static inline double vlen(v3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }          /* :25-27 (pow(.,2) folds to x*x) */
static inline v3 vnorm(v3 a) { double l = vlen(a); if (l == 0) return V(0, 0, 0); return sdiv(a, l); } /* :60-65 */
static inline double vget(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

/* ------------------------------------------------------------------ rng */
typedef struct {
    int mode;
    /* glibc random_r TYPE_3 state */
    int32_t r[31]; int f, b;
    /* counter stream */
    uint64_t state;
    uint64_t draws;
} orc_rng;

static void glibc_seed(orc_rng* g, unsigned seed)
{
    /* glibc stdlib/random_r.c __srandom_r, TYPE_3: degree 31, separation 3 */
    int32_t word = (int32_t) seed;
    if (word == 0) word = 1;
    g->r[0] = word;
    for (int i = 1; i < 31; i++) {
        long hi = word / 127773, lo = word % 127773;
        word = (int32_t) (16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        g->r[i] = word;
    }
    g->f = 3; g->b = 0;
    for (int i = 0; i < 310; i++) {
        g->r[g->f] = (int32_t) ((uint32_t) g->r[g->f] + (uint32_t) g->r[g->b]);
        g->f = (g->f + 1) % 31; g->b = (g->b + 1) % 31;
    }
}

static inline int32_t glibc_rand(orc_rng* g)
{
    uint32_t v = (uint32_t) g->r[g->f] + (uint32_t) g->r[g->b];
    g->r[g->f] = (int32_t) v;
    g->f = (g->f + 1) % 31; g->b = (g->b + 1) % 31;
    return (int32_t) (v >> 1);
}

/* counter-based stream shared bit-for-bit with raytracing_amd/csrc (rt_rng.h): PCG32 XSH-RR whose
 * state is a hash of (seed, pixel, sample); no state survives between samples or launches. */
static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
#define PCG_MULT 6364136223846793005ULL
#define PCG_INC  1442695040888963407ULL
static void counter_seed(orc_rng* g, uint64_t seed, uint32_t pixel, uint32_t sample)
{
    uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ULL * (1ULL + pixel));
    h = mix64(h ^ (0xD1B54A32D192ED03ULL * (1ULL + sample)));
    g->mode = ORC_RNG_COUNTER;
    g->state = h;
}
static inline uint32_t counter_next(orc_rng* g)
{
    uint64_t old = g->state;
    g->state = old * PCG_MULT + PCG_INC;
    uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t) (old >> 59);
    return (xs >> rot) | (xs << ((32 - rot) & 31));
}

/* util.hpp:30-33: rand() / (RAND_MAX + 1.0) */
static inline double rnd(orc_rng* g)
{
    g->draws++;
    if (g->mode == ORC_RNG_GLIBC) return glibc_rand(g) / 2147483648.0;
    return counter_next(g) * (1.0 / 4294967296.0);
}
static inline double rnd_range(orc_rng* g, double mn, double mx) { return mn + (mx - mn) * rnd(g); }  /* util.hpp:36-39 */
static inline int rnd_int(orc_rng* g, int mn, int mx) { return (int) rnd_range(g, mn, mx + 1); }     /* util.hpp:42-45 */

/* g++ evaluates the arguments of these constructor calls right-to-left (x86-64 SysV, pinned by the
 * golden fixtures): the LAST argument draws first. */
static inline v3 rnd_vec01(orc_rng* g)                                    /* util.hpp:48-51 */
{ double z = rnd(g), y = rnd(g), x = rnd(g); return V(x, y, z); }
static inline v3 rnd_vec(orc_rng* g, double mn, double mx)                /* util.hpp:54-57 */
{ double z = rnd_range(g, mn, mx), y = rnd_range(g, mn, mx), x = rnd_range(g, mn, mx); return V(x, y, z); }

static v3 rnd_sphere(orc_rng* g)                                          /* util.hpp:60-70 "random_shpere" */
{
    for (;;) {
        v3 p = rnd_vec(g, -1, 1);
        double l = vlen(p);
        if (l * l >= 1) continue;
        return vnorm(p);
    }
}
static v3 rnd_disk(orc_rng* g)                                            /* util.hpp:85-95 */
{
    for (;;) {
        double y = rnd_range(g, -1, 1), x = rnd_range(g, -1, 1);
        v3 p = V(x, y, 0);
        double l = vlen(p);
        if (l * l >= 1) continue;
        return p;
    }
}
static v3 rnd_cosine(orc_rng* g)                                          /* util.hpp:98-107 */
{
    double r1 = rnd(g), r2 = rnd(g);
    double z = sqrt(1 - r2);
    double phi = 2 * ORC_PI * r1;
    double x = LM_cos(phi) * sqrt(r2);
    double y = LM_sin(phi) * sqrt(r2);
    return V(x, y, z);
}
static inline v3 reflect(v3 in, v3 n) { return vsub(in, smul(2 * vdot(in, n), n)); }          /* util.hpp:154-157 */
static v3 refract(v3 in, v3 n, double ratio)                                                  /* util.hpp:160-166 */
{
    double cos_theta = fmin(-vdot(in, n), 1.0);
    v3 perp = smul(ratio, vadd(in, smul(cos_theta, n)));
    double pl = vlen(perp);
    v3 par = smul(-sqrt(fabs(1.0 - pl * pl)), n);
    return vadd(perp, par);
}

/* ------------------------------------------------------------------ ray / hit record (ray.hpp, intersection.hpp) */
typedef struct { v3 o, d; double time, t_min, t_max; } ray;
static inline ray mkray(v3 o, v3 d, double time, double t_min, double t_max) { ray r = { o, d, time, t_min, t_max }; return r; }
static inline ray ray_default(void) { return mkray(V(0, 0, 0), V(0, 0, 0), 0, 1e-3, ORC_INF); }
static inline v3 ray_at(const ray* r, double t) { return vadd(r->o, smul(t, r->d)); }

struct orc_obj;
typedef struct { v3 pos, normal; v2 uv; double t; int mat; int front; const struct orc_obj* leaf; } isect;
static inline void set_face_normal(isect* x, const ray* r, v3 n)        /* intersection.hpp:22-26 */
{
    x->front = vdot(r->d, n) < 0;
    x->normal = x->front ? n : vneg(n);
}

/* ------------------------------------------------------------------ aabb (aabb.hpp) */
typedef struct { v3 mn, mx; } aabb;
static int aabb_hit(const aabb* b, const ray* r)                        /* aabb.hpp:23-41 */
{
    double t_min = r->t_min, t_max = r->t_max;
    for (int i = 0; i < 3; i++) {
        double inv = 1.0 / vget(r->d, i);
        double t0 = (vget(b->mn, i) - vget(r->o, i)) * inv;
        double t1 = (vget(b->mx, i) - vget(r->o, i)) * inv;
        if (inv < 0.0) { double s = t0; t0 = t1; t1 = s; }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return 0;
    }
    return 1;
}
static aabb surrounding(aabb a, aabb b)                                 /* aabb.hpp:43-56 */
{
    aabb r;
    r.mn = V(fmin(a.mn.x, b.mn.x), fmin(a.mn.y, b.mn.y), fmin(a.mn.z, b.mn.z));
    r.mx = V(fmax(a.mx.x, b.mx.x), fmax(a.mx.y, b.mx.y), fmax(a.mx.z, b.mx.z));
    return r;
}

/* ------------------------------------------------------------------ textures / materials */
typedef struct { int kind; double d[ORC_TEX_STRIDE]; const uint8_t* pixels; int pix_off; } orc_tex;
typedef struct { int kind; int tex; double d[ORC_MAT_STRIDE]; } orc_mat;
/* mat d[]: [0] roughness | ior ; phong: [1..3] Kd [4..6] Ks [7..9] Ke [10..12] Tr [13] Ns [14] Ni */

enum { O_BVH = 100, O_FLIP = 101, O_TRANSLATE = 102, O_ROTATEY = 103, O_GROUP = 104 };   /* translate: g[0..2] target; rotatey: g[0] sin, g[1] cos, g[2] has_bbox, box; group: items */
typedef struct orc_obj {
    int kind;                 /* ORC_SPHERE.. | O_BVH | O_FLIP */
    int id;                   /* prim id (insertion order into scene.objects), -1 for bvh nodes / light-only objects */
    int mat;
    double g[ORC_PRIM_STRIDE];
    struct orc_obj *l, *r;    /* bvh children; flip / translate / rotatey / volume: l = inner */
    struct orc_obj** items; int n_items;   /* group used as an object (group.hpp:12-124) */
    int first_entry;          /* first pre-order index (pre-order view), -1 until walked */
    aabb box;                 /* bvh */
} orc_obj;

struct orc_scene {
    orc_rng rng;
    int width, height;
    orc_obj** objects; int n_objects; int cap_objects;       /* scene.objects (group) */
    orc_obj** lights; int n_lights; int cap_lights;          /* scene.lights  (group) */
    orc_obj** prims; int n_prims; int cap_prims;             /* leaves by prim id */
    orc_mat* mats; int n_mats;
    orc_tex* texs; int n_texs;
    uint8_t* pixels; int n_pixel_bytes;
    /* camera (camera.hpp:12-18,71-77) */
    v3 cam_pos, cam_lookat, cam_up; double fov, aspect, aperture, focus, shutter0, shutter1;
    v3 left_bottom, horizontal, vertical, cu, cv, cw; double lens_radius;
    /* pre-order view, built lazily */
    const orc_obj** pre; int* pre_first; int* pre_left; int* pre_right; int n_pre;
    int* root_entry;
    uint64_t rays;
    int wrapper_rows; /* orc_tree_rows also writes the wrapper chain of every entry (the `nested` fixture; ref_harness.cpp g_wrapper_rows) */
    int early_exit;   /* stop a path once its sample is known to be rejected (same image, fewer rays) */
    int compat_fixed; /* no counterpart in the reference: the estimator with its defects repaired (see orc_set_compat_fixed) */
};

static v3 tex_color(const orc_scene* s, int ti, v3 p, v2 uv)
{
    const orc_tex* t = &s->texs[ti];
    switch (t->kind) {
    case ORC_TEX_CONSTANT: return V(t->d[0], t->d[1], t->d[2]);             /* texture.hpp:32-35 */
    case ORC_TEX_CHECKER: {                                                 /* texture.hpp:65-74 */
        double sine = LM_sin(10 * p.x) * LM_sin(10 * p.y) * LM_sin(10 * p.z);
        return tex_color(s, (int) (sine < 0 ? t->d[0] : t->d[1]), p, uv);
    }
    case ORC_TEX_IMAGE: {                                                   /* texture.hpp:95-124 */
        int w = (int) t->d[0], h = (int) t->d[1], ch = (int) t->d[2];
        if (!t->pixels) return V(0, 1, 1);
        double u = uv.u < 0 ? 0 : (uv.u > 1 ? 1 : uv.u);
        double vv = uv.v < 0 ? 0 : (uv.v > 1 ? 1 : uv.v);
        double v = 1.0 - vv;
        /* a NaN uv (acos of a sphere normal component a rounding error beyond 1, primitive.hpp:69) makes the reference
         * index out of bounds (undefined behaviour); here, as on the GPU (v_cvt_i32_f32 of NaN is 0), it reads texel 0 */
        int x = isnan(u) ? 0 : (int) (u * w), y = isnan(v) ? 0 : (int) (v * h);
        if (x >= w) x = w - 1;
        if (y >= h) y = h - 1;
        const double scale = 1.0 / 255;
        const uint8_t* b = t->pixels + (y * w * ch + x * ch);
        v3 c = smul(scale, V(b[0], b[1], b[2]));
        if (t->d[3] != 0) return V(LM_pow(c.x, 2.2), LM_pow(c.y, 2.2), LM_pow(c.z, 2.2));
        return c;
    }
    }
    return V(0, 0, 0);
}

static inline int nz3(const double* k) { return k[0] > 0 || k[1] != 0 || k[2] != 0; }    /* material.hpp:298-302 */

static v3 mat_emit(const orc_scene* s, const orc_mat* m, const isect* x)
{
    if (m->kind == ORC_MAT_EMISSIVE) return x->front ? tex_color(s, m->tex, x->pos, x->uv) : V(0, 0, 0);  /* material.hpp:146-154 */
    if (m->kind == ORC_MAT_PHONG) return (nz3(m->d + 7) && x->front) ? V(m->d[7], m->d[8], m->d[9]) : V(0, 0, 0); /* :201-209 */
    return V(0, 0, 0);
}

static double reflectance(double cos_theta, double ior)                  /* material.hpp:128-133 */
{
    double r = (1 - ior) / (1 + ior);
    r = r * r;
    return r + (1 - r) * LM_pow((1 - cos_theta), 5);
}

static v3 cosine_sample(orc_rng* g, v3 normal)                           /* material.hpp:43-52 */
{
    v3 rc = rnd_cosine(g);
    v3 a2 = vnorm(normal);
    v3 x = (fabs(a2.x) > 0.9) ? V(0, 1, 0) : V(1, 0, 0);
    v3 a1 = vnorm(vcross(a2, x));
    v3 a0 = vcross(a2, a1);
    return vadd(vadd(smul(rc.x, a0), smul(rc.y, a1)), smul(rc.z, a2));
}

static v3 glass_dir(orc_rng* g, const ray* wi, const isect* x, double ior)   /* material.hpp:109-122 / :234-247 */
{
    double ratio = x->front ? (1.0 / ior) : ior;
    v3 next = vnorm(wi->d);
    double cos_theta = fmin(vdot(vneg(next), x->normal), 1.0);
    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
    int not_refract = ratio * sin_theta > 1.0;
    if (not_refract || reflectance(cos_theta, ratio) > rnd(g)) return reflect(next, x->normal);
    return refract(next, x->normal, ratio);
}

static int mat_shading(const orc_scene* s, orc_rng* g, const orc_mat* m, const ray* wi, const isect* x,
                       ray* wo, v3* att, int* spec)
{
    switch (m->kind) {
    case ORC_MAT_LAMBERTIAN:                                             /* material.hpp:29-35 */
        *spec = 0;
        *wo = mkray(x->pos, cosine_sample(g, x->normal), wi->time, wi->t_min, wi->t_max);
        *att = tex_color(s, m->tex, x->pos, x->uv);
        return 1;
    case ORC_MAT_METAL: {                                                /* material.hpp:74-82 */
        v3 next = reflect(vnorm(wi->d), x->normal);
        v3 fuzz = smul(m->d[0], rnd_sphere(g));
        *wo = mkray(x->pos, vadd(next, fuzz), wi->time, wi->t_min, wi->t_max);
        *att = tex_color(s, m->tex, x->pos, x->uv);
        *spec = 1;
        if (s->compat_fixed && !(vdot(wo->d, x->normal) > 0)) return 0;     /* fixed: a fuzzed reflection below the surface is absorbed */
        return 1;
    }
    case ORC_MAT_DIELECTRIC:                                             /* material.hpp:105-125 */
        *spec = 1;
        *att = V(1, 1, 1);
        *wo = mkray(x->pos, glass_dir(g, wi, x, m->d[0]), wi->time, wi->t_min, wi->t_max);
        return 1;
    case ORC_MAT_EMISSIVE:                                               /* material.hpp:156-159 */
        return 0;
    case ORC_MAT_FOG:                                                    /* material.hpp:174-180; pdf() is the base class's 0 (:14) */
        *spec = 0;
        *wo = mkray(x->pos, rnd_sphere(g), wi->time, wi->t_min, wi->t_max);
        *att = tex_color(s, m->tex, x->pos, x->uv);
        return 1;
    case ORC_MAT_PHONG: {                                                /* material.hpp:211-255 */
        if (nz3(m->d + 7)) return 0;
        v3 base = (m->tex >= 0) ? tex_color(s, m->tex, x->pos, x->uv) : V(m->d[1], m->d[2], m->d[3]);
        if (nz3(m->d + 4) && rnd(g) < 0.5) {       /* specular lobe: wo is NOT set */
            *att = V(m->d[4], m->d[5], m->d[6]);
            *spec = 1;
            if (s->compat_fixed) {                  /* fixed: a mirror reflection */
                *wo = mkray(x->pos, reflect(vnorm(wi->d), x->normal), wi->time, wi->t_min, wi->t_max);
                if (!(vdot(wo->d, x->normal) > 0)) return 0;
            }
            return 1;
        }
        if (nz3(m->d + 10)) {
            *spec = 1;
            *att = V(m->d[10], m->d[11], m->d[12]);
            *wo = mkray(x->pos, glass_dir(g, wi, x, m->d[14]), wi->time, wi->t_min, wi->t_max);
            return 1;
        }
        *spec = 0;                                  /* diffuse lobe: wo is NOT set */
        *att = base;
        if (s->compat_fixed) *wo = mkray(x->pos, cosine_sample(g, x->normal), wi->time, wi->t_min, wi->t_max);   /* fixed: cosine lobe */
        return 1;
    }
    }
    return 0;
}

static double mat_pdf(orc_rng* g, const orc_mat* m, const isect* x, const ray* wo, int fixed)
{
    switch (m->kind) {
    case ORC_MAT_LAMBERTIAN: {                                           /* material.hpp:37-41 */
        double c = vdot(x->normal, vnorm(wo->d));
        return c > 0 ? c / ORC_PI : 0;
    }
    case ORC_MAT_METAL: {                                                /* material.hpp:84-88 */
        double c = vdot(x->normal, vnorm(wo->d));
        return c > ORC_EPS ? 1 : 0;
    }
    case ORC_MAT_PHONG: {                                                /* material.hpp:257-266 */
        double c = vdot(x->normal, vnorm(wo->d));
        if (!fixed && nz3(m->d + 4) && rnd(g) < 0.5) return c > ORC_EPS ? 1 : 0;   /* fixed: the diffuse lobe's pdf, no second roll */
        return c > 0 ? c / ORC_PI : 0;
    }
    default: return 0.0;                                                 /* material.hpp:14 (dielectric, emissive) */
    }
}

/* ------------------------------------------------------------------ primitives */
static v2 sphere_uv(v3 p)                                                /* primitive.hpp:67-74 */
{
    double theta = LM_acos(-p.y);
    double phi = LM_atan2(-p.z, p.x) + ORC_PI;
    v2 r = { phi / (2 * ORC_PI), theta / ORC_PI };
    return r;
}

static int sphere_hit(const orc_obj* o, v3 center, double radius, const ray* r, isect* x)   /* primitive.hpp:29-56 / :103-130 */
{
    v3 oc = vsub(r->o, center);
    double a = vdot(r->d, r->d);
    double b = 2.0 * vdot(oc, r->d);
    double c = vdot(oc, oc) - radius * radius;
    double disc = b * b - 4 * a * c;
    if (disc < 0) return 0;
    double root = (-b - sqrt(disc)) / (a * 2);
    if (root < r->t_min || root > r->t_max) {
        root = (-b + sqrt(disc)) / (a * 2);
        if (root < r->t_min || root > r->t_max) return 0;
    }
    x->t = root;
    x->pos = ray_at(r, root);
    v3 n = sdiv(vsub(x->pos, center), radius);
    set_face_normal(x, r, n);
    x->uv = sphere_uv(n);
    x->mat = o->mat;
    x->leaf = o;
    return 1;
}

static inline v3 msphere_center(const orc_obj* o, double time)          /* primitive.hpp:98-101 */
{
    v3 c0 = V(o->g[0], o->g[1], o->g[2]), c1 = V(o->g[3], o->g[4], o->g[5]);
    return vadd(c0, smul((time - o->g[6]) / (o->g[7] - o->g[6]), vsub(c1, c0)));
}

static int plane_hit(const orc_obj* o, const ray* r, isect* x)          /* primitive.hpp:176-199 / 227-250 / 296-319 */
{
    int ax, bx, kx; v3 n;
    if (o->kind == ORC_PLANEXY) { ax = 0; bx = 1; kx = 2; n = V(0, 0, 1); }
    else if (o->kind == ORC_PLANEXZ) { ax = 0; bx = 2; kx = 1; n = V(0, 1, 0); }
    else { ax = 1; bx = 2; kx = 0; n = V(1, 0, 0); }
    double a0 = o->g[0], a1 = o->g[1], b0 = o->g[2], b1 = o->g[3], k = o->g[4];
    double t = (k - vget(r->o, kx)) / vget(r->d, kx);
    if (t < r->t_min || t > r->t_max) return 0;
    double a = vget(r->o, ax) + t * vget(r->d, ax);
    double b = vget(r->o, bx) + t * vget(r->d, bx);
    if (a < a0 || a > a1 || b < b0 || b > b1) return 0;
    double u = (a - a0) / (a1 - a0);
    double v = (b - b0) / (b1 - b0);
    set_face_normal(x, r, n);
    x->uv.u = u; x->uv.v = v;
    x->t = t;
    x->pos = ray_at(r, t);
    x->mat = o->mat;
    x->leaf = o;
    return 1;
}

static int tri_hit(const orc_obj* o, const ray* r, isect* x)            /* polygon.hpp:90-124 */
{
    const double* g = o->g;
    v3 p0 = V(g[0], g[1], g[2]), e1 = V(g[3], g[4], g[5]), e2 = V(g[6], g[7], g[8]);
    v3 q = vcross(r->d, e2);
    double a = vdot(e1, q);
    if (fabs(a) < ORC_EPS) return 0;
    double f = 1.0 / a;
    v3 s = vsub(r->o, p0);
    double u = f * vdot(s, q);
    if (u < 0) return 0;
    v3 k = vcross(s, e1);
    double v = f * vdot(r->d, k);
    if (v < 0 || u + v > 1) return 0;
    double t = f * vdot(e2, k);
    if (t < r->t_min || t > r->t_max) return 0;
    x->t = t;
    x->pos = ray_at(r, t);
    double w = 1 - u - v;
    /* polygon.hpp:174-182: u*v1 + v*v2 + (1-u-v)*v0 */
    x->uv.u = (u * g[20] + v * g[22]) + w * g[18];
    x->uv.v = (u * g[21] + v * g[23]) + w * g[19];
    v3 n0 = V(g[9], g[10], g[11]), n1 = V(g[12], g[13], g[14]), n2 = V(g[15], g[16], g[17]);
    v3 n = vadd(vadd(smul(u, n1), smul(v, n2)), smul(w, n0));
    set_face_normal(x, r, n);
    x->mat = o->mat;
    x->leaf = o;
    return 1;
}

/* box (primitive.hpp:399-458: six rects tested in a group, group.hpp:68-83) under rotatey and translate
 * (transform.hpp:9-121).  g[0..5] min,max  g[6] sin  g[7] cos  g[8..10] target  g[11] wrappers: 2 rotatey | 4 translate */
static int box_hit(const orc_obj* o, const ray* r, isect* x)
{
    const double* g = o->g;
    const int xf = (int) g[11];
    const v3 target = V(g[8], g[9], g[10]);
    const double sn = g[6], cs = g[7];
    ray tr = *r;                                                         /* translate::intersect, transform.hpp:18-28 */
    if (xf & 4) tr = mkray(vsub(r->o, target), r->d, r->time, r->t_min, r->t_max);
    ray rr = tr;                                                         /* rotatey::intersect, transform.hpp:81-110 */
    if (xf & 2) {
        v3 origin = tr.o, direction = tr.d;
        origin.x = cs * tr.o.x - sn * tr.o.z;
        origin.z = sn * tr.o.x + cs * tr.o.z;
        direction.x = cs * tr.d.x - sn * tr.d.z;
        direction.z = sn * tr.d.x + cs * tr.d.z;
        rr = mkray(origin, direction, tr.time, tr.t_min, tr.t_max);
    }
    orc_obj side[6];                                                     /* box ctor, primitive.hpp:409-431 */
    memset(side, 0, sizeof side);
    const double lo[3] = { g[0], g[1], g[2] }, hi[3] = { g[3], g[4], g[5] };
    for (int i = 0; i < 6; i++) side[i].mat = o->mat;
    side[0].kind = side[1].kind = ORC_PLANEXY; side[2].kind = side[3].kind = ORC_PLANEXZ; side[4].kind = side[5].kind = ORC_PLANEYZ;
    for (int i = 0; i < 2; i++) {
        double* a = side[i].g; a[0] = lo[0]; a[1] = hi[0]; a[2] = lo[1]; a[3] = hi[1]; a[4] = i ? lo[2] : hi[2];
        double* b = side[2 + i].g; b[0] = lo[0]; b[1] = hi[0]; b[2] = lo[2]; b[3] = hi[2]; b[4] = i ? lo[1] : hi[1];
        double* c = side[4 + i].g; c[0] = lo[1]; c[1] = hi[1]; c[2] = lo[2]; c[3] = hi[2]; c[4] = i ? lo[0] : hi[0];
    }
    isect tmp; memset(&tmp, 0, sizeof tmp);                              /* box::intersect -> group::intersect */
    int has = 0;
    double closest = rr.t_max;
    for (int i = 0; i < 6; i++) {
        ray q = mkray(rr.o, rr.d, rr.time, rr.t_min, closest);
        if (plane_hit(&side[i], &q, &tmp)) { has = 1; closest = tmp.t; *x = tmp; }
    }
    if (!has) return 0;
    if (xf & 2) {                                                        /* transform.hpp:99-110 */
        v3 position = x->pos, normal = x->normal;
        position.x = cs * x->pos.x + sn * x->pos.z;
        position.z = -sn * x->pos.x + cs * x->pos.z;
        normal.x = cs * x->normal.x + sn * x->normal.z;
        normal.z = -sn * x->normal.x + cs * x->normal.z;
        x->pos = position;
        set_face_normal(x, &rr, normal);
    }
    if (xf & 4) {                                                        /* transform.hpp:25-27 */
        x->pos = vadd(x->pos, target);
        set_face_normal(x, &tr, x->normal);
    }
    x->mat = o->mat;
    x->leaf = o;
    return 1;
}

static int obj_hit(const orc_obj* o, const ray* r, isect* x, orc_rng* g);
static int group_hit(orc_obj* const* objs, int n, const ray* r, isect* x, orc_rng* g);

static int bvh_hit(const orc_obj* o, const ray* r, isect* x, orc_rng* g) /* bvh.hpp:58-68 */
{
    if (!aabb_hit(&o->box, r)) return 0;
    int hl = obj_hit(o->l, r, x, g);
    double tmr = hl ? x->t : r->t_max;
    ray rr = mkray(r->o, r->d, r->time, r->t_min, tmr);
    int hr = obj_hit(o->r, &rr, x, g);
    return hl || hr;
}

/* volume (primitive.hpp:329-396): constant-density medium inside `surface` (o->l).  The free-flight distance is drawn
 * INSIDE intersect (:373), so the stream position depends on which volumes a traversal reaches and in what order.
 * g == NULL (the instrumented first-hit walk, which has no stream): a volume is never hit. */
static int volume_hit(const orc_obj* o, const ray* r, isect* x, orc_rng* g)
{
    isect c1, c2; memset(&c1, 0, sizeof c1); memset(&c2, 0, sizeof c2);
    if (!g) return 0;
    ray all = mkray(r->o, r->d, r->time, -ORC_INF, ORC_INF);
    if (!obj_hit(o->l, &all, &c1, g)) return 0;                          /* :349 */
    ray rest = mkray(r->o, r->d, r->time, c1.t + 0.0001, ORC_INF);
    if (!obj_hit(o->l, &rest, &c2, g)) return 0;                         /* :353 */
    if (c1.t < r->t_min) c1.t = r->t_min;
    if (c2.t > r->t_max) c2.t = r->t_max;
    if (c1.t >= c2.t) return 0;
    if (c1.t < 0) c1.t = 0;
    double ray_length = vlen(r->d);
    double inner_length = (c2.t - c1.t) * ray_length;
    double intersect_distance = o->g[4] * LM_log(rnd(g));                  /* :373, g[4] = neg_inv_d */
    if (intersect_distance > inner_length) return 0;
    x->t = c1.t + intersect_distance / ray_length;
    x->pos = ray_at(r, x->t);
    x->normal = V(1, 0, 0);
    x->front = 0;
    x->mat = o->mat;
    x->leaf = o;                                                        /* uv_coord keeps whatever the record held (:379-383) */
    return 1;
}

static int obj_hit(const orc_obj* o, const ray* r, isect* x, orc_rng* g)
{
    switch (o->kind) {
    case ORC_SPHERE: return sphere_hit(o, V(o->g[0], o->g[1], o->g[2]), o->g[3], r, x);
    case ORC_MSPHERE: return sphere_hit(o, msphere_center(o, r->time), o->g[8], r, x);
    case ORC_PLANEXY: case ORC_PLANEXZ: case ORC_PLANEYZ: return plane_hit(o, r, x);
    case ORC_TRIANGLE: return tri_hit(o, r, x);
    case ORC_BOX: return box_hit(o, r, x);
    case ORC_VOLUME: return volume_hit(o, r, x, g);
    case O_FLIP:                                                        /* transform.hpp:129-136 */
        if (!obj_hit(o->l, r, x, g)) return 0;
        x->front = !x->front;
        if (o->id >= 0) x->leaf = o;                                    /* (a flip registered as the primitive; else the leaf found inside) */
        return 1;
    case O_BVH: return bvh_hit(o, r, x, g);
    case O_GROUP: return group_hit(o->items, o->n_items, r, x, g);      /* a group used as an object: group.hpp:68-83 */
    case O_TRANSLATE: {                                                 /* transform.hpp:18-28 */
        const v3 target = V(o->g[0], o->g[1], o->g[2]);
        ray off = mkray(vsub(r->o, target), r->d, r->time, r->t_min, r->t_max);
        if (!obj_hit(o->l, &off, x, g)) return 0;
        x->pos = vadd(x->pos, target);
        set_face_normal(x, &off, x->normal);
        return 1;
    }
    case O_ROTATEY: {                                                   /* transform.hpp:81-110 */
        const double sn = o->g[0], cs = o->g[1];
        v3 origin = r->o, direction = r->d;
        origin.x = cs * r->o.x - sn * r->o.z;
        origin.z = sn * r->o.x + cs * r->o.z;
        direction.x = cs * r->d.x - sn * r->d.z;
        direction.z = sn * r->d.x + cs * r->d.z;
        ray rot = mkray(origin, direction, r->time, r->t_min, r->t_max);
        if (!obj_hit(o->l, &rot, x, g)) return 0;
        v3 position = x->pos, normal = x->normal;
        position.x = cs * x->pos.x + sn * x->pos.z;
        position.z = -sn * x->pos.x + cs * x->pos.z;
        normal.x = cs * x->normal.x + sn * x->normal.z;
        normal.z = -sn * x->normal.x + cs * x->normal.z;
        x->pos = position;
        set_face_normal(x, &rot, normal);
        return 1;
    }
    }
    return 0;
}

static int group_hit(orc_obj* const* objs, int n, const ray* r, isect* x, orc_rng* g)   /* group.hpp:68-83 */
{
    isect tmp; memset(&tmp, 0, sizeof tmp);
    int has = 0;
    double closest = r->t_max;
    for (int i = 0; i < n; i++) {
        ray rr = mkray(r->o, r->d, r->time, r->t_min, closest);
        if (obj_hit(objs[i], &rr, &tmp, g)) { has = 1; closest = tmp.t; *x = tmp; }
    }
    return has;
}

static int obj_bbox(const orc_obj* o, double start, double end, aabb* out)
{
    switch (o->kind) {
    case ORC_SPHERE: {                                                  /* primitive.hpp:58-64 */
        v3 c = V(o->g[0], o->g[1], o->g[2]), rr = V(o->g[3], o->g[3], o->g[3]);
        out->mn = vsub(c, rr); out->mx = vadd(c, rr); return 1;
    }
    case ORC_MSPHERE: {                                                 /* primitive.hpp:132-144 */
        v3 rr = V(o->g[8], o->g[8], o->g[8]);
        aabb a, b;
        a.mn = vsub(msphere_center(o, start), rr); a.mx = vadd(msphere_center(o, start), rr);
        b.mn = vsub(msphere_center(o, end), rr); b.mx = vadd(msphere_center(o, end), rr);
        *out = surrounding(a, b); return 1;
    }
    case ORC_PLANEXY: out->mn = V(o->g[0], o->g[2], o->g[4] - 0.001); out->mx = V(o->g[1], o->g[3], o->g[4] + 0.001); return 1; /* :201-205 */
    case ORC_PLANEXZ: out->mn = V(o->g[0], o->g[4] - 0.001, o->g[2]); out->mx = V(o->g[1], o->g[4] + 0.001, o->g[3]); return 1; /* :252-256 */
    case ORC_PLANEYZ: out->mn = V(o->g[4] - 0.001, o->g[0], o->g[2]); out->mx = V(o->g[4] + 0.001, o->g[1], o->g[3]); return 1; /* :321-325 */
    case ORC_TRIANGLE: {                                                /* polygon.hpp:126-138 */
        const double* v = (const double*) o->l;   /* 9 doubles: p0,p1,p2 (owned) */
        double mn[3], mx[3];
        for (int k = 0; k < 3; k++) {
            double a = v[k], b = v[3 + k], c = v[6 + k];
            double lo = a; if (b < lo) lo = b; if (c < lo) lo = c;
            double hi = a; if (hi < b) hi = b; if (hi < c) hi = c;
            mn[k] = lo - ORC_EPS; mx[k] = hi + ORC_EPS;
        }
        out->mn = V(mn[0], mn[1], mn[2]); out->mx = V(mx[0], mx[1], mx[2]); return 1;
    }
    case ORC_BOX: {                                                     /* primitive.hpp:441-445, transform.hpp:51-79, 30-37 */
        const double* g = o->g;
        const int xf = (int) g[11];
        aabb b; b.mn = V(g[0], g[1], g[2]); b.mx = V(g[3], g[4], g[5]);
        if (xf & 2) {
            double mn[3] = { ORC_INF, ORC_INF, ORC_INF }, mx[3] = { -ORC_INF, -ORC_INF, -ORC_INF };
            for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) {
                double px = i * b.mx.x + (1 - i) * b.mn.x;
                double py = j * b.mx.y + (1 - j) * b.mn.y;
                double pz = k * b.mx.z + (1 - k) * b.mn.z;
                double t[3] = { g[7] * px + g[6] * pz, py, -g[6] * px + g[7] * pz };
                for (int l = 0; l < 3; l++) { mn[l] = fmin(mn[l], t[l]); mx[l] = fmax(mx[l], t[l]); }
            }
            b.mn = V(mn[0], mn[1], mn[2]); b.mx = V(mx[0], mx[1], mx[2]);
        }
        if (xf & 4) { v3 target = V(g[8], g[9], g[10]); b.mn = vadd(b.mn, target); b.mx = vadd(b.mx, target); }
        *out = b;
        return 1;
    }
    case ORC_VOLUME: return obj_bbox(o->l, start, end, out);            /* primitive.hpp:387-390 */
    case O_FLIP: return obj_bbox(o->l, start, end, out);                /* transform.hpp:138-141 */
    case O_BVH: *out = o->box; return 1;                                /* bvh.hpp:70-74 */
    case O_TRANSLATE: {                                                 /* transform.hpp:30-37 */
        if (!obj_bbox(o->l, start, end, out)) return 0;
        v3 target = V(o->g[0], o->g[1], o->g[2]);
        out->mn = vadd(out->mn, target); out->mx = vadd(out->mx, target);
        return 1;
    }
    case O_ROTATEY: *out = o->box; return (int) o->g[2];                /* transform.hpp:112-116 */
    case O_GROUP: {                                                     /* group.hpp:85-102: `first` is never true, so the result is the union of
                                                                           whatever the CALLER's box held (a fresh aabb: zeros) with every member's;
                                                                           temp_bbox lives across the loop, so a member that is itself a group
                                                                           starts from its predecessor's box */
        if (o->n_items == 0) return 0;
        aabb tmp; memset(&tmp, 0, sizeof tmp);
        for (int i = 0; i < o->n_items; i++) {
            if (!obj_bbox(o->items[i], start, end, &tmp)) return 0;
            *out = surrounding(*out, tmp);
        }
        return 1;
    }
    }
    return 0;
}

static double obj_pdf(const orc_obj* o, v3 origin, v3 value)
{
    switch (o->kind) {
    case ORC_PLANEXZ: {                                                 /* primitive.hpp:258-269 */
        isect x; memset(&x, 0, sizeof x);
        ray r = mkray(origin, value, 0, 1e-3, ORC_INF);
        if (!plane_hit(o, &r, &x)) return 0;
        double area = (o->g[1] - o->g[0]) * (o->g[3] - o->g[2]);
        double dist2 = x.t * x.t * vlen(value) * vlen(value);
        double cosine = fabs(vdot(value, x.normal) / vlen(value));
        return dist2 / (cosine * area);
    }
    case ORC_TRIANGLE: {                                                /* polygon.hpp:140-151 */
        isect x; memset(&x, 0, sizeof x);
        ray r = mkray(origin, value, 0, 1e-3, ORC_INF);
        if (!tri_hit(o, &r, &x)) return 0;
        v3 e1 = V(o->g[3], o->g[4], o->g[5]), e2 = V(o->g[6], o->g[7], o->g[8]);
        double area = vlen(vcross(e1, e2)) / 2.0;
        double dist2 = x.t * x.t * vlen(value) * vlen(value);
        double cosine = fabs(vdot(value, x.normal) / vlen(value));
        return dist2 / (cosine * area);
    }
    case O_FLIP: return obj_pdf(o->l, origin, value);                   /* transform.hpp:143 */
    default: return 0.0;                                                /* object.hpp:16 */
    }
}

static v3 obj_rand(const orc_obj* o, orc_rng* g, v3 origin)
{
    switch (o->kind) {
    case ORC_PLANEXZ: {                                                 /* primitive.hpp:270-274 (args right-to-left) */
        double z = rnd_range(g, o->g[2], o->g[3]);
        double x = rnd_range(g, o->g[0], o->g[1]);
        return vsub(V(x, o->g[4], z), origin);
    }
    case ORC_TRIANGLE: {                                                /* polygon.hpp:153-167 */
        v3 p0 = V(o->g[0], o->g[1], o->g[2]), e1 = V(o->g[3], o->g[4], o->g[5]), e2 = V(o->g[6], o->g[7], o->g[8]);
        double u = rnd(g), v = rnd(g);
        v3 p;
        if (u + v > 1) p = vadd(vadd(p0, smul(1 - u, e1)), smul(1 - v, e2));
        else p = vadd(vadd(p0, smul(u, e1)), smul(v, e2));
        return vsub(p, origin);
    }
    case O_FLIP: return obj_rand(o->l, g, origin);                      /* transform.hpp:145 */
    default: return V(1, 0, 0);                                         /* object.hpp:17 */
    }
}

static double lights_pdf(const orc_scene* s, v3 origin, v3 value)       /* group.hpp:104-113 */
{
    double weight = 1.0 / s->n_lights;
    double avg = 0;
    for (int i = 0; i < s->n_lights; i++) avg += weight * obj_pdf(s->lights[i], origin, value);
    return avg;
}
static v3 lights_rand(const orc_scene* s, orc_rng* g, v3 origin)        /* group.hpp:115-119 */
{
    int idx = rnd_int(g, 0, s->n_lights - 1);
    return obj_rand(s->lights[idx], g, origin);
}

/* ------------------------------------------------------------------ bvh build (bvh.hpp:22-56) */
static double bbox_key(const orc_obj* o, int axis)                      /* bvh.hpp:76-83: boxes at time (0,0) */
{
    aabb b; memset(&b, 0, sizeof b);                                   /* a fresh aabb (zeros): a group's box starts from it */
    obj_bbox(o, 0, 0, &b);
    return vget(b.mn, axis);
}

/* glibc 2.35 qsort = stable merge sort taking the left run when cmp <= 0; the comparator (bvh.hpp:100-113)
 * returns +1 when a.min < b.min else -1  =>  stable DESCENDING order by min[axis]. */
static void merge_sort_desc(orc_obj** a, double* key, int n, orc_obj** ta, double* tk)
{
    if (n <= 1) return;
    int n1 = n / 2, n2 = n - n1;
    merge_sort_desc(a, key, n1, ta, tk);
    merge_sort_desc(a + n1, key + n1, n2, ta, tk);
    int i = 0, j = n1, k = 0;
    while (i < n1 && j < n) {
        int cmp = (key[i] < key[j]) ? 1 : -1;
        if (cmp <= 0) { ta[k] = a[i]; tk[k++] = key[i++]; }
        else { ta[k] = a[j]; tk[k++] = key[j++]; }
    }
    while (i < n1) { ta[k] = a[i]; tk[k++] = key[i++]; }
    while (j < n) { ta[k] = a[j]; tk[k++] = key[j++]; }
    memcpy(a, ta, sizeof(orc_obj*) * n);
    memcpy(key, tk, sizeof(double) * n);
}

static orc_obj* new_obj(int kind)
{
    orc_obj* o = (orc_obj*) calloc(1, sizeof(orc_obj));
    o->kind = kind; o->id = -1; o->mat = -1; o->first_entry = -1;
    return o;
}

static orc_obj* bvh_build(orc_rng* g, orc_obj** objs, int first, int last, double start, double end)
{
    orc_obj* n = new_obj(O_BVH);
    int axis = rnd_int(g, 0, 2);                                        /* one draw per node, before the span test */
    int span = last - first;
    if (span == 1) { n->l = n->r = objs[first]; }
    else if (span == 2) {
        if (bbox_key(objs[first], axis) < bbox_key(objs[first + 1], axis)) { n->l = objs[first]; n->r = objs[first + 1]; }
        else { n->l = objs[first + 1]; n->r = objs[first]; }
    }
    else {
        double* key = (double*) malloc(sizeof(double) * span * 2);
        orc_obj** tmp = (orc_obj**) malloc(sizeof(orc_obj*) * span);
        for (int i = 0; i < span; i++) key[i] = bbox_key(objs[first + i], axis);
        merge_sort_desc(objs + first, key, span, tmp, key + span);
        free(key); free(tmp);
        int middle = first + span / 2;
        n->l = bvh_build(g, objs, first, middle, start, end);
        n->r = bvh_build(g, objs, middle, last, start, end);
    }
    aabb bl, br;
    memset(&bl, 0, sizeof bl); memset(&br, 0, sizeof br);
    obj_bbox(n->l, start, end, &bl);
    obj_bbox(n->r, start, end, &br);
    n->box = surrounding(bl, br);
    return n;
}

/* ------------------------------------------------------------------ scene construction */
static void push(orc_obj*** arr, int* n, int* cap, orc_obj* o)
{
    if (*n == *cap) { *cap = *cap ? *cap * 2 : 64; *arr = (orc_obj**) realloc(*arr, sizeof(orc_obj*) * *cap); }
    (*arr)[(*n)++] = o;
}

static void setup_camera(orc_scene* s)                                  /* camera.hpp:46-60 */
{
    double arc = s->fov * ORC_PI / 180;                                 /* util.hpp:24-27 */
    double sensor_h = tan(arc / 2) * 2;
    double sensor_w = sensor_h * s->aspect;
    s->cw = vnorm(vsub(s->cam_pos, s->cam_lookat));
    s->cu = vnorm(vcross(s->cam_up, s->cw));
    s->cv = vcross(s->cw, s->cu);
    s->horizontal = smul(s->focus * sensor_w, s->cu);
    s->vertical = smul(s->focus * sensor_h, s->cv);
    s->left_bottom = vsub(vsub(vsub(s->cam_pos, sdiv(s->horizontal, 2)), sdiv(s->vertical, 2)), smul(s->focus, s->cw));
    s->lens_radius = s->aperture / 2;
}

orc_scene* orc_scene_new(int width, int height, unsigned seed)
{
    orc_scene* s = (orc_scene*) calloc(1, sizeof(orc_scene));
    s->width = width; s->height = height;
    s->rng.mode = ORC_RNG_GLIBC;
    glibc_seed(&s->rng, seed);
    s->aspect = (double) width / height;
    s->focus = 1;   /* camera.hpp:29 default */
    return s;
}

void orc_scene_free(orc_scene* s)
{
    /* objects are small and scenes short-lived in tests; nodes are leaked like the reference leaks them,
     * except for the big arrays */
    free(s->objects); free(s->lights); free(s->prims); free(s->mats); free(s->texs); free(s->pixels);
    free(s->root_entry); free(s->pre); free(s->pre_first); free(s->pre_left); free(s->pre_right);
    free(s);
}

static int add_tex(orc_scene* s, orc_tex t)
{
    s->texs = (orc_tex*) realloc(s->texs, sizeof(orc_tex) * (s->n_texs + 1));
    s->texs[s->n_texs] = t;
    return s->n_texs++;
}
static int add_mat(orc_scene* s, orc_mat m)
{
    s->mats = (orc_mat*) realloc(s->mats, sizeof(orc_mat) * (s->n_mats + 1));
    s->mats[s->n_mats] = m;
    return s->n_mats++;
}
int orc_tex_constant(orc_scene* s, double r, double g, double b)
{ orc_tex t; memset(&t, 0, sizeof t); t.kind = ORC_TEX_CONSTANT; t.d[0] = r; t.d[1] = g; t.d[2] = b; return add_tex(s, t); }
int orc_tex_checker(orc_scene* s, int even_tex, int odd_tex)
{ orc_tex t; memset(&t, 0, sizeof t); t.kind = ORC_TEX_CHECKER; t.d[0] = even_tex; t.d[1] = odd_tex; return add_tex(s, t); }
int orc_tex_image(orc_scene* s, const uint8_t* pixels, int w, int h, int channels, int color_based)
{
    orc_tex t; memset(&t, 0, sizeof t); t.kind = ORC_TEX_IMAGE;
    t.d[0] = w; t.d[1] = h; t.d[2] = channels; t.d[3] = color_based; t.d[4] = s->n_pixel_bytes;
    int nb = w * h * channels;
    if (pixels && nb > 0) {
        s->pixels = (uint8_t*) realloc(s->pixels, s->n_pixel_bytes + nb);
        memcpy(s->pixels + s->n_pixel_bytes, pixels, nb);
        t.pix_off = s->n_pixel_bytes;
        s->n_pixel_bytes += nb;
        t.pixels = (const uint8_t*) 1;   /* resolved after all reallocs, see fix_pixels */
    }
    return add_tex(s, t);
}
static void fix_pixels(orc_scene* s)
{ for (int i = 0; i < s->n_texs; i++) if (s->texs[i].kind == ORC_TEX_IMAGE && s->texs[i].pixels) s->texs[i].pixels = s->pixels + s->texs[i].pix_off; }

int orc_mat_lambertian(orc_scene* s, int tex) { orc_mat m; memset(&m, 0, sizeof m); m.kind = ORC_MAT_LAMBERTIAN; m.tex = tex; return add_mat(s, m); }
int orc_mat_metal(orc_scene* s, int tex, double roughness)
{ orc_mat m; memset(&m, 0, sizeof m); m.kind = ORC_MAT_METAL; m.tex = tex; m.d[0] = roughness < 1 ? roughness : 1; return add_mat(s, m); } /* material.hpp:65 */
int orc_mat_dielectric(orc_scene* s, double ior) { orc_mat m; memset(&m, 0, sizeof m); m.kind = ORC_MAT_DIELECTRIC; m.tex = -1; m.d[0] = ior; return add_mat(s, m); }
int orc_mat_fog(orc_scene* s, int tex) { orc_mat m; memset(&m, 0, sizeof m); m.kind = ORC_MAT_FOG; m.tex = tex; return add_mat(s, m); }
int orc_mat_emissive(orc_scene* s, int tex) { orc_mat m; memset(&m, 0, sizeof m); m.kind = ORC_MAT_EMISSIVE; m.tex = tex; return add_mat(s, m); }
int orc_mat_phong(orc_scene* s, const double* k, double ns, double ni, int diffuse_tex)
{
    orc_mat m; memset(&m, 0, sizeof m); m.kind = ORC_MAT_PHONG; m.tex = diffuse_tex;
    for (int i = 0; i < 12; i++) m.d[1 + i] = k[i];
    m.d[13] = (double) (float) ns; m.d[14] = (double) (float) ni;       /* material.hpp:191-192: float members */
    return add_mat(s, m);
}

static int finish_add(orc_scene* s, orc_obj* o, int flip, int to_lights)
{
    if (flip) { orc_obj* f = new_obj(O_FLIP); f->l = o; f->mat = o->mat; o = f; }
    if (to_lights) { push(&s->lights, &s->n_lights, &s->cap_lights, o); return s->n_lights - 1; }
    o->id = s->n_prims;
    if (o->kind == O_FLIP) o->l->id = o->id;
    push(&s->prims, &s->n_prims, &s->cap_prims, o);
    push(&s->objects, &s->n_objects, &s->cap_objects, o);
    return o->id;
}

int orc_add_sphere(orc_scene* s, const double c[3], double radius, int mat, int flip, int to_lights)
{ orc_obj* o = new_obj(ORC_SPHERE); o->mat = mat; o->g[0] = c[0]; o->g[1] = c[1]; o->g[2] = c[2]; o->g[3] = radius; return finish_add(s, o, flip, to_lights); }
int orc_add_msphere(orc_scene* s, const double c0[3], const double c1[3], double t0, double t1, double radius, int mat, int flip, int to_lights)
{
    orc_obj* o = new_obj(ORC_MSPHERE); o->mat = mat;
    for (int k = 0; k < 3; k++) { o->g[k] = c0[k]; o->g[3 + k] = c1[k]; }
    o->g[6] = t0; o->g[7] = t1; o->g[8] = radius;
    return finish_add(s, o, flip, to_lights);
}
int orc_add_plane(orc_scene* s, int kind, double a0, double a1, double b0, double b1, double k, int mat, int flip, int to_lights)
{ orc_obj* o = new_obj(kind); o->mat = mat; o->g[0] = a0; o->g[1] = a1; o->g[2] = b0; o->g[3] = b1; o->g[4] = k; return finish_add(s, o, flip, to_lights); }

/* box(minimum, maximum, mat), optionally wrapped rotatey(., angle) then translate(., target); sin/cos as the rotatey ctor
 * computes them (transform.hpp:50-52: sin/cos of angle * pi / 180 with util.hpp's truncated pi) */
int orc_add_box(orc_scene* s, const double mn[3], const double mx[3], int wrappers, double sin_theta, double cos_theta, const double target[3],
                int mat, int flip, int to_lights)
{
    orc_obj* o = new_obj(ORC_BOX); o->mat = mat;
    for (int k = 0; k < 3; k++) { o->g[k] = mn[k]; o->g[3 + k] = mx[k]; o->g[8 + k] = (wrappers & 4) ? target[k] : 0.0; }
    o->g[6] = (wrappers & 2) ? sin_theta : 0.0; o->g[7] = (wrappers & 2) ? cos_theta : 1.0; o->g[11] = wrappers & 6;
    return finish_add(s, o, flip, to_lights);
}

/* volume(new sphere(center, radius, <any material>), density, fog): the boundary's own material is never used */
int orc_add_volume(orc_scene* s, const double c[3], double radius, double density, int fog_mat, int to_lights)
{
    orc_obj* b = new_obj(ORC_SPHERE); b->g[0] = c[0]; b->g[1] = c[1]; b->g[2] = c[2]; b->g[3] = radius;
    orc_obj* o = new_obj(ORC_VOLUME); o->mat = fog_mat; o->l = b;
    for (int k = 0; k < 4; k++) o->g[k] = b->g[k];
    o->g[4] = -1.0 / density;                                           /* primitive.hpp:336 */
    return finish_add(s, o, 0, to_lights);
}

static orc_obj* mk_triangle(const double p[9], const double n[9], const double uv[6], int mat)   /* polygon.hpp:26-38 */
{
    orc_obj* o = new_obj(ORC_TRIANGLE); o->mat = mat;
    double* keep = (double*) malloc(sizeof(double) * 9);
    memcpy(keep, p, sizeof(double) * 9);
    o->l = (orc_obj*) keep;                     /* vertex positions, for bounding_box */
    for (int k = 0; k < 3; k++) { o->g[k] = p[k]; o->g[3 + k] = p[3 + k] - p[k]; o->g[6 + k] = p[6 + k] - p[k]; }
    for (int k = 0; k < 9; k++) o->g[9 + k] = n[k];
    for (int k = 0; k < 6; k++) o->g[18 + k] = uv[k];
    return o;
}
int orc_add_triangle_v(orc_scene* s, const double p[9], const double n[9], const double uv[6], int mat, int flip, int to_lights)
{ return finish_add(s, mk_triangle(p, n, uv, mat), flip, to_lights); }
static orc_obj* mk_triangle_p(const double p[9], int mat)                                        /* polygon.hpp:54-80 */
{
    v3 p0 = V(p[0], p[1], p[2]), p1 = V(p[3], p[4], p[5]), p2 = V(p[6], p[7], p[8]);
    v3 n0 = vnorm(vcross(vsub(p1, p0), vsub(p2, p0)));
    v3 n1 = vnorm(vcross(vsub(p2, p1), vsub(p0, p1)));
    v3 n2 = vnorm(vcross(vsub(p0, p2), vsub(p1, p2)));
    double n[9] = { n0.x, n0.y, n0.z, n1.x, n1.y, n1.z, n2.x, n2.y, n2.z };
    double uv[6] = { p0.x, p0.y, p1.x, p1.y, p2.x, p2.y };
    return mk_triangle(p, n, uv, mat);
}
int orc_add_triangle_p(orc_scene* s, const double p[9], int mat, int flip, int to_lights)
{ return finish_add(s, mk_triangle_p(p, mat), flip, to_lights); }
/* bulk add from flattened records (layout of include/rtcore.h): used by bench.py's cpu_baseline leg to time the oracle
 * on a scene the host mirror built.  Triangle corners are rebuilt as p0, p0+e1, p0+e2 (bounding boxes may differ from
 * the loader's by an ulp: good enough for a timing baseline, not used by the parity tests). */
void orc_add_prims_bulk(orc_scene* s, int n, const int32_t* kind, const int32_t* flags, const int32_t* mat, const double* data, int to_lights)
{
    for (int i = 0; i < n; i++) {
        const double* g = data + (size_t) i * ORC_PRIM_STRIDE;
        int fl = flags[i] & 1;
        switch (kind[i]) {
        case ORC_SPHERE: orc_add_sphere(s, g, g[3], mat[i], fl, to_lights); break;
        case ORC_MSPHERE: orc_add_msphere(s, g, g + 3, g[6], g[7], g[8], mat[i], fl, to_lights); break;
        case ORC_BOX: orc_add_box(s, g, g + 3, flags[i] & 6, g[6], g[7], g + 8, mat[i], fl, to_lights); break;
        case ORC_VOLUME: { int id = orc_add_volume(s, g, g[3], 1.0, mat[i], to_lights); orc_obj* o = to_lights ? s->lights[id] : s->prims[id]; o->g[4] = g[4]; break; }
        case ORC_TRIANGLE: {
            double p[9];
            for (int k = 0; k < 3; k++) { p[k] = g[k]; p[3 + k] = g[k] + g[3 + k]; p[6 + k] = g[k] + g[6 + k]; }
            orc_obj* o = mk_triangle(p, g + 9, g + 18, mat[i]);
            for (int k = 0; k < 6; k++) o->g[3 + k] = g[3 + k];
            finish_add(s, o, fl, to_lights);
            break;
        }
        default: orc_add_plane(s, kind[i], g[0], g[1], g[2], g[3], g[4], mat[i], fl, to_lights); break;
        }
    }
}

int orc_add_light_alias(orc_scene* s, int prim_id)
{ push(&s->lights, &s->n_lights, &s->cap_lights, s->prims[prim_id]); return s->n_lights - 1; }

void orc_set_camera(orc_scene* s, const double pos[3], const double lookat[3], const double up[3], double fov,
                    double aperture, double focus, double shutter0, double shutter1)
{
    s->cam_pos = V(pos[0], pos[1], pos[2]); s->cam_lookat = V(lookat[0], lookat[1], lookat[2]); s->cam_up = V(up[0], up[1], up[2]);
    s->fov = fov; s->aperture = aperture; s->focus = focus; s->shutter0 = shutter0; s->shutter1 = shutter1;
    s->aspect = (double) s->width / s->height;                           /* scene.hpp:57-62 overrides the user's aspect */
    setup_camera(s);
}

void orc_set_camera_fields(orc_scene* s, const double c[24])    /* derived fields as exported by orc_camera_fields */
{
    s->cam_pos = V(c[0], c[1], c[2]); s->left_bottom = V(c[3], c[4], c[5]); s->horizontal = V(c[6], c[7], c[8]); s->vertical = V(c[9], c[10], c[11]);
    s->cu = V(c[12], c[13], c[14]); s->cv = V(c[15], c[16], c[17]); s->cw = V(c[18], c[19], c[20]);
    s->lens_radius = c[21]; s->shutter0 = c[22]; s->shutter1 = c[23];
}

void orc_build_bvh_range(orc_scene* s, double start, double end)          /* bvh(grp, start, end) as demo.hpp:58 */
{
    orc_obj* root = bvh_build(&s->rng, s->objects, 0, s->n_objects, start, end);
    s->objects[0] = root;
    s->n_objects = 1;
    fix_pixels(s);
}

void orc_build_bvh(orc_scene* s)                                        /* scene.hpp:46-51: bvh(objects) with start=end=0 */
{
    orc_obj* root = bvh_build(&s->rng, s->objects, 0, s->n_objects, 0, 0);
    s->objects[0] = root;
    s->n_objects = 1;
    fix_pixels(s);
}

/* demo.hpp:17-60 random_scene(): consumes the scene rng exactly like the reference, then wraps everything in bvh(grp,0,1) */
static void random_scene(orc_scene* s)
{
    orc_rng* g = &s->rng;
    int chk = orc_tex_checker(s, orc_tex_constant(s, 0, 0, 0), orc_tex_constant(s, 1, 1, 1));
    double c[3] = { 0, -1000, 0 };
    orc_add_sphere(s, c, 1000, orc_mat_lambertian(s, chk), 0, 0);
    for (int i = -11; i < 11; i++) {
        for (int j = -11; j < 11; j++) {
            double choose = rnd(g);
            /* vec3d(i + 0.9*rand, 0.2, j + 0.9*rand): last argument draws first */
            double cz = j + 0.9 * rnd(g);
            double cx = i + 0.9 * rnd(g);
            v3 center = V(cx, 0.2, cz);
            if (vlen(vsub(center, V(4, 0.2, 0))) > 0.9) {
                double cc[3] = { center.x, center.y, center.z };
                if (choose < 0.8) {
                    /* random_vector() * random_vector(): operands of operator* -- right operand first (pinned by fixture) */
                    v3 b = rnd_vec01(g);
                    v3 a = rnd_vec01(g);
                    v3 albedo = vmul(a, b);
                    int m = orc_mat_lambertian(s, orc_tex_constant(s, albedo.x, albedo.y, albedo.z));
                    v3 c2 = vadd(center, V(0, rnd_range(g, 0, 0.5), 0));
                    double cc2[3] = { c2.x, c2.y, c2.z };
                    orc_add_msphere(s, cc, cc2, 0, 1, 0.2, m, 0, 0);
                }
                else if (choose < 0.95) {
                    v3 albedo = rnd_vec(g, 0.5, 1);
                    double rough = rnd_range(g, 0, 0.5);
                    int m = orc_mat_metal(s, orc_tex_constant(s, albedo.x, albedo.y, albedo.z), rough);
                    orc_add_sphere(s, cc, 0.2, m, 0, 0);
                }
                else {
                    orc_add_sphere(s, cc, 0.2, orc_mat_dielectric(s, 1.5), 0, 0);
                }
            }
        }
    }
    double a[3] = { 0, 1, 0 }, b[3] = { -4, 1, 0 }, d[3] = { 4, 1, 0 };
    orc_add_sphere(s, a, 1.0, orc_mat_dielectric(s, 1.5), 0, 0);
    orc_add_sphere(s, b, 1.0, orc_mat_lambertian(s, orc_tex_constant(s, 0.4, 0.2, 0.1)), 0, 0);
    orc_add_sphere(s, d, 1.0, orc_mat_metal(s, orc_tex_constant(s, 0.7, 0.6, 0.5), 0), 0, 0);
    orc_obj* root = bvh_build(g, s->objects, 0, s->n_objects, 0, 1);   /* demo.hpp:58 */
    s->objects[0] = root;
    s->n_objects = 1;
}

/* test / bench plumbing, no counterpart in the reference: another image size over the same scene and the same (already derived)
 * camera fields -- the caller keeps the aspect ratio -- so one scene serves a full-size timing run and a small parity render */
void orc_set_size(orc_scene* s, int width, int height) { s->width = width; s->height = height; }

orc_scene* orc_build_c0(int width, int height, unsigned seed)
{
    orc_scene* s = orc_scene_new(width, height, seed);
    random_scene(s);
    int em = orc_mat_emissive(s, orc_tex_constant(s, 4, 4, 4));
    orc_add_plane(s, ORC_PLANEXZ, -20, 20, -20, 20, 30, em, 1, 0);
    int em2 = orc_mat_emissive(s, orc_tex_constant(s, 4, 4, 4));
    orc_add_plane(s, ORC_PLANEXZ, -20, 20, -20, 20, 30, em2, 1, 1);
    double pos[3] = { 13, 2, 3 }, at[3] = { 0, 0, 0 }, up[3] = { 0, 1, 0 };
    orc_set_camera(s, pos, at, up, 20, 0.1, 10, 0, 1);
    orc_build_bvh(s);
    return s;
}

orc_scene* orc_build_mixed(int width, int height, unsigned seed)
{
    orc_scene* s = orc_scene_new(width, height, seed);
    int chk = orc_tex_checker(s, orc_tex_constant(s, 0.2, 0.3, 0.1), orc_tex_constant(s, 0.9, 0.9, 0.9));
    double c0[3] = { 0, -100.5, -1 }, c1[3] = { 0, 0, -1 }, c2[3] = { -1.05, 0, -1 }, c3[3] = { 1.05, 0, -1 };
    double m0[3] = { 0.3, 0.7, -0.6 }, m1[3] = { 0.3, 0.9, -0.6 }, c5[3] = { -0.4, -0.3, -0.2 };
    orc_add_sphere(s, c0, 100, orc_mat_lambertian(s, chk), 0, 0);
    orc_add_sphere(s, c1, 0.5, orc_mat_lambertian(s, orc_tex_constant(s, 0.1, 0.2, 0.5)), 0, 0);
    orc_add_sphere(s, c2, 0.5, orc_mat_dielectric(s, 1.5), 0, 0);
    orc_add_sphere(s, c3, 0.5, orc_mat_metal(s, orc_tex_constant(s, 0.8, 0.6, 0.2), 0.3), 0, 0);
    orc_add_msphere(s, m0, m1, 0, 1, 0.15, orc_mat_lambertian(s, orc_tex_constant(s, 0.7, 0.2, 0.2)), 0, 0);
    orc_add_sphere(s, c5, 0.2, orc_mat_metal(s, orc_tex_constant(s, 0.9, 0.9, 0.9), 0.0), 0, 0);
    double t0[9] = { -2, -0.5, -2.5, 2, -0.5, -2.5, 0, 2.0, -2.8 };
    double t1[9] = { 1.2, -0.5, 0.2, 1.9, -0.5, -0.4, 1.5, 0.6, -0.1 };
    double t2[9] = { -1.9, -0.5, 0.3, -1.2, -0.5, 0.1, -1.6, 0.5, 0.0 };
    orc_add_triangle_p(s, t0, orc_mat_lambertian(s, orc_tex_constant(s, 0.3, 0.7, 0.3)), 0, 0);
    orc_add_triangle_p(s, t1, orc_mat_metal(s, orc_tex_constant(s, 0.7, 0.7, 0.9), 0.1), 0, 0);
    double ph[12] = { 0.6, 0.5, 0.2, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    orc_add_triangle_p(s, t2, orc_mat_phong(s, ph, 0, 0, -1), 0, 0);
    orc_add_plane(s, ORC_PLANEXY, -3, 3, -0.5, 3, -3.2, orc_mat_lambertian(s, orc_tex_constant(s, 0.6, 0.6, 0.7)), 0, 0);
    orc_add_plane(s, ORC_PLANEYZ, -0.5, 3, -3.2, 1, -3, orc_mat_lambertian(s, orc_tex_constant(s, 0.7, 0.3, 0.3)), 0, 0);
    orc_add_plane(s, ORC_PLANEXZ, -1, 1, -2, 0, 3, orc_mat_emissive(s, orc_tex_constant(s, 6, 6, 6)), 1, 0);
    double lamp[12] = { 0, 0, 0, 0, 0, 0, 5, 4, 3, 0, 0, 0 };
    double t3[9] = { 2.5, 2.5, -1.5, 2.5, 0.5, -0.5, 2.5, 0.5, -2.5 };
    int lamp_id = orc_add_triangle_p(s, t3, orc_mat_phong(s, lamp, 0, 0, -1), 0, 0);
    orc_add_plane(s, ORC_PLANEXZ, -1, 1, -2, 0, 3, orc_mat_emissive(s, orc_tex_constant(s, 6, 6, 6)), 1, 1);
    orc_add_light_alias(s, lamp_id);
    double pos[3] = { -0.5, 1.2, 3.5 }, at[3] = { 0, 0.2, -1 }, up[3] = { 0, 1, 0 };
    orc_set_camera(s, pos, at, up, 45, 0.05, 4.5, 0, 1);
    orc_build_bvh(s);
    return s;
}

/* the reference's own main() scene: raytracing.cpp:77-88 over cornell_box() (demo.hpp:63-91) */
orc_scene* orc_build_cornell(int width, int height, unsigned seed)
{
    orc_scene* s = orc_scene_new(width, height, seed);
    int red = orc_mat_lambertian(s, orc_tex_constant(s, 0.65, 0.05, 0.05));
    int green = orc_mat_lambertian(s, orc_tex_constant(s, 0.12, 0.45, 0.15));
    int white = orc_mat_lambertian(s, orc_tex_constant(s, 0.75, 0.75, 0.75));
    int light = orc_mat_emissive(s, orc_tex_constant(s, 30, 30, 30));
    int mirror = orc_mat_metal(s, orc_tex_constant(s, 1, 1, 1), 0);
    orc_add_plane(s, ORC_PLANEYZ, 0, 555, 0, 555, 555, green, 0, 0);
    orc_add_plane(s, ORC_PLANEYZ, 0, 555, 0, 555, 0, red, 0, 0);
    orc_add_plane(s, ORC_PLANEXZ, 0, 555, 0, 555, 0, white, 0, 0);
    orc_add_plane(s, ORC_PLANEXZ, 0, 555, 0, 555, 555, white, 0, 0);
    orc_add_plane(s, ORC_PLANEXY, 0, 555, 0, 555, 555, white, 0, 0);
    orc_add_plane(s, ORC_PLANEXZ, 213, 343, 227, 332, 554, light, 1, 0);
    double zero[3] = { 0, 0, 0 }, b1[3] = { 165, 330, 165 }, b2[3] = { 165, 165, 165 }, t1[3] = { 265, 0, 295 }, t2[3] = { 130, 0, 65 };
    double a1 = 15 * ORC_PI / 180, a2 = -18 * ORC_PI / 180;               /* util.hpp:24-27 */
    orc_add_box(s, zero, b1, 6, sin(a1), cos(a1), t1, mirror, 0, 0);
    orc_add_box(s, zero, b2, 6, sin(a2), cos(a2), t2, white, 0, 0);
    int light2 = orc_mat_emissive(s, orc_tex_constant(s, 30, 30, 30));
    orc_add_plane(s, ORC_PLANEXZ, 213, 343, 227, 332, 554, light2, 1, 1);
    double pos[3] = { 278, 278, -800 }, at[3] = { 278, 278, 0 }, up[3] = { 0, 1, 0 };
    orc_set_camera(s, pos, at, up, 40, 0, 1, 0, 1);
    orc_build_bvh(s);
    return s;
}

static orc_obj* mk_translate(orc_obj* inner, v3 target)
{ orc_obj* o = new_obj(O_TRANSLATE); o->l = inner; o->g[0] = target.x; o->g[1] = target.y; o->g[2] = target.z; return o; }
static orc_obj* mk_rotatey(orc_obj* inner, double angle)                /* transform.hpp:47-79 */
{
    orc_obj* o = new_obj(O_ROTATEY); o->l = inner;
    double arc = angle * ORC_PI / 180;                                  /* util.hpp:24-27 */
    o->g[0] = sin(arc); o->g[1] = cos(arc);
    aabb b; memset(&b, 0, sizeof b);
    o->g[2] = obj_bbox(inner, 0, 1, &b);
    double mn[3] = { ORC_INF, ORC_INF, ORC_INF }, mx[3] = { -ORC_INF, -ORC_INF, -ORC_INF };
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++) {
        double x = i * b.mx.x + (1 - i) * b.mn.x;
        double y = j * b.mx.y + (1 - j) * b.mn.y;
        double z = k * b.mx.z + (1 - k) * b.mn.z;
        double t[3] = { o->g[1] * x + o->g[0] * z, y, -o->g[0] * x + o->g[1] * z };
        for (int l = 0; l < 3; l++) { mn[l] = fmin(mn[l], t[l]); mx[l] = fmax(mx[l], t[l]); }
    }
    o->box.mn = V(mn[0], mn[1], mn[2]); o->box.mx = V(mx[0], mx[1], mx[2]);
    return o;
}
/* a leaf that gets a primitive id but is not (yet) a member of scene.objects: the contents of a nested bvh */
static orc_obj* detached_sphere(orc_scene* s, v3 c, double radius, int mat)
{
    orc_obj* o = new_obj(ORC_SPHERE); o->mat = mat; o->g[0] = c.x; o->g[1] = c.y; o->g[2] = c.z; o->g[3] = radius;
    o->id = s->n_prims;
    push(&s->prims, &s->n_prims, &s->cap_prims, o);
    return o;
}

/* all_feature_test() (demo.hpp:94-143) under a light and a camera (the reference never renders it: raytracing.cpp:84 picks
 * cornell_box()).  As there: 400 ground boxes, rect light (not flipped: it shines upwards), moving / glass / fuzzy-metal
 * spheres, haze and blue-fog volumes, a sphere whose image texture fails to load (cyan, texture.hpp:97-99), a mirror sphere,
 * and translate(rotatey(bvh(1000 spheres))) -- all inside bvh(grp, 0, 1).  Same scene as ref_harness.cpp build_features. */
static orc_scene* build_features(int width, int height, unsigned seed, int lit)
{
    orc_scene* s = orc_scene_new(width, height, seed);
    orc_rng* g = &s->rng;
    int white = orc_mat_lambertian(s, orc_tex_constant(s, 0.73, 0.73, 0.73));
    int ground = orc_mat_lambertian(s, orc_tex_constant(s, 0.48, 0.83, 0.53));
    int light = orc_mat_emissive(s, orc_tex_constant(s, 7, 7, 7));
    double zero3[3] = { 0, 0, 0 };
    for (int i = 0; i < 20; i++) {
        for (int j = 0; j < 20; j++) {
            double w = 100, x0 = -1000 + i * w, z0 = -1000 + j * w, y0 = 0, x1 = x0 + w, z1 = z0 + w;
            double y1 = rnd_range(g, 1, 101);
            double mn[3] = { x0, y0, z0 }, mx[3] = { x1, y1, z1 };
            orc_add_box(s, mn, mx, 0, 0, 1, zero3, ground, 0, 0);
        }
    }
    orc_add_plane(s, ORC_PLANEXZ, 123, 423, 147, 412, 554, light, 0, 0);
    double m0[3] = { 400, 400, 200 }, m1[3] = { 500, 400, 200 };
    orc_add_msphere(s, m0, m1, 0, 1, 50, orc_mat_lambertian(s, orc_tex_constant(s, 0.7, 0.3, 0.1)), 0, 0);
    double c1[3] = { 260, 150, 45 }, c2[3] = { 0, 150, 145 }, c3[3] = { 360, 150, 145 }, c4[3] = { 400, 200, 400 }, c5[3] = { 220, 280, 300 };
    orc_add_sphere(s, c1, 50, orc_mat_dielectric(s, 1.5), 0, 0);
    orc_add_sphere(s, c2, 50, orc_mat_metal(s, orc_tex_constant(s, 0.8, 0.8, 0.9), 1), 0, 0);
    orc_add_volume(s, zero3, 5000, 0.0002, orc_mat_fog(s, orc_tex_constant(s, 1, 1, 1)), 0);
    orc_add_sphere(s, c3, 70, orc_mat_dielectric(s, 1.5), 0, 0);
    orc_add_volume(s, c3, 70, 0.2, orc_mat_fog(s, orc_tex_constant(s, 0.2, 0.4, 0.9)), 0);
    orc_add_sphere(s, c4, 100, orc_mat_lambertian(s, orc_tex_image(s, NULL, 0, 0, 0, 0)), 0, 0);
    orc_add_sphere(s, c5, 50, orc_mat_metal(s, orc_tex_constant(s, 1, 1, 1), 0), 0, 0);
    orc_obj** cloud = (orc_obj**) malloc(sizeof(orc_obj*) * 1000);
    for (int i = 0; i < 1000; i++) cloud[i] = detached_sphere(s, rnd_vec(g, 0, 165), 10, white);
    orc_obj* nested = bvh_build(g, cloud, 0, 1000, 0, 0);               /* new bvh(vol_boxes): start = end = 0 */
    free(cloud);
    orc_obj* inst = mk_translate(mk_rotatey(nested, 15), V(-100, 270, 395));
    push(&s->objects, &s->n_objects, &s->cap_objects, inst);
    orc_obj* root = bvh_build(g, s->objects, 0, s->n_objects, 0, 1);   /* demo.hpp:141 */
    s->objects[0] = root;
    s->n_objects = 1;
    if (lit) {   /* the demo's own lamp faces up and the scene renders black: a flipped rect just under it, as an object and as the light */
        orc_add_plane(s, ORC_PLANEXZ, 123, 423, 147, 412, 553, orc_mat_emissive(s, orc_tex_constant(s, 7, 7, 7)), 1, 0);
        orc_add_plane(s, ORC_PLANEXZ, 123, 423, 147, 412, 553, orc_mat_emissive(s, orc_tex_constant(s, 7, 7, 7)), 1, 1);
    }
    else orc_add_plane(s, ORC_PLANEXZ, 123, 423, 147, 412, 554, orc_mat_emissive(s, orc_tex_constant(s, 7, 7, 7)), 0, 1);
    double pos[3] = { 478, 278, -600 }, at[3] = { 278, 278, 0 }, up[3] = { 0, 1, 0 };
    orc_set_camera(s, pos, at, up, 40, 0, 10, 0, 1);
    orc_build_bvh(s);
    return s;
}
orc_scene* orc_build_features(int width, int height, unsigned seed) { return build_features(width, height, seed, 0); }
/* ... with a light that shines downwards (no counterpart in the reference; raytracing_amd/host rth_build_features_lit builds the same) */
orc_scene* orc_build_features_lit(int width, int height, unsigned seed) { return build_features(width, height, seed, 1); }

/* same scene as ref_harness.cpp build_smoke: sphere-bounded media (primitive.hpp:329-396) with fog (material.hpp:165-181) */
orc_scene* orc_build_smoke(int width, int height, unsigned seed)
{
    orc_scene* s = orc_scene_new(width, height, seed);
    orc_add_plane(s, ORC_PLANEXZ, -1000, 1000, -1000, 1000, 0, orc_mat_lambertian(s, orc_tex_constant(s, 0.48, 0.83, 0.53)), 0, 0);
    orc_add_plane(s, ORC_PLANEXZ, 123, 423, 147, 412, 554, orc_mat_emissive(s, orc_tex_constant(s, 7, 7, 7)), 1, 0);
    double m0[3] = { 400, 400, 200 }, m1[3] = { 430, 400, 200 }, c1[3] = { 260, 150, 45 }, c2[3] = { 0, 150, 145 }, zero[3] = { 0, 0, 0 };
    double c3[3] = { 360, 150, 145 }, c4[3] = { 400, 200, 400 }, c5[3] = { 220, 280, 300 }, c6[3] = { 150, 100, 250 };
    orc_add_msphere(s, m0, m1, 0, 1, 50, orc_mat_lambertian(s, orc_tex_constant(s, 0.7, 0.3, 0.1)), 0, 0);
    orc_add_sphere(s, c1, 50, orc_mat_dielectric(s, 1.5), 0, 0);
    orc_add_sphere(s, c2, 50, orc_mat_metal(s, orc_tex_constant(s, 0.8, 0.8, 0.9), 1), 0, 0);
    orc_add_volume(s, zero, 5000, 0.0002, orc_mat_fog(s, orc_tex_constant(s, 1, 1, 1)), 0);
    orc_add_sphere(s, c3, 70, orc_mat_dielectric(s, 1.5), 0, 0);
    orc_add_volume(s, c3, 70, 0.2, orc_mat_fog(s, orc_tex_constant(s, 0.2, 0.4, 0.9)), 0);
    int chk = orc_tex_checker(s, orc_tex_constant(s, 0.2, 0.3, 0.1), orc_tex_constant(s, 0.9, 0.9, 0.9));
    orc_add_sphere(s, c4, 100, orc_mat_lambertian(s, chk), 0, 0);
    orc_add_sphere(s, c5, 50, orc_mat_metal(s, orc_tex_constant(s, 1, 1, 1), 0), 0, 0);
    orc_add_volume(s, c6, 80, 0.01, orc_mat_fog(s, orc_tex_constant(s, 0.9, 0.9, 0.9)), 0);
    orc_add_plane(s, ORC_PLANEXZ, 123, 423, 147, 412, 554, orc_mat_emissive(s, orc_tex_constant(s, 7, 7, 7)), 1, 1);
    double pos[3] = { 478, 278, -600 }, at[3] = { 278, 278, 0 }, up[3] = { 0, 1, 0 };
    orc_set_camera(s, pos, at, up, 40, 0, 10, 0, 1);
    orc_build_bvh(s);
    return s;
}

/* ---- the `nested` fixture: wrappers nested every way the reference's object model allows (transform.hpp:9-149 compose freely), volumes
 * around boxes (primitive.hpp:332-343 takes any object*), groups used as objects (group.hpp:12,68-102), an instance inside an instance.
 * Same scene, same construction order as ref_harness.cpp build_nested and raytracing_amd/host/demo.hpp nested_test(). */
static orc_obj* reg_leaf(orc_scene* s, orc_obj* o) { o->id = s->n_prims; push(&s->prims, &s->n_prims, &s->cap_prims, o); return o; }
static orc_obj* mk_sphere(orc_scene* s, v3 c, double radius, int mat)
{ orc_obj* o = new_obj(ORC_SPHERE); o->mat = mat; o->g[0] = c.x; o->g[1] = c.y; o->g[2] = c.z; o->g[3] = radius; return o; }
static orc_obj* mk_box(v3 mn, v3 mx, int mat)
{ orc_obj* o = new_obj(ORC_BOX); o->mat = mat; o->g[0] = mn.x; o->g[1] = mn.y; o->g[2] = mn.z; o->g[3] = mx.x; o->g[4] = mx.y; o->g[5] = mx.z; o->g[6] = 0; o->g[7] = 1; return o; }
static orc_obj* mk_plane(int kind, double a0, double a1, double b0, double b1, double k, int mat)
{ orc_obj* o = new_obj(kind); o->mat = mat; o->g[0] = a0; o->g[1] = a1; o->g[2] = b0; o->g[3] = b1; o->g[4] = k; return o; }
static orc_obj* mk_flip(orc_obj* inner) { orc_obj* o = new_obj(O_FLIP); o->l = inner; o->mat = inner->mat; return o; }
static orc_obj* mk_volume(orc_obj* surface, double density, int fog_mat)          /* primitive.hpp:332-344 */
{ orc_obj* o = new_obj(ORC_VOLUME); o->l = surface; o->mat = fog_mat; o->g[4] = -1.0 / density; return o; }
static orc_obj* mk_group(int n, orc_obj** items)
{
    orc_obj* o = new_obj(O_GROUP);
    o->items = (orc_obj**) malloc(sizeof(orc_obj*) * n); o->n_items = n;
    memcpy(o->items, items, sizeof(orc_obj*) * n);
    return o;
}
static orc_obj* mk_bvh(orc_scene* s, int n, orc_obj** items, double start, double end)   /* new bvh(group): the builder permutes a copy */
{
    orc_obj** tmp = (orc_obj**) malloc(sizeof(orc_obj*) * n);
    memcpy(tmp, items, sizeof(orc_obj*) * n);
    orc_obj* b = bvh_build(&s->rng, tmp, 0, n, start, end);
    free(tmp);
    return b;
}
static void add_obj(orc_scene* s, orc_obj* o) { push(&s->objects, &s->n_objects, &s->cap_objects, o); }
static int lam(orc_scene* s, double r, double g, double b) { return orc_mat_lambertian(s, orc_tex_constant(s, r, g, b)); }

orc_scene* orc_build_nested(int width, int height, unsigned seed)
{
    orc_scene* s = orc_scene_new(width, height, seed);
    s->wrapper_rows = 1;
    /* 0 ground, 1 lamp */
    add_obj(s, reg_leaf(s, mk_plane(ORC_PLANEXZ, -40, 40, -40, 40, 0, lam(s, 0.5, 0.5, 0.5))));
    { orc_obj* lamp = reg_leaf(s, mk_plane(ORC_PLANEXZ, -4, 4, -4, 4, 14, orc_mat_emissive(s, orc_tex_constant(s, 8, 8, 8)))); add_obj(s, mk_flip(lamp)); }
    /* 2 translate(translate(sphere)) */
    add_obj(s, mk_translate(mk_translate(reg_leaf(s, mk_sphere(s, V(0, 1, 0), 1, lam(s, 0.8, 0.3, 0.3))), V(-6, 0, 0)), V(0, 0, -3)));
    /* 3 translate(rotatey(rotatey(box))) */
    add_obj(s, mk_translate(mk_rotatey(mk_rotatey(reg_leaf(s, mk_box(V(-1, 0, -1), V(1, 2, 1), lam(s, 0.3, 0.8, 0.3))), 20), 25), V(-2, 0, 4)));
    /* 4 translate(rotatey(translate(rotatey(sphere)))), glass */
    add_obj(s, mk_translate(mk_rotatey(mk_translate(mk_rotatey(reg_leaf(s, mk_sphere(s, V(0, 1, 0), 1, orc_mat_dielectric(s, 1.5))), 30), V(3, 0, 0)), 40), V(2, 0, -6)));
    /* 5 rotatey(translate(sphere)), metal */
    add_obj(s, mk_rotatey(mk_translate(reg_leaf(s, mk_sphere(s, V(0, 1.2, 0), 1.2, orc_mat_metal(s, orc_tex_constant(s, 0.8, 0.8, 0.6), 0.1))), V(7, 0, 0)), 50));
    /* 6 volume(box) */
    add_obj(s, reg_leaf(s, mk_volume(mk_box(V(-9, 0, 2), V(-6, 3, 5), orc_mat_dielectric(s, 1.5)), 0.4, orc_mat_fog(s, orc_tex_constant(s, 0.9, 0.9, 0.9)))));
    /* 7 volume(translate(rotatey(box))) */
    add_obj(s, reg_leaf(s, mk_volume(mk_translate(mk_rotatey(mk_box(V(0, 0, 0), V(2.5, 3, 2.5), orc_mat_dielectric(s, 1.5)), -18), V(5, 0, 3)), 0.5,
                                     orc_mat_fog(s, orc_tex_constant(s, 0.2, 0.4, 0.9)))));
    /* 8 translate(volume(sphere)) */
    add_obj(s, mk_translate(reg_leaf(s, mk_volume(mk_sphere(s, V(0, 1.5, 0), 1.5, orc_mat_dielectric(s, 1.5)), 0.6, orc_mat_fog(s, orc_tex_constant(s, 0.9, 0.5, 0.2)))), V(-3, 0, -8)));
    /* 9-11 a group as an object */
    {
        double tri[9] = { -12, 0, -4, -9, 0, -4, -10.5, 3, -4.5 };
        orc_obj* m[3];
        m[0] = reg_leaf(s, mk_sphere(s, V(-10, 1, -6), 1, lam(s, 0.7, 0.7, 0.2)));
        m[1] = reg_leaf(s, mk_triangle_p(tri, lam(s, 0.2, 0.7, 0.7)));
        m[2] = reg_leaf(s, mk_box(V(-13, 0, -9), V(-11, 2, -7), lam(s, 0.7, 0.2, 0.7)));
        add_obj(s, mk_group(3, m));
    }
    /* 12-13 translate(group) */
    {
        orc_obj* m[2];
        m[0] = reg_leaf(s, mk_sphere(s, V(0, 1, 0), 1, lam(s, 0.9, 0.6, 0.3)));
        m[1] = reg_leaf(s, mk_sphere(s, V(2.2, 1, 0), 1, orc_mat_metal(s, orc_tex_constant(s, 0.9, 0.9, 0.9), 0.0)));
        add_obj(s, mk_translate(mk_group(2, m), V(8, 0, -9)));
    }
    /* 14-18 an instance inside an instance: translate(rotatey(bvh{ sphere, translate(bvh{3 spheres}), sphere })) */
    {
        orc_obj* in[3];
        in[0] = reg_leaf(s, mk_sphere(s, V(0, 0.6, 0), 0.6, lam(s, 0.9, 0.2, 0.2)));
        in[1] = reg_leaf(s, mk_sphere(s, V(1.5, 0.6, 0), 0.6, lam(s, 0.2, 0.9, 0.2)));
        in[2] = reg_leaf(s, mk_sphere(s, V(0, 0.6, 1.5), 0.6, lam(s, 0.2, 0.2, 0.9)));
        orc_obj* inner = mk_bvh(s, 3, in, 0, 0);
        orc_obj* mid[3];
        mid[0] = reg_leaf(s, mk_sphere(s, V(-2, 1, 0), 1, lam(s, 0.6, 0.6, 0.6)));
        mid[1] = mk_translate(inner, V(0, 2.2, 0));
        mid[2] = reg_leaf(s, mk_sphere(s, V(2.5, 1, 1), 1, orc_mat_dielectric(s, 1.5)));
        add_obj(s, mk_translate(mk_rotatey(mk_bvh(s, 3, mid, 0, 0), 35), V(-9, 0, 9)));
    }
    /* 19 flip(translate(sphere)): the flip counts.  20 translate(flip(sphere)): translate's set_face_normal overwrites it */
    add_obj(s, mk_flip(mk_translate(reg_leaf(s, mk_sphere(s, V(0, 1, 0), 1, orc_mat_dielectric(s, 1.5))), V(4, 0, 8))));
    add_obj(s, mk_translate(mk_flip(reg_leaf(s, mk_sphere(s, V(0, 1, 0), 1, orc_mat_dielectric(s, 1.5)))), V(7, 0, 7)));
    /* 21-22 flip(bvh{sphere, rect}) */
    {
        orc_obj* m[2];
        m[0] = reg_leaf(s, mk_sphere(s, V(11, 1, 2), 1, orc_mat_dielectric(s, 1.5)));
        m[1] = reg_leaf(s, mk_plane(ORC_PLANEXY, 9, 13, 0, 3, 0, lam(s, 0.4, 0.4, 0.8)));
        add_obj(s, mk_flip(mk_bvh(s, 2, m, 0, 0)));
    }
    /* 23 translate(msphere) */
    {
        orc_obj* o = new_obj(ORC_MSPHERE); o->mat = lam(s, 0.7, 0.3, 0.1);
        o->g[0] = 0; o->g[1] = 1; o->g[2] = 0; o->g[3] = 0; o->g[4] = 1.5; o->g[5] = 0; o->g[6] = 0; o->g[7] = 1; o->g[8] = 0.8;
        add_obj(s, mk_translate(reg_leaf(s, o), V(-5, 0, 6)));
    }
    orc_add_plane(s, ORC_PLANEXZ, -4, 4, -4, 4, 14, orc_mat_emissive(s, orc_tex_constant(s, 8, 8, 8)), 1, 1);
    double pos[3] = { 2, 9, 26 }, at[3] = { 0, 1, 0 }, up[3] = { 0, 1, 0 };
    orc_set_camera(s, pos, at, up, 40, 0.1, 25, 0, 1);
    orc_build_bvh(s);
    return s;
}

/* ------------------------------------------------------------------ pre-order view */
static int pre_walk(orc_scene* s, const orc_obj* o, int* cap)
{
    if (s->n_pre == *cap) {
        *cap *= 2;
        s->pre = (const orc_obj**) realloc(s->pre, sizeof(void*) * *cap);
        s->pre_first = (int*) realloc(s->pre_first, sizeof(int) * *cap);
        s->pre_left = (int*) realloc(s->pre_left, sizeof(int) * *cap);
        s->pre_right = (int*) realloc(s->pre_right, sizeof(int) * *cap);
    }
    int idx = s->n_pre++;
    s->pre[idx] = o; s->pre_left[idx] = s->pre_right[idx] = -1;
    if (o->first_entry < 0) ((orc_obj*) o)->first_entry = idx;        /* an object reached twice (both children of a span-1 node) */
    s->pre_first[idx] = o->first_entry;
    const orc_obj* base = o;                                            /* ref_harness.cpp tree_dump::walk: a wrapped bvh is dumped too */
    while (base->kind == O_FLIP || base->kind == O_TRANSLATE || base->kind == O_ROTATEY) base = base->l;
    if (base->kind == O_BVH) {
        int l = pre_walk(s, base->l, cap);
        int r = pre_walk(s, base->r, cap);
        s->pre_left[idx] = l; s->pre_right[idx] = r;
    }
    return idx;
}
static void ensure_pre(const orc_scene* cs)
{
    orc_scene* s = (orc_scene*) cs;
    if (s->pre) return;
    int cap = 1024;
    s->pre = (const orc_obj**) malloc(sizeof(void*) * cap);
    s->pre_first = (int*) malloc(sizeof(int) * cap);
    s->pre_left = (int*) malloc(sizeof(int) * cap);
    s->pre_right = (int*) malloc(sizeof(int) * cap);
    s->n_pre = 0;
    s->root_entry = (int*) malloc(sizeof(int) * (s->n_objects + 1));
    for (int i = 0; i < s->n_objects; i++) { s->root_entry[i] = s->n_pre; pre_walk(s, s->objects[i], &cap); }
}

static void put_tex_row(const orc_scene* s, int ti, double* row)
{
    const orc_tex* t = &s->texs[ti];
    row[0] = t->kind;
    if (t->kind == ORC_TEX_CONSTANT) { row[1] = t->d[0]; row[2] = t->d[1]; row[3] = t->d[2]; }
    else if (t->kind == ORC_TEX_CHECKER) {
        const orc_tex* e = &s->texs[(int) t->d[0]]; const orc_tex* o = &s->texs[(int) t->d[1]];
        row[1] = e->d[0]; row[2] = e->d[1]; row[3] = e->d[2]; row[4] = o->d[0]; row[5] = o->d[1]; row[6] = o->d[2];
    }
    else if (t->kind == ORC_TEX_IMAGE) { row[1] = t->d[0]; row[2] = t->d[1]; row[3] = t->d[2]; row[4] = t->d[3]; }
}
static void put_mat_row(const orc_scene* s, int mi, double* row)
{
    const orc_mat* m = &s->mats[mi];
    row[0] = m->kind;
    if (m->kind == ORC_MAT_METAL || m->kind == ORC_MAT_DIELECTRIC) row[1] = m->d[0];
    if (m->kind != ORC_MAT_DIELECTRIC && m->tex >= 0) put_tex_row(s, m->tex, row + 2);   /* fog: base_color, like lambertian */
    if (m->kind == ORC_MAT_PHONG) { for (int i = 0; i < 12; i++) row[9 + i] = m->d[1 + i]; row[21] = m->d[13]; row[22] = m->d[14]; }
}
static void put_obj_row(const orc_scene* s, const orc_obj* o, double* r, int left, int right)
{
    const orc_obj* base = o; int flipped = 0, xf = 0; double sn = 0, cs = 1; v3 target = V(0, 0, 0);
    for (;;) {                                                          /* ref_harness.cpp unwrap(): outermost first */
        if (base->kind == O_FLIP) { base = base->l; flipped ^= 1; }
        else if (base->kind == O_TRANSLATE) { target = V(base->g[0], base->g[1], base->g[2]); xf |= 4; base = base->l; }
        else if (base->kind == O_ROTATEY) { sn = base->g[0]; cs = base->g[1]; xf |= 2; base = base->l; }
        else break;
    }
    r[1] = flipped;
    double* g = r + 4; double* m = r + 30;
    if (s->wrapper_rows) {                                              /* ref_harness.cpp tree_dump::rows under g_wrapper_rows */
        double code = 0, sum = 0; int i = 1;
        for (const orc_obj* w = o; w->kind == O_FLIP || w->kind == O_TRANSLATE || w->kind == O_ROTATEY; w = w->l, i++) {
            code = code * 4 + (w->kind == O_FLIP ? 1 : w->kind == O_ROTATEY ? 2 : 3);
            if (w->kind == O_TRANSLATE) sum += i * (w->g[0] + 2 * w->g[1] + 3 * w->g[2]);
            if (w->kind == O_ROTATEY) sum += i * (w->g[0] + 2 * w->g[1]);
        }
        r[28] = code; r[29] = sum;
    }
    if (base->kind == O_GROUP) { r[0] = 9; g[0] = base->n_items; return; }
    if (base->kind == O_BVH) {
        r[0] = 0;
        g[0] = base->box.mn.x; g[1] = base->box.mn.y; g[2] = base->box.mn.z;
        g[3] = base->box.mx.x; g[4] = base->box.mx.y; g[5] = base->box.mx.z;
        g[6] = left; g[7] = right;
        if (xf) { r[3] = xf; g[8] = sn; g[9] = cs; g[10] = target.x; g[11] = target.y; g[12] = target.z; }
        return;
    }
    r[0] = base->kind;
    int ng = base->kind == ORC_SPHERE ? 4 : base->kind == ORC_VOLUME ? 5 : base->kind == ORC_MSPHERE ? 9 : base->kind == ORC_TRIANGLE ? 24 : base->kind == ORC_BOX ? 11 : 5;
    for (int k = 0; k < ng; k++) g[k] = base->g[k];
    if (base->kind == ORC_BOX) {                                        /* ref_harness.cpp tree_dump::rows, K_BOX */
        aabb bb; obj_bbox(o, 0, 0, &bb);
        r[3] = base->g[11];
        if (xf) { r[3] = xf; g[6] = sn; g[7] = cs; g[8] = target.x; g[9] = target.y; g[10] = target.z; }   /* wrapper OBJECTS around the box: the innermost of each kind, as unwrap() keeps them */
        g[11] = bb.mn.x; g[12] = bb.mn.y; g[13] = bb.mn.z; g[14] = bb.mx.x; g[15] = bb.mx.y; g[16] = bb.mx.z;
    }
    if (base->kind == ORC_VOLUME) {                                     /* K_VOLUME: the boundary sphere's fields if it is one, neg_inv_d */
        const orc_obj* b = base->l;
        for (int k = 0; k < 4; k++) g[k] = 0;
        if (b->kind == ORC_SPHERE) for (int k = 0; k < 4; k++) g[k] = b->g[k];
        if (s->wrapper_rows) {
            double code = 0;
            while (b->kind == O_FLIP || b->kind == O_TRANSLATE || b->kind == O_ROTATEY) { code = code * 4 + (b->kind == O_FLIP ? 1 : b->kind == O_ROTATEY ? 2 : 3); b = b->l; }
            g[5] = code * 16 + (b->kind == ORC_SPHERE ? 1 : b->kind == ORC_BOX ? 7 : 0);
            if (b->kind == ORC_BOX) for (int k = 0; k < 6; k++) g[6 + k] = b->g[k];
        }
    }
    if (base->mat >= 0) put_mat_row(s, base->mat, m);
}

int orc_tree_rows(const orc_scene* s, double* rows, int max_rows)
{
    ensure_pre(s);
    int n = s->n_pre < max_rows ? s->n_pre : max_rows;
    memset(rows, 0, sizeof(double) * ORC_TREE_ROW * n);
    for (int i = 0; i < n; i++) {
        double* r = rows + (size_t) i * ORC_TREE_ROW;
        put_obj_row(s, s->pre[i], r, s->pre_left[i], s->pre_right[i]);
        r[2] = s->pre_first[i];
    }
    return s->n_pre;
}
int orc_light_rows(const orc_scene* s, double* rows, int max_rows)
{
    int n = s->n_lights < max_rows ? s->n_lights : max_rows;
    memset(rows, 0, sizeof(double) * ORC_TREE_ROW * n);
    for (int i = 0; i < n; i++) {
        double* r = rows + (size_t) i * ORC_TREE_ROW;
        put_obj_row(s, s->lights[i], r, -1, -1);
        r[2] = i;
    }
    return s->n_lights;
}
int orc_entry_prim_ids(const orc_scene* s, int32_t* out, int max_rows)
{
    ensure_pre(s);
    for (int i = 0; i < s->n_pre && i < max_rows; i++) out[i] = s->pre[i]->kind == O_BVH ? -1 : s->pre[i]->id;
    return s->n_pre;
}
static void put3(double* o, v3 a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }
void orc_camera_fields(const orc_scene* s, double out[24])
{
    put3(out, s->cam_pos); put3(out + 3, s->left_bottom); put3(out + 6, s->horizontal); put3(out + 9, s->vertical);
    put3(out + 12, s->cu); put3(out + 15, s->cv); put3(out + 18, s->cw);
    out[21] = s->lens_radius; out[22] = s->shutter0; out[23] = s->shutter1;
}

/* ------------------------------------------------------------------ flattened export */
/* One walk serves the counts (all outputs NULL) and the export, in the layout of include/rtcore.h.
 *   translate / rotatey around anything become instance transforms: the members keep their object-space geometry and name a CHAIN of
 *     records, outermost first.  One record holds translate(rotatey(.)), rotatey(.) or translate(.): a rotatey joins the record before
 *     it when that one holds a lone translate; every other wrapper opens a new record.
 *   a box takes the innermost record into its own fields (RT_PRIM_FLAG_ROTATEY / _TRANSLATE), as cornell_box()'s boxes always did;
 *   a flip counts only where no translate / rotatey lies outside it -- their set_face_normal calls overwrite `frontward`
 *     (transform.hpp:25, 105) -- and reaches the leaves below it as RT_PRIM_FLAG_FLIP (no chain) or RT_PRIM_FLAG_FLIP_OUTER;
 *   a group used as an object becomes group links (nodes with an inverted box): link(m0, link(m1, ...));
 *   a volume's boundary is a sphere or a box inside translate(rotatey(.)). */
typedef struct { int fl; double sn, cs; v3 t; } exp_level;
typedef struct {
    double* bbox; int32_t* child; double* xform_data; int32_t* node_xform; int32_t* prim_xform; int32_t* prim_flags; double* prim_data;
    int next, n_xf, cur;                 /* cur: first record of the current chain once written, -1: not yet */
    exp_level chain[32]; int n_chain;
    int flip;                            /* parity of the flips that count */
} exp_ctx;

static int chain_records(exp_ctx* c)
{
    if (c->n_chain == 0) return -1;
    if (c->cur >= 0) return c->cur;
    c->cur = c->n_xf;
    for (int k = 0; k < c->n_chain; k++) {
        const exp_level* l = &c->chain[k];
        if (c->xform_data) {
            double* x = c->xform_data + 6 * c->n_xf;
            x[0] = (l->fl & 2) ? l->sn : 0.0; x[1] = (l->fl & 2) ? l->cs : 1.0; x[2] = l->t.x; x[3] = l->t.y; x[4] = l->t.z;
            x[5] = l->fl | (k + 1 < c->n_chain ? 8 : 0);
        }
        c->n_xf++;
    }
    return c->cur;
}

static int export_node(const orc_obj* o, exp_ctx* c);
static int export_group(orc_obj* const* items, int n, exp_ctx* c)      /* link(m0, link(m1, ... link(mk-1, mk))) */
{
    if (n == 1) return export_node(items[0], c);
    int idx = c->next++;
    if (c->bbox) { double* b = c->bbox + 6 * idx; b[0] = b[1] = b[2] = ORC_INF; b[3] = b[4] = b[5] = -ORC_INF; }
    if (c->node_xform) c->node_xform[idx] = -1;
    int l = export_node(items[0], c);
    int r = export_group(items + 1, n - 1, c);
    if (c->child) { c->child[2 * idx] = l; c->child[2 * idx + 1] = r; }
    return idx;
}

static int export_node(const orc_obj* o, exp_ctx* c)
{
    if (o->kind == O_FLIP && o->id < 0) {                              /* (a flip registered as the primitive itself is a leaf: export_prim unwraps it) */
        const int counts = c->n_chain == 0;
        if (counts) c->flip ^= 1;
        int code = export_node(o->l, c);
        if (counts) c->flip ^= 1;
        return code;
    }
    if (o->kind == O_TRANSLATE || o->kind == O_ROTATEY) {
        const exp_ctx saved = *c;
        exp_level* last = c->n_chain ? &c->chain[c->n_chain - 1] : NULL;
        if (o->kind == O_ROTATEY && last && last->fl == 4) { last->fl |= 2; last->sn = o->g[0]; last->cs = o->g[1]; }
        else {
            if (c->n_chain == 32) { fprintf(stderr, "oracle export: more than 32 transform records in one chain\n"); abort(); }
            exp_level* l = &c->chain[c->n_chain++];
            memset(l, 0, sizeof *l); l->cs = 1;
            if (o->kind == O_TRANSLATE) { l->fl = 4; l->t = V(o->g[0], o->g[1], o->g[2]); }
            else { l->fl = 2; l->sn = o->g[0]; l->cs = o->g[1]; }
        }
        c->cur = -1;
        int code = export_node(o->l, c);
        const int next = c->next, n_xf = c->n_xf;
        *c = saved; c->next = next; c->n_xf = n_xf;
        return code;
    }
    if (o->kind == O_GROUP) {
        if (o->n_items == 0) { fprintf(stderr, "oracle export: empty group\n"); abort(); }
        return export_group(o->items, o->n_items, c);
    }
    if (o->kind != O_BVH) {                                             /* a leaf */
        const orc_obj* base = o;
        while (base->kind == O_FLIP) base = base->l;
        int extra = 0, n_chain = c->n_chain, cur = c->cur;
        if (base->kind == ORC_BOX && n_chain > 0 && ((int) base->g[11] & 6) == 0) {   /* the innermost record moves into the box */
            const exp_level* l = &c->chain[n_chain - 1];
            extra |= l->fl;
            if (c->prim_data) {
                double* d = c->prim_data + (size_t) o->id * ORC_PRIM_STRIDE;
                d[6] = (l->fl & 2) ? l->sn : 0.0; d[7] = (l->fl & 2) ? l->cs : 1.0; d[8] = l->t.x; d[9] = l->t.y; d[10] = l->t.z;
            }
            c->n_chain--; c->cur = -1;
        }
        const int xi = chain_records(c);
        if (c->n_chain != n_chain) { c->n_chain = n_chain; c->cur = cur; }
        if (c->flip) extra |= xi >= 0 ? 16 : 1;
        if (c->prim_xform && xi >= 0) c->prim_xform[o->id] = xi;
        if (c->prim_flags) c->prim_flags[o->id] |= extra;
        return ~o->id;
    }
    int idx = c->next++;
    if (c->bbox) { put3(c->bbox + 6 * idx, o->box.mn); put3(c->bbox + 6 * idx + 3, o->box.mx); }
    const int xi = chain_records(c);
    if (c->node_xform) c->node_xform[idx] = xi;
    int l = export_node(o->l, c);
    int r = o->r == o->l ? l : export_node(o->r, c);                  /* span 1 (bvh.hpp:31-34): the same child twice */
    if (c->child) { c->child[2 * idx] = l; c->child[2 * idx + 1] = r; }
    return idx;
}
void orc_export_counts(const orc_scene* s, int32_t c[8])
{
    exp_ctx x; memset(&x, 0, sizeof x); x.cur = -1;
    for (int i = 0; i < s->n_objects; i++) export_node(s->objects[i], &x);
    c[0] = x.next; c[1] = s->n_prims; c[2] = s->n_mats; c[3] = s->n_texs; c[4] = s->n_lights; c[5] = s->n_objects;
    c[6] = s->n_pixel_bytes; c[7] = x.n_xf;
}
/* instance transforms of the flattened scene: xform_data n_xforms x 6 (sin, cos, target, flags), node_xform / prim_xform -1 = none */
void orc_export_instances(const orc_scene* s, double* xform_data, int32_t* node_xform, int32_t* prim_xform)
{
    int32_t c[8]; orc_export_counts(s, c);
    for (int i = 0; i < c[0]; i++) node_xform[i] = -1;
    for (int i = 0; i < c[1]; i++) prim_xform[i] = -1;
    exp_ctx x; memset(&x, 0, sizeof x); x.cur = -1; x.xform_data = xform_data; x.node_xform = node_xform; x.prim_xform = prim_xform;
    for (int i = 0; i < s->n_objects; i++) export_node(s->objects[i], &x);
}
static void export_prim(const orc_obj* o, int32_t* kind, int32_t* flags, int32_t* mat, double* data)
{
    const orc_obj* base = o; int flipped = 0;
    while (base->kind == O_FLIP) { base = base->l; flipped ^= 1; }
    *kind = base->kind; *flags = flipped; *mat = base->mat;
    memset(data, 0, sizeof(double) * ORC_PRIM_STRIDE);
    memcpy(data, base->g, sizeof(double) * ORC_PRIM_STRIDE);
    if (base->kind == ORC_BOX) { *flags |= (int) base->g[11]; data[11] = 0; }   /* wrappers travel as RT_PRIM_FLAG_ROTATEY / _TRANSLATE */
    if (base->kind == ORC_VOLUME) {                                     /* the boundary: a sphere, or a box inside translate(rotatey(.)) */
        const orc_obj* b = base->l;
        int fl = 0; double sn = 0, cs = 1; v3 target = V(0, 0, 0);
        while (b->kind == O_FLIP) b = b->l;
        if (b->kind == O_TRANSLATE) { target = V(b->g[0], b->g[1], b->g[2]); fl |= 4; b = b->l; while (b->kind == O_FLIP) b = b->l; }
        if (b->kind == O_ROTATEY) { sn = b->g[0]; cs = b->g[1]; fl |= 2; b = b->l; while (b->kind == O_FLIP) b = b->l; }
        memset(data, 0, sizeof(double) * ORC_PRIM_STRIDE);
        if (b->kind == ORC_SPHERE && fl == 0) { for (int k = 0; k < 4; k++) data[k] = b->g[k]; data[4] = base->g[4]; }
        else if (b->kind == ORC_BOX && ((int) b->g[11] & 6) == 0) {
            for (int k = 0; k < 6; k++) data[k] = b->g[k];
            data[6] = sn; data[7] = cs; data[8] = target.x; data[9] = target.y; data[10] = target.z; data[11] = base->g[4];
            *flags |= 8 | fl;
        }
        else { fprintf(stderr, "oracle export: a volume's boundary must be a sphere or a box inside translate(rotatey(.))\n"); abort(); }
    }
}
void orc_export(const orc_scene* s, double* node_bbox, int32_t* node_child, int32_t* roots,
                int32_t* prim_kind, int32_t* prim_flags, int32_t* prim_mat, double* prim_data,
                int32_t* mat_kind, int32_t* mat_tex, double* mat_data,
                int32_t* tex_kind, double* tex_data, uint8_t* tex_pixels,
                int32_t* light_kind, int32_t* light_flags, int32_t* light_mat, double* light_data, double cam[24])
{
    for (int i = 0; i < s->n_prims; i++) export_prim(s->prims[i], prim_kind + i, prim_flags + i, prim_mat + i, prim_data + (size_t) i * ORC_PRIM_STRIDE);
    exp_ctx x; memset(&x, 0, sizeof x); x.cur = -1; x.bbox = node_bbox; x.child = node_child; x.prim_flags = prim_flags; x.prim_data = prim_data;
    for (int i = 0; i < s->n_objects; i++) roots[i] = export_node(s->objects[i], &x);
    for (int i = 0; i < s->n_mats; i++) { mat_kind[i] = s->mats[i].kind; mat_tex[i] = s->mats[i].tex; memcpy(mat_data + i * ORC_MAT_STRIDE, s->mats[i].d, sizeof(double) * ORC_MAT_STRIDE); }
    for (int i = 0; i < s->n_texs; i++) { tex_kind[i] = s->texs[i].kind; memcpy(tex_data + i * ORC_TEX_STRIDE, s->texs[i].d, sizeof(double) * ORC_TEX_STRIDE); }
    if (s->n_pixel_bytes) memcpy(tex_pixels, s->pixels, s->n_pixel_bytes);
    for (int i = 0; i < s->n_lights; i++) export_prim(s->lights[i], light_kind + i, light_flags + i, light_mat + i, light_data + (size_t) i * ORC_PRIM_STRIDE);
    orc_camera_fields(s, cam);
}

/* ------------------------------------------------------------------ instrumented first hit */
typedef struct { uint32_t nbox, nleaf; uint64_t hash; int hit_entry; } walk_stats;
static inline void fnv(uint64_t* h, uint32_t v) { *h = (*h ^ (uint64_t) v) * 0x100000001B3ULL; }

static int rewalk(const orc_scene* s, int idx, const ray* r, isect* x, walk_stats* st, orc_rng* g)
{
    const orc_obj* o = s->pre[idx];
    fnv(&st->hash, (uint32_t) idx);
    if (o->kind == O_BVH) {
        st->nbox++;
        if (!aabb_hit(&o->box, r)) return 0;
        int hl = rewalk(s, s->pre_left[idx], r, x, st, g);
        double tmr = hl ? x->t : r->t_max;
        ray rr = mkray(r->o, r->d, r->time, r->t_min, tmr);
        int hr = rewalk(s, s->pre_right[idx], &rr, x, st, g);
        return hl || hr;
    }
    st->nleaf++;
    int hit = obj_hit(o, r, x, g);
    if (hit) st->hit_entry = idx;
    return hit;
}

void orc_first_hit(const orc_scene* s, int n, const double* rays, int32_t* entry_first, int32_t* prim_id, double* t,
                   int32_t* nbox, int32_t* nleaf, int32_t* front, uint64_t* hash, double* rec)
{
    ensure_pre(s);
    for (int i = 0; i < n; i++) {
        const double* p = rays + (size_t) i * 7;
        ray r = mkray(V(p[0], p[1], p[2]), V(p[3], p[4], p[5]), p[6], 1e-3, ORC_INF);
        walk_stats st = { 0, 0, 0xcbf29ce484222325ULL, -1 };
        isect x; memset(&x, 0, sizeof x);
        /* group.hpp:68-83 over the root group: children in order, t_max shrinking */
        int has = 0; double closest = r.t_max; isect tmp; memset(&tmp, 0, sizeof tmp);
        for (int k = 0; k < s->n_objects; k++) {
            ray rr = mkray(r.o, r.d, r.time, r.t_min, closest);
            int he = st.hit_entry;
            if (rewalk(s, s->root_entry[k], &rr, &tmp, &st, (orc_rng*) &s->rng)) {   /* volumes draw from the scene's stream, like the harness' walk */ has = 1; closest = tmp.t; x = tmp; } else st.hit_entry = he;
        }
        entry_first[i] = has ? s->pre_first[st.hit_entry] : -1;
        prim_id[i] = has ? (x.leaf ? x.leaf->id : s->pre[st.hit_entry]->id) : -1;   /* the leaf itself, also inside an instance */
        t[i] = has ? x.t : 0.0;
        nbox[i] = st.nbox; nleaf[i] = st.nleaf; hash[i] = st.hash;
        front[i] = has ? x.front : -1;
        if (rec) {
            double* w = rec + (size_t) i * 9;
            memset(w, 0, sizeof(double) * 9);
            if (has) { put3(w, x.pos); put3(w + 3, x.normal); w[6] = x.uv.u; w[7] = x.uv.v; }
        }
    }
}

/* ------------------------------------------------------------------ integrator (raytracing.cpp:30-70) */
static double* g_path_out = 0; static int g_path_n = 0, g_path_cap = 0;   /* orc_debug_path capture (single-threaded use) */

static v3 trace(orc_scene* s, orc_rng* g, const ray* wi, int depth, uint64_t* rays)
{
    if (depth <= 0) return V(0, 0, 0);
    if (g_lm_out) g_lm_segment++;                        /* 1-based index of the segment being traced (orc_debug_libm only: nothing is written otherwise) */
    if (g_path_out && g_path_n < g_path_cap) { double* w = g_path_out + 7 * g_path_n++; w[0] = wi->o.x; w[1] = wi->o.y; w[2] = wi->o.z; w[3] = wi->d.x; w[4] = wi->d.y; w[5] = wi->d.z; w[6] = wi->time; }
    isect x; memset(&x, 0, sizeof x);
    (*rays)++;
    if (!group_hit(s->objects, s->n_objects, wi, &x, g)) return V(0, 0, 0);

    ray wo = ray_default();
    v3 att = V(0, 0, 0);
    int spec = 0;
    double pdf, mpdf;
    const orc_mat* m = &s->mats[x.mat];
    v3 emission = mat_emit(s, m, &x);
    if (!mat_shading(s, g, m, wi, &x, &wo, &att, &spec)) return emission;

    if (s->compat_fixed) {
        /* No counterpart in the reference ("compat=fixed"): the same estimator with three defects repaired.  A specular
         * lobe is a delta: its weight is the attenuation (the reference divides by material::pdf, which is 0 for a dielectric and for a
         * metal reflection below the surface: 0 * inf).  phong sets wo (mat_shading above).  A diffuse direction of zero probability
         * contributes nothing instead of 0/0. */
        if (spec) return vadd(emission, vmul(att, trace(s, g, &wo, depth - 1, rays)));
        if (rnd(g) < 0.5) wo = mkray(x.pos, lights_rand(s, g, x.pos), wi->time, wi->t_min, wi->t_max);
        mpdf = mat_pdf(g, m, &x, &wo, 1);
        pdf = 0.5 * (lights_pdf(s, x.pos, wo.d) + mpdf);
        if (!(pdf > 0) || !(mpdf > 0)) return emission;
        return vadd(emission, sdiv(vmul(smul(mpdf, att), trace(s, g, &wo, depth - 1, rays)), pdf));
    }
    if (spec) {
        mpdf = mat_pdf(g, m, &x, &wo, 0);
        pdf = mpdf;
    }
    else {
        if (rnd(g) < 0.5) wo = mkray(x.pos, lights_rand(s, g, x.pos), wi->time, wi->t_min, wi->t_max);
        mpdf = mat_pdf(g, m, &x, &wo, 0);
        pdf = 0.5 * (lights_pdf(s, x.pos, wo.d) + mpdf);
    }
    if (s->early_exit) {
        /* the sample is rejected (raytracing.cpp:108-110) whatever the rest of the path does: see the note in
         * raytracing_amd/csrc/rt_device.h render_kernel; NaN makes the caller's test fire */
        v3 w = smul(mpdf, att);
        double inv = 1 / pdf;
        if (!isfinite(inv) || !isfinite(w.x) || !isfinite(w.y) || !isfinite(w.z)) return V(NAN, NAN, NAN);
    }
    v3 li = trace(s, g, &wo, depth - 1, rays);
    /* emission + attenuation * mat_pdf * L / pdf  ==  emission + (1/pdf) * ((mat_pdf*attenuation) (*) L) */
    return vadd(emission, sdiv(vmul(smul(mpdf, att), li), pdf));
}

static ray camera_emit(const orc_scene* s, orc_rng* g, double x, double y)   /* camera.hpp:62-68 */
{
    v3 defocus = smul(s->lens_radius, rnd_disk(g));
    v3 offset = vadd(smul(defocus.x, s->cu), smul(defocus.y, s->cv));
    v3 dir = vsub(vsub(vadd(vadd(s->left_bottom, smul(x, s->horizontal)), smul(y, s->vertical)), s->cam_pos), offset);
    return mkray(vadd(s->cam_pos, offset), dir, rnd_range(g, s->shutter0, s->shutter1), 1e-3, ORC_INF);
}

/* primary rays through the camera on a side x side grid of pixel centres, drawn from the scene's own glibc
 * stream -- the same calls ref_harness.cpp do_firsthit makes, so the stream position stays in step */
void orc_primary_rays(orc_scene* s, int side, double* rays7)
{
    for (int i = 0; i < side; i++) for (int j = 0; j < side; j++) {
        ray r = camera_emit(s, &s->rng, (j + 0.5) / side, (i + 0.5) / side);
        double* w = rays7 + (size_t) (i * side + j) * 7;
        put3(w, r.o); put3(w + 3, r.d); w[6] = r.time;
    }
}

static inline int bad3(v3 c)                                              /* util.hpp:142-151 */
{ return isnan(c.x) || isnan(c.y) || isnan(c.z) || fabs(c.x) == ORC_INF || fabs(c.y) == ORC_INF || fabs(c.z) == ORC_INF; }

static inline double clamp01(double v) { return v < 0 ? 0 : (v > 1 ? 1 : v); }   /* util.hpp:119-130 */

void orc_resolve(const double* sum, int n_pixels, int spp, uint8_t* rgb)   /* raytracing.cpp:115-121 */
{
    for (int p = 0; p < n_pixels; p++) {
        v3 c = V(sum[3 * p], sum[3 * p + 1], sum[3 * p + 2]);
        c = smul(1 / (double) spp, c);                                       /* color /= spp  (vec3.hpp:56-58) */
        c = V(pow(c.x, 0.45), pow(c.y, 0.45), pow(c.z, 0.45));
        c = V(clamp01(c.x), clamp01(c.y), clamp01(c.z));
        rgb[3 * p] = (uint8_t) (int) (c.x * 255);
        rgb[3 * p + 1] = (uint8_t) (int) (c.y * 255);
        rgb[3 * p + 2] = (uint8_t) (int) (c.z * 255);
    }
}

void orc_render(orc_scene* s, int spp_begin, int spp_end, int max_depth, int rng_mode, uint64_t seed, int n_threads,
                double* sum_out, uint8_t* rgb_out, uint64_t counters[4])
{
    int w = s->width, h = s->height;
    uint64_t rays = 0, bad = 0;
    fix_pixels(s);
    if (rng_mode == ORC_RNG_GLIBC) {
        orc_rng* g = &s->rng;
        for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) {
            v3 color = V(0, 0, 0);
            for (int k = spp_begin; k < spp_end; k++) {
                double x = (double) (j + rnd(g)) / (w - 1);
                double y = (double) (i + rnd(g)) / (h - 1);
                ray r = camera_emit(s, g, x, y);
                v3 sample = trace(s, g, &r, max_depth, &rays);
                if (bad3(sample)) { sample = V(0, 0, 0); bad++; }
                color = vadd(color, sample);
            }
            put3(sum_out + 3 * ((size_t) i * w + j), color);
        }
        orc_rng peek = *g;
        counters[2] = (uint64_t) glibc_rand(&peek);
    }
    else {
        if (n_threads < 1) n_threads = 1;
        #pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads) reduction(+:rays, bad)
        for (int i = 0; i < h; i++) for (int j = 0; j < w; j++) {
            v3 color = V(0, 0, 0);
            uint32_t pixel = (uint32_t) (i * w + j);
            for (int k = spp_begin; k < spp_end; k++) {
                orc_rng g; memset(&g, 0, sizeof g);
                counter_seed(&g, seed, pixel, (uint32_t) k);
                double x = (double) (j + rnd(&g)) / (w - 1);
                double y = (double) (i + rnd(&g)) / (h - 1);
                ray r = camera_emit(s, &g, x, y);
                v3 sample = trace(s, &g, &r, max_depth, &rays);
                if (bad3(sample)) { sample = V(0, 0, 0); bad++; }
                color = vadd(color, sample);
            }
            put3(sum_out + 3 * ((size_t) i * w + j), color);
        }
        counters[2] = 0;
    }
    counters[0] = rays; counters[1] = bad; counters[3] = 0;
    if (rgb_out) orc_resolve(sum_out, w * h, spp_end - spp_begin, rgb_out);
}

/* rays (o, d, time) of the path of one (pixel, sample) stream in COUNTER mode; returns the number of rays */
int orc_debug_path(orc_scene* s, int i, int j, int k, uint64_t seed, int max_depth, double* rays7, int cap)
{
    orc_rng g; memset(&g, 0, sizeof g);
    counter_seed(&g, seed, (uint32_t) (i * s->width + j), (uint32_t) k);
    double x = (double) (j + rnd(&g)) / (s->width - 1);
    double y = (double) (i + rnd(&g)) / (s->height - 1);
    ray r = camera_emit(s, &g, x, y);
    uint64_t n = 0;
    g_path_out = rays7; g_path_n = 0; g_path_cap = cap;
    trace(s, &g, &r, max_depth, &n);
    g_path_out = 0;
    return g_path_n;
}

/* the libm calls of the same path: rows of (segment, operation, x, y, result); segment s = made while tracing / shading the s-th ray of
 * the path (1-based; 0 = camera), operations: 1 sin, 2 cos, 3 acos, 4 atan2(x = y-argument, y = x-argument), 5 pow, 6 log.  Returns the row count.
 * Single-threaded use only (test tooling). */
int orc_debug_libm(orc_scene* s, int i, int j, int k, uint64_t seed, int max_depth, double* rows5, int cap)
{
    orc_rng g; memset(&g, 0, sizeof g);
    counter_seed(&g, seed, (uint32_t) (i * s->width + j), (uint32_t) k);
    g_lm_out = rows5; g_lm_n = 0; g_lm_cap = cap; g_lm_segment = 0;
    double x = (double) (j + rnd(&g)) / (s->width - 1);
    double y = (double) (i + rnd(&g)) / (s->height - 1);
    ray r = camera_emit(s, &g, x, y);
    uint64_t n = 0;
    trace(s, &g, &r, max_depth, &n);
    g_lm_out = 0;
    return g_lm_n;
}

void orc_set_early_exit(orc_scene* s, int on) { s->early_exit = on; }
void orc_set_compat_fixed(orc_scene* s, int on) { s->compat_fixed = on; }

/* ------------------------------------------------------------------ rng fills */
void orc_glibc_rand_fill(unsigned seed, int n, int32_t* out)
{ orc_rng g; memset(&g, 0, sizeof g); glibc_seed(&g, seed); for (int i = 0; i < n; i++) out[i] = glibc_rand(&g); }
void orc_counter_rand_fill(uint64_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out)
{ orc_rng g; memset(&g, 0, sizeof g); counter_seed(&g, seed, pixel, sample); for (int i = 0; i < n; i++) out[i] = counter_next(&g); }

/* ------------------------------------------------------------------ unit entry points (layouts: ref_harness.cpp do_units) */
void orc_unit_aabb(int n, const double* in, int32_t* out)
{
    for (int i = 0; i < n; i++) {
        const double* p = in + i * 14;
        aabb b = { V(p[0], p[1], p[2]), V(p[3], p[4], p[5]) };
        ray r = mkray(V(p[6], p[7], p[8]), V(p[9], p[10], p[11]), 0, p[12], p[13]);
        out[i] = aabb_hit(&b, &r);
    }
}

void orc_unit_prim(int n, const double* in, double* out, double* pdfs)
{
    orc_scene* s = orc_scene_new(16, 16, 1);
    int m = orc_mat_lambertian(s, orc_tex_constant(s, 0.5, 0.5, 0.5));
    double c[3] = { 0.3, -0.2, 0.1 }, c1[3] = { 0.3, 0.4, 0.1 };
    double p[9] = { -1, -0.5, 0.2, 1.2, -0.4, -0.3, 0.1, 1.3, 0.4 };
    double nn[9] = { 0.1, 0.2, 0.97, -0.2, 0.1, 0.95, 0.0, -0.3, 0.9 };
    double uv[6] = { 0.1, 0.2, 0.9, 0.15, 0.45, 0.95 };
    orc_add_sphere(s, c, 1.25, m, 0, 0);
    orc_add_msphere(s, c, c1, 0, 1, 0.8, m, 0, 0);
    orc_add_plane(s, ORC_PLANEXZ, -1.5, 1.0, -0.5, 2.0, 0.75, m, 0, 0);
    orc_add_triangle_v(s, p, nn, uv, m, 0, 0);
    orc_add_plane(s, ORC_PLANEXZ, -1.5, 1.0, -0.5, 2.0, 0.75, m, 1, 0);
    for (int i = 0; i < n; i++) {
        const double* q = in + i * 9;
        v3 o = V(q[0], q[1], q[2]), d = V(q[3], q[4], q[5]);
        ray r = mkray(o, d, q[6], q[7], q[8]);
        for (int k = 0; k < 5; k++) {
            isect x; memset(&x, 0, sizeof x);
            int h = obj_hit(s->prims[k], &r, &x, NULL);
            double* w = out + ((size_t) k * n + i) * 12;
            memset(w, 0, sizeof(double) * 12);
            w[0] = h;
            if (h) { w[1] = x.t; put3(w + 2, x.pos); put3(w + 5, x.normal); w[8] = x.uv.u; w[9] = x.uv.v; w[10] = x.front; }
            pdfs[k * n + i] = obj_pdf(s->prims[k], o, d);
        }
    }
    orc_scene_free(s);
}

void orc_unit_mat(int n, const double* in, double* out)
{
    orc_scene* s = orc_scene_new(16, 16, 1);
    int chk = orc_tex_checker(s, orc_tex_constant(s, 0.2, 0.3, 0.1), orc_tex_constant(s, 0.9, 0.9, 0.9));
    int mats[6];
    mats[0] = orc_mat_lambertian(s, chk);
    mats[1] = orc_mat_metal(s, orc_tex_constant(s, 0.8, 0.6, 0.2), 0.4);
    mats[2] = orc_mat_dielectric(s, 1.5);
    mats[3] = orc_mat_emissive(s, orc_tex_constant(s, 3, 2, 1));
    double p1[12] = { 0.6, 0.5, 0.2, 0.5, 0.5, 0.5, 0, 0, 0, 0, 0, 0 };
    double p2[12] = { 0.6, 0.5, 0.2, 0, 0, 0, 0, 0, 0, 0.9, 0.9, 0.9 };
    mats[4] = orc_mat_phong(s, p1, 20, 0, -1);
    mats[5] = orc_mat_phong(s, p2, 0, 1.45f, -1);
    for (int i = 0; i < n; i++) {
        const double* p = in + i * 13;
        v3 pos = V(p[0], p[1], p[2]), nrm = V(p[3], p[4], p[5]), wd = V(p[6], p[7], p[8]);
        ray wi = mkray(vsub(pos, wd), wd, p[9], 1e-3, ORC_INF);
        for (int q = 0; q < 6; q++) {
            isect x; memset(&x, 0, sizeof x);
            x.pos = pos; x.uv.u = p[10]; x.uv.v = p[11]; x.t = 1; x.mat = mats[q];
            set_face_normal(&x, &wi, nrm);
            orc_rng g; memset(&g, 0, sizeof g); g.mode = ORC_RNG_GLIBC; glibc_seed(&g, (unsigned) p[12]);
            ray wo = ray_default(); v3 att = V(0, 0, 0); int spec = 0;
            const orc_mat* m = &s->mats[mats[q]];
            v3 em = mat_emit(s, m, &x);
            int sc = mat_shading(s, &g, m, &wi, &x, &wo, &att, &spec);
            double pdf = sc ? mat_pdf(&g, m, &x, &wo, 0) : 0.0;
            double* w = out + ((size_t) q * n + i) * 12;
            w[0] = sc; w[1] = spec; w[2] = pdf;
            put3(w + 3, em);
            put3(w + 6, sc ? att : V(0, 0, 0));
            put3(w + 9, sc ? wo.d : V(0, 0, 0));
        }
    }
    orc_scene_free(s);
}

void orc_unit_cam(int n, double* out, double* fields)
{
    orc_scene* s = orc_scene_new(1200, 800, 1);
    double pos[3] = { 13, 2, 3 }, at[3] = { 0, 0, 0 }, up[3] = { 0, 1, 0 };
    orc_set_camera(s, pos, at, up, 20, 0.1, 10, 0, 1);
    for (int i = 0; i < n; i++) {
        orc_rng g; memset(&g, 0, sizeof g); g.mode = ORC_RNG_GLIBC; glibc_seed(&g, 500 + i);
        ray r = camera_emit(s, &g, (i % 8 + 0.5) / 8.0, (i / 8 + 0.5) / 8.0);
        double* w = out + i * 9;
        put3(w, r.o); put3(w + 3, r.d); w[6] = r.time; w[7] = r.t_min; w[8] = r.t_max;
    }
    orc_camera_fields(s, fields);
    orc_scene_free(s);
}
