// raytracing_amd/csrc/rt_wavefront.h -- RT_MODE_FAST_FP32 as a wavefront path tracer (fp32, the library's SAH tree).
//
// The lane-synchronous bounce loop (rt_device.h) keeps a path in a lane's registers from camera to termination, so a
// wave's traversal lasts as long as its slowest ray and its shading pass serialises every material present: measured
// 25-31 % lane utilisation in the traversal, ~1300 VALU instructions per shading pass, while the kernel is VALU-issue
// bound (profiles/).  Here paths live in a pool in HBM (structure of arrays over slots, 80 B per slot), and the
// per-bounce work is cut into kernels whose lanes are interchangeable:
//
//   wf_extend  persistent waves; a lane that finishes a traversal stores (t, primitive) and at once takes the next
//              slot from the wave's chunk, so no lane waits for a neighbour's longer ray.  Small register footprint
//              (no shading state) => more waves per SIMD to hide the node-fetch latency of large meshes.  The
//              traversal stack is in LDS (lane-interleaved, conflict-free), levels beyond the LDS share spill to HBM.
//   wf_shade   one lane per slot: the trace() body (raytracing.cpp:41-69) for that slot's hit, writes the next segment's
//              ray, or retires the sample (atomic add into the frame) and regenerates the slot with the next
//              (sample, pixel) item -- the pool stays full until the frame's items run out.
//
// Items are numbered sample-major over 8x8 pixel tiles, so neighbouring slots start as neighbouring pixels of one
// sample.  RNG streams are keyed by (seed, pixel, sample) as everywhere else, so the frame does not depend on the pool
// size or on which slot renders a sample (fp32 addition order into the frame aside).
#pragma once

#include "rt_device.h"

namespace rt {

#define RT_WF_DEAD ((int32_t) 0x80000001)   // ray_d.w of a slot that holds no path
#define RT_WF_CHUNK 128u                    // slots a wave reserves per fetch from an extend cursor
#define RT_WF_LDS_LEVELS_MAX 32              // traversal stack levels kept in LDS (1 KiB per level and workgroup)
#define RT_WF_CURSORS 16u                   // extend cursors, one per region of the pool (each on its own 128-B line)

// Device-side control block, one per scene.  Atomics on ONE address cost ~20-50 ns each on this chip (measured: a
// per-wave item counter made wf_shade 5x slower), so nothing on the per-iteration path funnels through a single word:
// items are dealt to slots statically, liveness is a plain-store flag, ray counts are per-workgroup partial sums, and the
// extend cursor is split over RT_WF_CURSORS regions.
struct WfCtl {
    unsigned int alive;                     // set (plain store) by wf_shade when any slot still holds a path
    unsigned int pad[31];
    unsigned int extend_cursor[RT_WF_CURSORS * 32];   // cursor r at [r * 32]: next slot offset inside region r
    unsigned long long n_box, n_leaf, w_node, w_leaf, w_bounce;   // diagnostic launches only (RT_FLAG_COUNT_TESTS)
};

struct WfPool {
    float4* ray_o;       // origin xyz, time
    float4* ray_d;       // direction xyz, w = bits of `from` (id of the primitive the origin lies on, -1 none, RT_WF_DEAD no path)
    float2* hit;         // t, bits of the fprims index (-1: miss)
    float4* beta;        // throughput xyz, w = bits of: remaining depth | poisoned << 30
    float4* rad;         // radiance gathered so far xyz, w = bits of the pixel index
    unsigned long long* rng;
    uint32_t* item;      // the slot's current item; its next one is item + n_slots (items are dealt statically: no shared counter)
    unsigned long long* wg_rays;      // per wf_shade workgroup: rays shaded / samples rejected (summed by the host)
    unsigned long long* wg_rejected;
    uint32_t* spill;     // traversal stack levels beyond the LDS share: [level][extend thread]
    WfCtl* ctl;
    uint32_t n_slots;
    uint32_t n_items;        // items of this pass: samples x padded pixels
    uint32_t n_pix_padded;   // (tiles_x * tiles_y) * 64
    uint32_t tiles_x;
    uint32_t lds_levels;     // stack levels kept in LDS
    uint32_t extend_threads; // grid * block of wf_extend (stride of `spill`)
    uint32_t region_slots;   // slots per extend-cursor region (multiple of RT_WF_CHUNK)
};

#ifdef RT_WITH_FAST_BVH
__device__ inline float as_f(int32_t v) { return __int_as_float(v); }
__device__ inline int32_t as_i(float v) { return __float_as_int(v); }

// (sample, pixel) of an item; false for the padding of edge tiles
__device__ inline bool wf_item(const WfPool& p, const RenderArgs& a, uint32_t item, int& k, int& i, int& j)
{
    const uint32_t s = item / p.n_pix_padded, r = item - s * p.n_pix_padded;
    const uint32_t tile = r >> 6, within = r & 63u;
    const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    k = a.spp_begin + (int) s;
    j = (int) (tx * 8 + (within & 7u));
    i = a.y_begin + (int) (ty * 8 + (within >> 3));
    return j < a.width && i < a.y_end;
}

// First valid item at or after `item` in the slot's arithmetic sequence (stride n_slots); false when the pass has no more.
__device__ inline bool wf_next_item(const WfPool& p, const RenderArgs& a, uint32_t& item, int& k, int& i, int& j)
{
    for (; item < p.n_items; item += p.n_slots)
        if (wf_item(p, a, item, k, i, j)) return true;     // false only for the padding of edge tiles
    return false;
}

// camera sample (raytracing.cpp:102-105) into slot `slot`
__device__ inline void wf_new_path(const Scene<float>& sc, const WfPool& p, const RenderArgs& a, uint32_t slot, uint32_t item, int k, int i, int j)
{
    p.item[slot] = item;
    Rng g;
    const uint32_t pixel = (uint32_t) (i * a.width + j);
    g.seed(a.seed, pixel, (uint32_t) k);
    float x = (float(j) + uniform<float>(g)) / float(a.width - 1);
    float y = (float(i) + uniform<float>(g)) / float(a.height - 1);
    Ray<float> r = camera_emit(sc.cam, g, x, y);
    p.ray_o[slot] = make_float4(r.o.x, r.o.y, r.o.z, r.time);
    p.ray_d[slot] = make_float4(r.d.x, r.d.y, r.d.z, as_f(-1));
    p.beta[slot] = make_float4(1.f, 1.f, 1.f, as_f(a.max_depth));
    p.rad[slot] = make_float4(0.f, 0.f, 0.f, as_f((int32_t) pixel));
    p.rng[slot] = g.s;
}

// ------------------------------------------------------------------ first fill of the pool
__global__ void __launch_bounds__(256) wf_generate_kernel(Scene<float> sc, WfPool p, RenderArgs a)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= p.n_slots) return;
    if (threadIdx.x == 0) { p.wg_rays[blockIdx.x] = 0; p.wg_rejected[blockIdx.x] = 0; }
    int k = 0, i = 0, j = 0;
    uint32_t item = slot;
    if (wf_next_item(p, a, item, k, i, j)) wf_new_path(sc, p, a, slot, item, k, i, j);
    else p.ray_d[slot] = make_float4(0.f, 0.f, 0.f, as_f(RT_WF_DEAD));
}

// ------------------------------------------------------------------ extend: closest hit of every slot's ray
struct WfTrav {
    vec3<float> idir, noid;   // 1/d and -o/d: a slab plane distance is one fma
    float slack;              // absolute slack of the slab comparison (rounding of o/d)
    float closest;
    int32_t hit, cur, sp;
};

__device__ inline void wf_trav_begin(WfTrav& t, const Ray<float>& r)
{
    // a zero direction component would make 0 * inf = NaN in the slab test: nudge it (box tests only)
    auto safe = [](float d) { return fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d; };
    t.idir = vec3<float>(1.0f / safe(r.d.x), 1.0f / safe(r.d.y), 1.0f / safe(r.d.z));
    t.noid = vec3<float>(-r.o.x * t.idir.x, -r.o.y * t.idir.y, -r.o.z * t.idir.z);
    // t = fma(plane, idir, -o*idir) rounds once; what is left is the rounding of o*idir itself (<= 1/2 ulp of it)
    t.slack = 2.4e-7f * fmaxf(fmaxf(fabsf(t.noid.x), fabsf(t.noid.y)), fabsf(t.noid.z));
    t.closest = rt_inf<float>();
    t.hit = -1;
    t.cur = 0;
    t.sp = 0;
}

__device__ inline void wf_slab(const float* b, const WfTrav& t, float t_min, float& t_near, bool& hit)
{
    float x0 = fmaf(b[0], t.idir.x, t.noid.x), x1 = fmaf(b[3], t.idir.x, t.noid.x);
    float y0 = fmaf(b[1], t.idir.y, t.noid.y), y1 = fmaf(b[4], t.idir.y, t.noid.y);
    float z0 = fmaf(b[2], t.idir.z, t.noid.z), z1 = fmaf(b[5], t.idir.z, t.noid.z);
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_min));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t.closest));
    t_near = tn;
    hit = tn <= fmaf(tf, 1.0000004f, t.slack);     // conservative under fp32 rounding
}

#ifndef RT_WF_EXTEND_WAVES
#define RT_WF_EXTEND_WAVES 5    // waves per SIMD wf_extend is compiled for
#endif

// kSpill: the tree is deeper than the LDS share of the stack (rare: > RT_WF_LDS_LEVELS_MAX levels)
template <bool kCount, bool kSpill>
__global__ void __launch_bounds__(256, RT_WF_EXTEND_WAVES) wf_extend_kernel(Scene<float> sc, WfPool p, RenderArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* stack = reinterpret_cast<uint32_t*>(smem) + threadIdx.x;                  // level l of this lane: stack[l * 256]
    uint32_t* spill = p.spill + (blockIdx.x * 256u + threadIdx.x);                      // level l >= lds_levels: spill[(l - lds_levels) * extend_threads]
    const int lds_levels = (int) p.lds_levels;
    const int lane = (int) __lane_id();
    const float t_min = consts<float>::tmin;
    const BNode* __restrict__ nodes = sc.bnodes;
    const Prim<float>* __restrict__ fprims = sc.fprims;
    const int node_keep = (int) a.node_keep, refill_min = (int) a.regen_min;
    if (blockIdx.x == 0 && threadIdx.x == 0) p.ctl->alive = 0;                           // wf_shade sets it again (no other kernel runs now)
    stack[0] = (uint32_t) RT_BVH_SENTINEL;

    // (volatile keeps the two address spaces apart: merged into one flat access they cost a memory round trip per pop)
    auto push = [&](int level, uint32_t v) {
        if (!kSpill || level < lds_levels) stack[level * 256] = v;
        else *(volatile uint32_t*) &spill[(size_t) (level - lds_levels) * p.extend_threads] = v;
    };
    auto pop = [&](int level) -> uint32_t {
        if (!kSpill || level < lds_levels) return stack[level * 256];
        return *(volatile uint32_t*) &spill[(size_t) (level - lds_levels) * p.extend_threads];
    };

    VisitStats vs; vs.n_box = 0; vs.n_leaf = 0; vs.hash = 0; vs.w_node = vs.w_leaf = vs.w_bounce = 0;
    uint32_t chunk_pos = 0, chunk_end = 0;     // wave-uniform: the wave's reserved run of slots
    bool exhausted = false;                    // wave-uniform: every region's cursor has passed its end
    uint32_t region = (blockIdx.x * 4u + (threadIdx.x >> 6)) % RT_WF_CURSORS, regions_tried = 0;   // home region first, then its neighbours
    int32_t slot = -1;
    Ray<float> ray;
    WfTrav tv; tv.cur = RT_BVH_SENTINEL; tv.sp = 0; tv.hit = -1; tv.closest = 0.f; tv.slack = 0.f;

    for (;;) {
        // ---- refill idle lanes from the wave's chunk
        const unsigned long long idle = __ballot(slot < 0);
        if (idle && !(exhausted && chunk_pos == chunk_end) && (idle == ~0ull || __popcll(idle) >= refill_min)) {
            while (chunk_pos == chunk_end && !exhausted) {
                const uint32_t r0 = region * p.region_slots;
                const uint32_t rsize = r0 >= p.n_slots ? 0u : min(p.region_slots, p.n_slots - r0);
                unsigned int base = rsize;
                // a glance at the cursor first (it only grows): an exhausted region costs a load, not an atomic
                if (lane == 0 && rsize && __hip_atomic_load(&p.ctl->extend_cursor[region * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < rsize)
                    base = atomicAdd(&p.ctl->extend_cursor[region * 32], RT_WF_CHUNK);
                base = (unsigned int) __builtin_amdgcn_readfirstlane((int) base);
                if (base < rsize) { chunk_pos = r0 + base; chunk_end = r0 + min(base + RT_WF_CHUNK, rsize); regions_tried = 0; }
                else {
                    region = (region + 1) % RT_WF_CURSORS;
                    if (++regions_tried >= RT_WF_CURSORS) exhausted = true;
                }
            }
            const uint32_t avail = chunk_end - chunk_pos;
            if (avail) {
                const uint32_t rank = (uint32_t) __popcll(idle & ((1ull << lane) - 1ull));
                if (slot < 0 && rank < avail) {
                    const uint32_t s = chunk_pos + rank;
                    const float4 d = p.ray_d[s];
                    if (as_i(d.w) != RT_WF_DEAD) {
                        const float4 o = p.ray_o[s];
                        ray.o = vec3<float>(o.x, o.y, o.z); ray.time = o.w;
                        ray.d = vec3<float>(d.x, d.y, d.z); ray.from = as_i(d.w);
                        slot = (int32_t) s;
                        wf_trav_begin(tv, ray);
                    }
                }
                chunk_pos += min(avail, (uint32_t) __popcll(idle));
            }
        }
        const int n_trav = __popcll(__ballot(slot >= 0));
        if (n_trav == 0) {
            if (exhausted && chunk_pos == chunk_end) break;
            continue;
        }

        // ---- inner nodes: while enough of the traversing lanes stand on one
        for (;;) {
            const int n_inner = __popcll(__ballot(tv.cur >= 0));
            if (n_inner == 0 || n_inner * 256 < n_trav * node_keep) break;
            if (tv.cur >= 0) {
                const BNode n = nodes[tv.cur];
                if (kCount && first_active_lane()) vs.w_node++;
                float tl, tr; bool hl, hr;
                wf_slab(n.lb, tv, t_min, tl, hl);
                wf_slab(n.rb, tv, t_min, tr, hr);
                if (kCount) vs.n_box += 2;
                if (hl && hr) {
                    const bool left_first = tl <= tr;
                    const int32_t far = left_first ? n.right : n.left;
                    tv.cur = left_first ? n.left : n.right;
                    tv.sp++;
                    push(tv.sp, (uint32_t) far);
                }
                else if (hl) tv.cur = n.left;
                else if (hr) tv.cur = n.right;
                else { tv.cur = (int32_t) pop(tv.sp); tv.sp--; }
            }
        }
        // ---- leaves
        if (tv.cur < 0 && tv.cur != RT_BVH_SENTINEL) {
            const int code = ~tv.cur, first = code >> 3, count = (code & 7) + 1;
            for (int k = 0; k < count; k++) {
                float t;
                if (kCount && first_active_lane()) vs.w_leaf++;
                if (kCount) vs.n_leaf++;
                if (prim_t(fprims[first + k], sc.side(), ray, t_min, tv.closest, t)) { tv.closest = t; tv.hit = first + k; }
            }
            tv.cur = (int32_t) pop(tv.sp); tv.sp--;
        }
        // ---- retire finished traversals
        if (slot >= 0 && tv.cur == RT_BVH_SENTINEL) {
            p.hit[slot] = make_float2(tv.closest, as_f(tv.hit));
            slot = -1;
        }
    }

    if (kCount) {
        unsigned long long nb = vs.n_box, nl = vs.n_leaf, wn = vs.w_node, wl = vs.w_leaf;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            nb += __shfl_down(nb, off, 64); nl += __shfl_down(nl, off, 64);
            wn += __shfl_down(wn, off, 64); wl += __shfl_down(wl, off, 64);
        }
        if (lane == 0) { atomicAdd(&p.ctl->n_box, nb); atomicAdd(&p.ctl->n_leaf, nl); atomicAdd(&p.ctl->w_node, wn); atomicAdd(&p.ctl->w_leaf, wl); }
    }
}

// ------------------------------------------------------------------ shade: trace() body for every slot's hit
template <bool kCount>
__global__ void __launch_bounds__(256, RT_WAVES_FP32) wf_shade_kernel(Scene<float> sc, WfPool p, RenderArgs a, float* __restrict__ accum)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    const int lane = (int) __lane_id();
    if (blockIdx.x == 0 && threadIdx.x < RT_WF_CURSORS) p.ctl->extend_cursor[threadIdx.x * 32] = 0;   // for the next wf_extend launch
    unsigned long long n_rays = 0, n_rejected = 0;
    bool live = false, done = false;
    float4 rd = make_float4(0.f, 0.f, 0.f, as_f(RT_WF_DEAD));
    if (slot < p.n_slots) rd = p.ray_d[slot];
    live = as_i(rd.w) != RT_WF_DEAD;

    if (live) {
        const float4 ro = p.ray_o[slot];
        const float2 ht = p.hit[slot];
        float4 bt = p.beta[slot];
        float4 rdn = p.rad[slot];
        Rng g; g.s = p.rng[slot];
        Ray<float> ray;
        ray.o = vec3<float>(ro.x, ro.y, ro.z); ray.time = ro.w;
        ray.d = vec3<float>(rd.x, rd.y, rd.z); ray.from = as_i(rd.w);
        vec3<float> beta(bt.x, bt.y, bt.z), L(rdn.x, rdn.y, rdn.z);
        int depth = as_i(bt.w) & 0xffff;
        const int traced = a.max_depth - depth + 1;                                     // segments of this path, this one included
        bool poisoned = (as_i(bt.w) >> 30) & 1;
        const int32_t prim = as_i(ht.y);
        n_rays = 1;
        done = true;
        if (prim >= 0) {
            Hit<float> h;
            fill_record(sc.fprims[prim], sc.side(), ray, ht.x, sc.need_uv != 0, h);
            const Mat<float>& m = sc.mats[h.mat];
            const vec3<float> texc = m.tex >= 0 ? tex_color(sc, m.tex, h.pos, h.u, h.v) : vec3<float>();
                vec3<float> emission = mat_emit(m, h, texc);
            Ray<float> wo; wo.time = 0.f;                                               // `ray wo;` raytracing.cpp:41
            vec3<float> att;
            bool spec = false;
            L += beta * emission;
            if (mat_shading(g, m, ray, h, texc, wo, att, spec)) {
                float pdf, mpdf;
                if (spec) {                                                             // raytracing.cpp:51-54
                    mpdf = mat_pdf(g, m, h, wo);
                    pdf = mpdf;
                }
                else {                                                                  // raytracing.cpp:55-61
                    if (uniform<float>(g) < 0.5f) { wo.o = h.pos; wo.from = h.id; wo.d = light_rand(sc, g, h.pos); wo.time = ray.time; }
                    mpdf = mat_pdf(g, m, h, wo);
                    pdf = 0.5f * (lights_pdf(sc, h.pos, wo.d) + mpdf);
                }
                float inv = 1.0f / pdf;                                                 // raytracing.cpp:69, vec3.hpp:121-123
                vec3<float> w = mpdf * att;
                if (!rt_finite(inv) || !rt_finite(w.x) || !rt_finite(w.y) || !rt_finite(w.z)) poisoned = true;
                beta = beta * (inv * w);
                depth--;
                done = depth <= 0;
                if (poisoned && (a.flags & RT_FLAG_EARLY_EXIT)) done = true;
                if (!done) {
                    p.ray_o[slot] = make_float4(wo.o.x, wo.o.y, wo.o.z, wo.time);
                    p.ray_d[slot] = make_float4(wo.d.x, wo.d.y, wo.d.z, as_f(wo.from));
                    p.beta[slot] = make_float4(beta.x, beta.y, beta.z, as_f(depth | (poisoned ? 1 << 30 : 0)));
                    p.rad[slot] = make_float4(L.x, L.y, L.z, rdn.w);
                    p.rng[slot] = g.s;
                }
            }
        }
        if (done) {
            bool bad = poisoned || !rt_finite(L.x) || !rt_finite(L.y) || !rt_finite(L.z);  // raytracing.cpp:108-110
            if (bad) n_rejected = 1;
            if (a.flags & RT_FLAG_DEBUG_RAY_COUNTS) { L = vec3<float>(float(traced), bad ? 1.f : 0.f, 0.f); bad = false; }
            if (!bad) {
                float* dst = accum + (size_t) (uint32_t) as_i(rdn.w) * 3;
                unsafeAtomicAdd(dst, L.x);
                unsafeAtomicAdd(dst + 1, L.y);
                unsafeAtomicAdd(dst + 2, L.z);
            }
        }
    }

    // ---- regenerate retired slots with their next items; a slot that has none left is dead from here on
    bool holds = live && !done;
    if (live && done) {
        int k = 0, i = 0, j = 0;
        uint32_t item = p.item[slot] + p.n_slots;
        holds = wf_next_item(p, a, item, k, i, j);
        if (holds) wf_new_path(sc, p, a, slot, item, k, i, j);
        else p.ray_d[slot] = make_float4(0.f, 0.f, 0.f, as_f(RT_WF_DEAD));
    }
    if (__ballot(holds) && lane == 0) p.ctl->alive = 1u;

    // ---- ray / rejected counts: wave shuffle, then one plain read-modify-write of this workgroup's own partial sum
    __shared__ unsigned long long part[8];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_rays += __shfl_down(n_rays, off, 64);
        n_rejected += __shfl_down(n_rejected, off, 64);
    }
    if (lane == 0) { part[threadIdx.x >> 6] = n_rays; part[4 + (threadIdx.x >> 6)] = n_rejected; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long r = part[0] + part[1] + part[2] + part[3], x = part[4] + part[5] + part[6] + part[7];
        if (r) p.wg_rays[blockIdx.x] += r;
        if (x) p.wg_rejected[blockIdx.x] += x;
        if (kCount && r) atomicAdd(&p.ctl->w_bounce, (r + 63) / 64);
    }
}

#endif  // RT_WITH_FAST_BVH

}  // namespace rt
