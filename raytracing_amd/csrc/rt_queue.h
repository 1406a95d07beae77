// raytracing_amd/csrc/rt_queue.h -- the fast modes' render kernel with traversal and shading DECOUPLED per wave.
//
// Why: in rt_wave.h a lane owns a path from camera to termination, so a lane whose closest hit is known idles until enough of its
// wave's lanes want the shade phase, and a lane without a path until enough want regeneration: measured on the mesh workload, of 64
// lanes per node step 30 stand on an inner node, 5 on a leaf, 20 wait with a known hit, 8 have no path -- 48 % lane utilisation in the
// loop that takes 70 % of the time, in a kernel bound by instruction issue.
//
// Here a PATH is a slot of a per-wave pool in global memory (one 64-byte / 128-byte record: ray, throughput, RNG, depth, tile slot,
// hit) and lanes are interchangeable workers.  Every wave keeps two LIFO lists of slot numbers in LDS (one byte per entry, both in
// one array: known hits grow from the bottom, rays ready to traverse from the top) and alternates three wave-uniform steps:
//   hand-off : lanes whose traversal has ended store (t, primitive) into their slot, push it on the hit list and, together with the
//              idle lanes, pop ready rays: load the ray, start its traversal.  Batched: done when >= `handoff_min` lanes wait;
//   shade    : when 64 hits are listed (fewer only when lanes starve), 64 lanes load one slot each and run the trace() body at full
//              width; a path that goes on is written back and listed as a ready ray, a path that ends adds its sample to the LDS
//              tile and its slot at once takes the tile's next (sample, pixel) item (camera ray) -- so a slot only dies when the
//              tile's items run out;
//   traverse : one round of the node loop and the leaf step of rt_wave.h, unchanged, over the lanes that hold a ray.
// A lane carries only its traversal state and ray in registers; throughput, RNG state, depth and pixel stay in the slot.
// Same arithmetic, RNG streams and per-sample results as rt_wave.h; only the order of the tile additions differs.
// Nothing is shared between waves but the tile sums and the item counter, so there is no cross-wave synchronisation.
#pragma once

#include "rt_wave.h"

namespace rt {

#define RT_Q_FRESH 1u               // slot state: no path yet (or its item fell outside the image): takes the next item when shaded
#define RT_Q_LIVE 0u

template <typename R> struct QSlot;
template <> struct alignas(16) QSlot<float> {        // 64 B: four 16-byte accesses
    float o[3]; float time;
    float d[3]; int32_t from;
    float beta[3]; uint32_t meta;                     // meta: depth (21 bits) | tile slot << 21 (8 bits) | poisoned << 29 | state << 30
    uint32_t rng_lo, rng_hi; float t; int32_t hit;
};
template <> struct alignas(16) QSlot<double> {       // 128 B: seven 16-byte accesses (the last 16 bytes are padding)
    double o[3]; double time;
    double d[3]; double t;
    double beta[3]; uint32_t rng_lo, rng_hi;
    uint32_t meta; int32_t hit; int32_t from; int32_t pad0;
    int32_t pad[4];
};

// what the hand-off reads / writes: the ray, and (t, hit)
__device__ inline void q_load_ray(const QSlot<float>* s, Ray<float>& r)
{
    const uint4* p = reinterpret_cast<const uint4*>(s);
    const uint4 a = p[0], b = p[1];
    r.o = vec3<float>(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)); r.time = __uint_as_float(a.w);
    r.d = vec3<float>(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z)); r.from = (int32_t) b.w;
}
__device__ inline void q_load_ray(const QSlot<double>* s, Ray<double>& r)
{
    const double2* p = reinterpret_cast<const double2*>(s);
    const double2 a = p[0], b = p[1], c = p[2], d = p[3];
    r.o = vec3<double>(a.x, a.y, b.x); r.time = b.y;
    r.d = vec3<double>(c.x, c.y, d.x);
    r.from = -1;                                      // only the fp32 primitive tests read it
}
__device__ inline void q_store_hit(QSlot<float>* s, float t, int32_t hit)
{
    uint2 v; v.x = __float_as_uint(t); v.y = (uint32_t) hit;
    *reinterpret_cast<uint2*>(&s->t) = v;
}
__device__ inline void q_store_hit(QSlot<double>* s, double t, int32_t hit) { s->t = t; s->hit = hit; }

template <typename R, bool kCount, bool kQuant, int kFull>
__global__ void __launch_bounds__(256, (sizeof(R) == 8 ? (kFull == 5 ? RT_WAVES_FP64_SPHERES : 3) : kFull == 3 ? RT_WAVES_OBJ : kFull == 5 ? RT_WAVES_SPHERES : RT_WAVES_FP32))     // fp64: 3 waves (this kernel's slots spill 250-300 B at 128 VGPRs)
render_queue_kernel(Scene<R> sc, RenderArgs a, R* __restrict__ accum, unsigned long long* __restrict__ counters, QSlot<R>* __restrict__ pools,
                    unsigned int* __restrict__ pool_locks)
{
    const int tiles_x = (a.width + 15) >> 4;
    int tile_x, tile_y;
    if (a.xcd_tiles) {                                                        // see rt_wave.h
        const int tiles_y = (a.y_end - a.y_begin + 15) >> 4;
        const int stx = (tiles_x + 7) >> 3;
        const int s = (int) (blockIdx.x >> 3), super = (s >> 6) * 8 + (int) (blockIdx.x & 7u), within = s & 63;
        const int sy = super / stx, sx = super - sy * stx;
        tile_x = sx * 8 + (within & 7); tile_y = sy * 8 + (within >> 3);
        if (tile_x >= tiles_x || tile_y >= tiles_y) return;                   // whole workgroup, before any barrier
    }
    else { tile_x = (int) (blockIdx.x % (unsigned) tiles_x); tile_y = (int) (blockIdx.x / (unsigned) tiles_x); }
    extern __shared__ __align__(16) unsigned char smem[];
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t stack_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned char*) smem + threadIdx.x * 4u;
    auto stack_at = [&](uint32_t lds_address) -> lds_u32& { return *(lds_u32*) (uintptr_t) lds_address; };
    R* tile_sum = reinterpret_cast<R*>(smem + a.lds_tile_offset);
    unsigned int* item_counter = reinterpret_cast<unsigned int*>(tile_sum + 768);
    unsigned int* pool_id = item_counter + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_slots = (int) a.q_slots;                                      // per wave, <= RT_Q_SLOTS_MAX
    unsigned char* q = reinterpret_cast<unsigned char*>(item_counter + 4) + wave * RT_Q_SLOTS_MAX;
    tile_sum[threadIdx.x * 3] = R(0); tile_sum[threadIdx.x * 3 + 1] = R(0); tile_sum[threadIdx.x * 3 + 2] = R(0);
    if (threadIdx.x == 0) {
        *item_counter = 0u;
        // a pool for this workgroup: pools outnumber the workgroups the chip can hold at once, so a free one exists
        uint32_t h = blockIdx.x % a.q_pools;
        while (atomicCAS(&pool_locks[h], 0u, 1u) != 0u) h = h + 1 == a.q_pools ? 0 : h + 1;
        *pool_id = h;
    }
    stack_at(stack_base) = (uint32_t) RT_BVH_SENTINEL;
    __syncthreads();
    QSlot<R>* __restrict__ pool = pools + ((size_t) *pool_id * 4 + (size_t) wave) * RT_Q_SLOTS_MAX;

    auto pixel_of = [&](int local, int& jj, int& ii) {       // slot 0..255 of the 16x16 tile: waves are 8x8 sub-tiles
        const int wv = local >> 6, ln = local & 63;
        jj = tile_x * 16 + (wv & 1) * 8 + (ln & 7);
        ii = a.y_begin + tile_y * 16 + (wv >> 1) * 8 + (ln >> 3);
    };
    const unsigned int n_items = 256u * (unsigned int) (a.spp_end - a.spp_begin);
    const float t_min = consts<float>::tmin;
    const Prim<R>* __restrict__ fprims = sc.fprims;
    const int node_keep = (int) a.node_keep, handoff_min = (int) a.q_handoff_min, shade_starved = (int) a.q_shade_starved;
    const unsigned long long below = (1ull << lane) - 1ull;

    unsigned long long wave_rays = 0, wave_rejected = 0, wave_untraced = 0;
    VisitStats vs; vs.n_box = 0; vs.n_leaf = 0; vs.hash = 0; vs.w_node = vs.w_leaf = vs.w_bounce = 0; vs.n_zero = 0;
    unsigned long long cyc[4] = { 0, 0, 0, 0 }, t0 = kCount ? wall_clock64() : 0;   // wave time: hand-off, nodes, leaves, shade
    auto lap = [&](int phase) { if (kCount) { const unsigned long long t1 = wall_clock64(); cyc[phase] += t1 - t0; t0 = t1; } };

    // every slot starts FRESH and listed as a "hit": the first shade passes are pure regeneration
    int n_hits = n_slots, n_rays = 0, n_live = n_slots;
    for (int i = lane; i < n_slots; i += 64) { q[i] = (unsigned char) i; pool[i].meta = RT_Q_FRESH << 30; }
    if (a.max_depth <= 0 || n_items == 0) { n_hits = 0; n_live = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

    int slot = -1;                                      // the pool slot this lane traverses for; -1: idle
    Ray<R> ray;
    Trav tv; tv.cur = RT_BVH_SENTINEL; tv.sp = stack_base; tv.hit = -1; tv.closest = 0.f; tv.slack = 0.f; tv.sg = 0u; tv.sel[0] = tv.sel[1] = tv.sel[2] = 0x03020100u;
    Closest<R> best;
    if constexpr (sizeof(R) == 8) best.t = R(0);

    while (n_live > 0) {
        // ---------------------------------------------------------------- hand-off
        {
            const unsigned long long fin = __ballot(slot >= 0 && tv.cur == RT_BVH_SENTINEL);
            const unsigned long long idle = __ballot(slot < 0);
            const int n_fin = __popcll(fin), n_idle = __popcll(idle);
            const int n_trav = 64 - n_fin - n_idle;
            const bool useful = n_fin > 0 || (n_idle > 0 && n_rays > 0);
            if (useful && (n_fin + n_idle >= handoff_min || n_trav == 0)) {
                if (slot >= 0 && tv.cur == RT_BVH_SENTINEL) {
                    q_store_hit(pool + slot, closest_of(tv, best), tv.hit);
                    q[n_hits + __popcll(fin & below)] = (unsigned char) slot;
                    slot = -1;
                }
                n_hits += n_fin;
                const unsigned long long want = fin | idle;                 // every lane without a ray now
                const int k = min(__popcll(want), n_rays);
                bool skipped = false;
                if (slot < 0) {
                    const int r = __popcll(want & below);
                    if (r < k) {
                        slot = (int) q[RT_Q_SLOTS_MAX - n_rays + r];
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        q_load_ray(pool + slot, ray);
                        trav_begin(tv, best, ray, stack_base);
                        if (class_is_obj(kFull) && ray.d.x == R(0) && ray.d.y == R(0) && ray.d.z == R(0)) { tv.cur = RT_BVH_SENTINEL; skipped = true; }   // see rt_wave.h
                    }
                }
                wave_untraced += (unsigned long long) __popcll(__ballot(skipped));
                n_rays -= k;
            }
        }
        lap(0);

        // ---------------------------------------------------------------- shade: full passes; partial ones only when lanes starve
        {
            const int n_trav = __popcll(__ballot(slot >= 0 && tv.cur != RT_BVH_SENTINEL));
            const int n_waiting = 64 - n_trav;
            const bool full = n_hits >= 64;
            const bool starved = n_hits > 0 && n_rays == 0 && (n_trav == 0 || (n_hits >= shade_starved && n_waiting >= shade_starved));
            if (full || starved) {
                const int m = min(64, n_hits);
                const bool mine = lane < m;
                int s = 0;
                bool fresh = false, done = false, goes_on = false, rejected_now = false;
                Rng g; g.s = 0;
                Ray<R> pr;
                vec3<R> beta;
                int depth = 0, local = 0;
                bool poisoned = false;
                if (mine) {
                    s = (int) q[n_hits - m + lane];
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const QSlot<R> S = pool[s];
                    fresh = (S.meta >> 30) == RT_Q_FRESH;
                    done = fresh;
                    if (!fresh) {
                        depth = (int) (S.meta & 0x1fffffu); local = (int) ((S.meta >> 21) & 0xffu); poisoned = ((S.meta >> 29) & 1u) != 0;
                        g.s = (uint64_t) S.rng_lo | ((uint64_t) S.rng_hi << 32);
                        pr.o = vec3<R>(S.o[0], S.o[1], S.o[2]); pr.d = vec3<R>(S.d[0], S.d[1], S.d[2]); pr.time = S.time; pr.from = S.from;
                        beta = vec3<R>(S.beta[0], S.beta[1], S.beta[2]);
                        const int path_rays = a.max_depth - depth + 1;               // this path's segments so far, this one included
                        if (kCount && pr.d.x == R(0) && pr.d.y == R(0) && pr.d.z == R(0)) vs.n_zero++;
                        // ---- the trace() body, raytracing.cpp:41-69 (as in rt_wave.h)
                        vec3<R> L;
                        done = true;
                        if (S.hit >= 0) {
                            Hit<R> h;
                            fill_record<R, kFull>(fprims[S.hit], sc.side(), pr, S.t, sc.need_uv != 0, h);
                            const Mat<R>& mt = sc.mats[h.mat];
                            const vec3<R> texc = mt.tex >= 0 ? tex_color(sc, mt.tex, h.pos, h.u, h.v) : vec3<R>();
                            vec3<R> emission = mat_emit<R, class_mat(kFull)>(mt, h, texc);
                            Ray<R> wo; wo.time = R(0);
                            vec3<R> att;
                            bool spec = false;
                            if (mat_shading<R, class_mat(kFull)>(g, mt, pr, h, texc, wo, att, spec)) {
                                R pdf, mpdf;
                                if (spec) { mpdf = mat_pdf<R, class_mat(kFull)>(g, mt, h, wo); pdf = mpdf; }
                                else {
                                    if (uniform<R>(g) < R(0.5)) { wo.o = h.pos; wo.from = h.id; wo.d = light_rand(sc, g, h.pos); wo.time = pr.time; }
                                    mpdf = mat_pdf<R, class_mat(kFull)>(g, mt, h, wo);
                                    pdf = R(0.5) * (lights_pdf(sc, h.pos, wo.d) + mpdf);
                                }
                                R inv = R(1) / pdf;
                                vec3<R> w = mpdf * att;
                                beta = beta * (inv * w);
                                if (!rt_finite(beta.x) || !rt_finite(beta.y) || !rt_finite(beta.z)) poisoned = true;
                                pr = wo;
                                depth--;
                                done = depth <= 0;
                                if (poisoned && (a.flags & RT_FLAG_EARLY_EXIT)) done = true;
                            }
                            else L = beta * emission;
                        }
                        if (done) {
                            bool bad = poisoned || !rt_finite(L.x) || !rt_finite(L.y) || !rt_finite(L.z);
                            rejected_now = bad;
                            if (a.flags & RT_FLAG_DEBUG_RAY_COUNTS) { L = vec3<R>(R(path_rays), bad ? R(1) : R(0), R(0)); bad = false; }
                            if (!bad) {
                                R* dst = tile_sum + local * 3;
                                __hip_atomic_fetch_add(dst, L.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(dst + 1, L.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_fetch_add(dst + 2, L.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                        else goes_on = true;
                    }
                }
                wave_rays += (unsigned long long) __popcll(__ballot(mine && !fresh));
                wave_rejected += (unsigned long long) __popcll(__ballot(rejected_now));
                if (kCount && lane == 0) vs.w_bounce++;
                // ---- slots whose path ended (or never began) take the tile's next items: one LDS atomic per wave
                const unsigned long long need = __ballot(mine && done);
                bool refresh = false;
                if (need) {
                    const int leader = __ffsll(need) - 1;
                    unsigned int base = 0;
                    if (lane == leader) base = atomicAdd(item_counter, (unsigned int) __popcll(need));
                    base = __shfl(base, leader, 64);
                    if (mine && done) {
                        const unsigned int item = base + (unsigned int) __popcll(need & below);
                        if (item < n_items) {
                            local = (int) (item & 255u);
                            const int k = a.spp_begin + (int) (item >> 8);
                            int j, i;
                            pixel_of(local, j, i);
                            if (j < a.width && i < a.y_end) {
                                g.seed(a.seed, (uint32_t) (i * a.width + j), (uint32_t) k);
                                R x = (R(j) + uniform<R>(g)) / R(a.width - 1);                          // raytracing.cpp:102-103
                                R y = (R(i) + uniform<R>(g)) / R(a.height - 1);
                                pr = camera_emit(sc.cam, g, x, y);
                                beta = vec3<R>(R(1), R(1), R(1));
                                depth = a.max_depth;
                                poisoned = false;
                                goes_on = true;
                            }
                            else refresh = true;            // a slot of an edge tile outside the image: try the next item on a later pass
                        }
                    }
                }
                // ---- write back and list
                if (goes_on) {
                    QSlot<R> S;
                    S.o[0] = pr.o.x; S.o[1] = pr.o.y; S.o[2] = pr.o.z; S.time = pr.time;
                    S.d[0] = pr.d.x; S.d[1] = pr.d.y; S.d[2] = pr.d.z; S.from = pr.from;
                    S.beta[0] = beta.x; S.beta[1] = beta.y; S.beta[2] = beta.z;
                    S.meta = (uint32_t) depth | ((uint32_t) local << 21) | (poisoned ? 1u << 29 : 0u) | (RT_Q_LIVE << 30);
                    S.rng_lo = (uint32_t) g.s; S.rng_hi = (uint32_t) (g.s >> 32);
                    S.t = R(0); S.hit = -1;
                    if constexpr (sizeof(R) == 8) { S.pad0 = 0; S.pad[0] = S.pad[1] = S.pad[2] = S.pad[3] = 0; }
                    pool[s] = S;
                }
                n_hits -= m;
                const unsigned long long on = __ballot(goes_on), again = __ballot(refresh);
                if (goes_on) q[RT_Q_SLOTS_MAX - n_rays - 1 - __popcll(on & below)] = (unsigned char) s;
                if (refresh) { q[n_hits + __popcll(again & below)] = (unsigned char) s; pool[s].meta = RT_Q_FRESH << 30; }
                n_rays += __popcll(on);
                n_hits += __popcll(again);
                n_live -= m - __popcll(on) - __popcll(again);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            }
        }
        lap(3);

        // ---------------------------------------------------------------- traverse: one round of the node loop, then the leaves
        {
            const int n_trav = __popcll(__ballot(tv.cur != RT_BVH_SENTINEL));
            if (n_trav > 0) {
                for (;;) {
                    const int n_inner = __popcll(__ballot(tv.cur >= 0));
                    if (n_inner == 0 || n_inner * 256 < n_trav * node_keep) break;
                    if (tv.cur >= 0) node_step<R, kQuant, kCount>(sc, tv, t_min, stack_at, vs);
                }
                lap(1);
                if (tv.cur < 0 && tv.cur != RT_BVH_SENTINEL) {
                    const int code = ~tv.cur, first = code >> 3, count = (code & 7) + 1;
                    for (int k = 0; k < count; k++) {
                        R t;
                        if (kCount && first_active_lane()) vs.w_leaf++;
                        if (kCount) vs.n_leaf++;
                        Prim<R> rec = *at32(fprims, (uint32_t) (first + k) * (uint32_t) sizeof(Prim<R>));
                        asm volatile("" : "+v"(rec.g[0]), "+v"(rec.g[4]), "+v"(rec.g[8]), "+v"(rec.id));
                        if (prim_t<R, kFull>(rec, sc.side(), ray, consts<R>::tmin, closest_of(tv, best), t, nullptr) && take_hit(sc, t, closest_of(tv, best), tv.hit, rec.id)) { set_closest(tv, best, t); tv.hit = first + k; }
                    }
                    tv.cur = (int32_t) stack_at(tv.sp); tv.sp -= 1024u;
                }
                lap(2);
            }
        }
    }

    __syncthreads();
    if (threadIdx.x == 0) atomicExch(&pool_locks[*pool_id], 0u);
    int j, i;
    pixel_of((int) threadIdx.x, j, i);
    if (j < a.width && i < a.y_end) {
        R* out = accum + ((size_t) i * a.width + j) * 3;
        const R sx = tile_sum[threadIdx.x * 3], sy = tile_sum[threadIdx.x * 3 + 1], sz = tile_sum[threadIdx.x * 3 + 2];
        if (a.flags & RT_FLAG_ACCUMULATE) { out[0] += sx; out[1] += sy; out[2] += sz; }
        else { out[0] = sx; out[1] = sy; out[2] = sz; }
    }
    if (lane == 0) {
        atomicAdd(&counters[0], wave_rays);
        if (wave_rejected) atomicAdd(&counters[1], wave_rejected);
        if (wave_untraced) atomicAdd(&counters[15], wave_untraced);
    }
    if (kCount) {
        unsigned long long nb = vs.n_box, nl = vs.n_leaf, wn = vs.w_node, wl = vs.w_leaf, wb = vs.w_bounce, nz = vs.n_zero;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            nb += __shfl_down(nb, off, 64); nl += __shfl_down(nl, off, 64); nz += __shfl_down(nz, off, 64);
            wn += __shfl_down(wn, off, 64); wl += __shfl_down(wl, off, 64); wb += __shfl_down(wb, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&counters[2], nb); atomicAdd(&counters[3], nl); atomicAdd(&counters[4], wn); atomicAdd(&counters[5], wl); atomicAdd(&counters[6], wb); atomicAdd(&counters[7], nz);
            for (int k = 0; k < 4; k++) atomicAdd(&counters[8 + k], cyc[k]);
        }
    }
}

}  // namespace rt
