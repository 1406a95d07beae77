// raytracing_amd/csrc/rt_wave.h -- render kernel of RT_MODE_FAST_FP64 and RT_MODE_FAST_FP32: a per-wavefront state machine.
//
// Replaces, for the fast mode, the lane-synchronous bounce loop of rt_device.h (one traversal to completion for all
// 64 lanes, then one shading pass).  Measured on that loop (profiles/, tools/lane_efficiency.py): the traversal ran at
// 25-31 % lane utilisation, because a wave's traversal lasts as long as its slowest ray (4x the mean node count on the
// mesh workload) while the kernel is VALU-issue bound.  Here every lane carries an explicit state
//     no path  ->  traversing (cur != sentinel)  ->  hit known (cur == sentinel)  ->  traversing | no path
// and the wave alternates between three wave-uniform phases, each entered only when enough lanes want it:
//   regenerate : lanes without a path pull (sample, pixel) items from the workgroup's LDS counter and emit camera rays;
//   traverse   : lanes step through the SAH tree; the phase is left as soon as fewer than `trav_keep`/256 of the lanes
//                holding paths are still traversing -- their traversal state (node, stack pointer, closest hit) simply
//                stays in registers / LDS and resumes on the next round;
//   shade      : lanes whose closest hit is known run the trace() body (raytracing.cpp:41-69) and either start the
//                next segment's traversal or retire the sample into the LDS tile.
// Same arithmetic, RNG streams and sample sums as the loop it replaces (order of the fp32 tile additions aside).
//
// The kernel is templated on the scalar R of everything that decides a RESULT -- rays, primitive tests, hit records, shading,
// throughput, sums.  Box culling is fp32 for both: a box test only decides whether a sub-tree is looked at, never what is found in
// it, and it is conservative (boxes rounded outward, slack on the comparison).  R = float is RT_MODE_FAST_FP32; R = double is
// RT_MODE_FAST_FP64: the reference's own arithmetic (fp64, no FMA contraction, its evaluation order) behind the fast tree, so
// closest hits, t bits and radiance are the reference mode's, at the cost of 96-byte leaf records and fp64 shading.
#pragma once

#include "rt_device.h"
// Waves per SIMD the variants are compiled for.  With the four-wide nodes a visit holds 16 node dwords and 24 plane distances.  fp32: mesh
// workload, 4K, 64 spp, 60.0 ms at 4 waves (128 VGPRs), 59.3 at 3.  fp64 (rays, leaf tests and shading in double): while the shade pass
// carried the library's pow / sincos / sines inline, 3 waves (168 VGPRs) beat 4 (312 B of scratch per lane through the path the node
// fetches use: +8 %, HISTORY); since their replacement the OBJ-loader variant needs ~56 B at 128 VGPRs and 4 waves win: mesh 4K 128 spp
// 152.7 -> 145.7 ms.  Four workgroups must also share a CU's LDS: the OBJ-loader kernels keep no tile sums there (wave_global_sums
// below), so a 37-entry tree's 39-level stack fits four times.  The boxes-and-volumes twin of the spheres variant (6: the Cornell
// room) likewise: 198.0 -> 181.3 ms.  The general variants (0, 1, 2) and the instanced twin (7) still spill 220-480 B at 128 VGPRs: 3.
#ifndef RT_WAVES_OBJ
#define RT_WAVES_OBJ 4
#endif
#ifndef RT_WAVES_FP64_OBJ
#define RT_WAVES_FP64_OBJ 4
#endif
#ifndef RT_WAVES_FP64_SORTED
#define RT_WAVES_FP64_SORTED 4      // render_wave_sorted_kernel<double, ., 3>: the sparse soups
#endif
#ifndef RT_WAVES_FP64_BOXES
#define RT_WAVES_FP64_BOXES 4
#endif
#ifndef RT_WAVES_FP64
#define RT_WAVES_FP64 3
#endif
// the spheres-and-rects variants (rt_device.h class_*: no triangle, phong, fog or image-texture code) are lean enough for more waves.
// 4K, random spheres 32 spp / deep bounce 16 spp, ms: fp64 85.0 / 91.2 at 3 waves, 81.1 / 84.4 at 4, 129 / 128 at 5 (spills);
// fp32 52.7 / 59.6 at 4, 48.9 / 57.9 at 5, the same at 6.  Their twins with boxes / volumes (6) and instances (7) keep the general
// variants' occupancy: Cornell box 29.8 (fp64) / 15.1 ms (fp32) there against 29.9 / 16.0 at the spheres variants' occupancy.
#ifndef RT_WAVES_FP64_SPHERES
#define RT_WAVES_FP64_SPHERES 4
#endif
#ifndef RT_WAVES_SPHERES
#define RT_WAVES_SPHERES 5
#endif

// tools/node_step_isa.sh compiles with -DRT_ASM_MARK(text)=asm volatile(text) to find the node visit, the leaf step and the parts of
// the shade pass in the assembly; otherwise the markers are nothing
#ifndef RT_ASM_MARK
#define RT_ASM_MARK(text)
#endif

namespace rt {

struct Trav {                 // per-lane traversal state, kept across phases
    vec3<float> idir, noid;   // 1/d and -o/d: a slab plane distance is one fma
    float slack;              // absolute slack of the slab comparison (rounding of o/d)
    float closest;            // far bound of the slab tests: the closest hit so far (R = double: rounded from Closest<double>::t)
    int32_t hit, cur;
    // wide nodes (node_step): byte selectors for v_perm_b32 over a node's { high-plane word, low-plane word } pair
    uint32_t sel[3];          // per axis: 0x07060504 (the near planes are the high ones: the ray goes down the axis) or 0x03020100
    uint32_t sg;              // byte a = 0x03 if the ray goes down axis a, else 0: XORed into a selector it reverses the order of the four children
    uint32_t sp;              // LDS byte address of this lane's top stack entry (level * 1024 + lane * 4): push / pop are one add
};
// the closest hit's distance in the precision of the results.  R = float: Trav::closest is that value already.
template <typename R> struct Closest { R t; };
template <> struct Closest<float> {};
__device__ inline float closest_of(const Trav& t, const Closest<float>&) { return t.closest; }
__device__ inline double closest_of(const Trav&, const Closest<double>& c) { return c.t; }
__device__ inline void set_closest(Trav& t, Closest<float>&, float v) { t.closest = v; }
__device__ inline void set_closest(Trav& t, Closest<double>& c, double v) { c.t = v; t.closest = (float) v; }   // the slab test's slack covers the rounding

// Slack of the slab comparison `near <= far * rel + slack_k * max|o/d|`.
// fp32 rays: t = fma(plane, idir, -o*idir) rounds once; what is left is the rounding of o*idir itself (<= 1/2 ulp of it).
// fp64 rays culled in fp32: 1/d comes from the rounded d through the hardware reciprocal (1 ulp), o is rounded, the far bound is the
// rounded closest hit: each plane distance is off by <= ~2e-7 (|t| + 2 |o/d|); the constants below leave 5x that.
template <typename R> struct cull_slack;
template <> struct cull_slack<float> { static constexpr float rel = 1.0000004f, k = 2.4e-7f; };
template <> struct cull_slack<double> { static constexpr float rel = 1.000002f, k = 2e-6f; };

template <typename R> __device__ inline void trav_begin(Trav& t, Closest<R>& c, const Ray<R>& r, uint32_t stack_base)
{
    // a zero direction component would make 0 * inf = NaN in the slab test: nudge it (box tests only)
    auto safe = [](float d) { return fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d; };
    if constexpr (sizeof(R) == sizeof(float)) t.idir = vec3<float>(1.0f / safe(r.d.x), 1.0f / safe(r.d.y), 1.0f / safe(r.d.z));
    else t.idir = vec3<float>(__builtin_amdgcn_rcpf(safe((float) r.d.x)), __builtin_amdgcn_rcpf(safe((float) r.d.y)), __builtin_amdgcn_rcpf(safe((float) r.d.z)));
    t.noid = vec3<float>(-(float) r.o.x * t.idir.x, -(float) r.o.y * t.idir.y, -(float) r.o.z * t.idir.z);
    t.slack = cull_slack<R>::k * fmaxf(fmaxf(fabsf(t.noid.x), fabsf(t.noid.y)), fabsf(t.noid.z));
    t.sel[0] = t.idir.x < 0.f ? 0x07060504u : 0x03020100u; t.sel[1] = t.idir.y < 0.f ? 0x07060504u : 0x03020100u; t.sel[2] = t.idir.z < 0.f ? 0x07060504u : 0x03020100u;
    t.sg = (t.idir.x < 0.f ? 0x03u : 0u) | (t.idir.y < 0.f ? 0x0300u : 0u) | (t.idir.z < 0.f ? 0x030000u : 0u);
    set_closest(t, c, rt_inf<R>());
    t.hit = -1;
    t.cur = 0;
    t.sp = stack_base;
}

template <typename R> __device__ inline void slab_fma(const float* b, const Trav& t, float t_min, float& t_near, bool& hit)
{
    float x0 = fmaf(b[0], t.idir.x, t.noid.x), x1 = fmaf(b[3], t.idir.x, t.noid.x);
    float y0 = fmaf(b[1], t.idir.y, t.noid.y), y1 = fmaf(b[4], t.idir.y, t.noid.y);
    float z0 = fmaf(b[2], t.idir.z, t.noid.z), z1 = fmaf(b[5], t.idir.z, t.noid.z);
    float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_min));
    float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t.closest));
    t_near = tn;
    hit = tn <= fmaf(tf, cull_slack<R>::rel, t.slack);     // conservative under fp32 rounding
}

// record at a 32-bit byte offset from a wave-uniform base: one 32-bit multiply / shift and the scalar-base addressing mode instead
// of 64-bit address arithmetic per lane (rtcore.hip refuses trees or record arrays beyond 4 GiB)
template <typename T> __device__ inline const T* at32(const T* base, uint32_t byte_offset)
{
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_offset);
}

// All four children of a wide node (rt_device.h WNode), in VISITING order: child k of the results is slot k for a ray going up the node's
// sort axis, slot 3 - k for one going down it (`flip` = 0x03030303 then, else 0).  One v_perm_b32 per plane word does both jobs -- it picks
// the near (far) planes' word by the sign of the ray's direction and puts its four bytes in visiting order.
template <typename R> __device__ inline void slab_wide(const WNode& n, const Trav& t, uint32_t flip, float t_min, bool h[4], float* tn = nullptr)
{
    float nr[3][4], fr[3][4];
    const float idir[3] = { t.idir.x, t.idir.y, t.idir.z }, noid[3] = { t.noid.x, t.noid.y, t.noid.z };
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float A = ldexpf(idir[a], (int) (int8_t) (n.o[a] & 0xffu)), B = fmaf(__uint_as_float(n.o[a]), idir[a], noid[a]);
        const uint32_t sel = t.sel[a] ^ flip;
        const uint32_t qn = __builtin_amdgcn_perm(n.hi[a], n.lo[a], sel), qf = __builtin_amdgcn_perm(n.lo[a], n.hi[a], sel);
        // two planes per instruction (v_pk_fma_f32): 12 VALU instructions fewer per visit, mesh workload -0.8 % time
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 A2 = { A, A }, B2 = { B, B };
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
            const f2 qa = { float((qn >> (8 * k)) & 0xffu), float((qn >> (8 * k + 8)) & 0xffu) }, qb = { float((qf >> (8 * k)) & 0xffu), float((qf >> (8 * k + 8)) & 0xffu) };
            const f2 ra = __builtin_elementwise_fma(qa, A2, B2), rb = __builtin_elementwise_fma(qb, A2, B2);
            nr[a][k] = ra.x; nr[a][k + 1] = ra.y; fr[a][k] = rb.x; fr[a][k + 1] = rb.y;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float near = fmaxf(fmaxf(nr[0][k], nr[1][k]), fmaxf(nr[2][k], t_min)), far = fminf(fminf(fr[0][k], fr[1][k]), fminf(fr[2][k], t.closest));
        h[k] = near <= fmaf(far, cull_slack<R>::rel, t.slack);
        if (tn) tn[k] = near;                 // (RT_SORTED_VISIT only)
    }
}

// One node visit of one lane: test the children's boxes, go on with the nearest one hit and push the others (farthest first), or
// pop.  kQuant: four-wide quantised nodes (WNode), else the binary fp32 nodes (BNode).  stack_at(address) names an LDS stack entry;
// kStride: bytes between a lane's stack levels (4 x the workgroup's threads).
template <typename R, bool kQuant, bool kCount, bool kSorted = false, uint32_t kStride = 1024u, typename StackAt>
__device__ inline void node_step(const Scene<R>& sc, Trav& tv, float t_min, StackAt&& stack_at, VisitStats& vs)
{
    if constexpr (kQuant) {
        RT_ASM_MARK("; wide node_step begin");
        const WNode n = *at32(sc.wnodes, (uint32_t) tv.cur * (uint32_t) sizeof(WNode));
        // Slot order is visiting order (rt_bvh_build.h widen_node: children sorted along one axis, whose byte of tv.sg the selector n.n
        // broadcasts): forwards for a ray going up that axis, backwards for one going down.  No entry distances are compared: the
        // five-exchange sort this replaces was 29 of the visit's 122 vector instructions, for 1..3 % fewer visits
        // (tools/trav_policy_sim.cpp).  The reversal is folded into the byte selectors of the plane words, so the hit flags come out
        // of the box tests in visiting order; the four child codes take a select each.
        // ... unless the tree says otherwise (Scene::sorted_visit): in a sparse soup of small triangles slot order is far from distance order
        // (+27 % box tests on the 10 M-triangle soup), and there the hit children are sorted by entry distance with a five-exchange
        // network, as in round 2.  A kernel of its own (render_wave_sorted_kernel): as a wave-uniform branch in one kernel the second
        // path cost the first 12 % (mesh 4K fp64 42.7 -> 47.9 ms per 32 spp)
        if constexpr (kSorted) {
        float t4[4]; bool h4[4];
        slab_wide<R>(n, tv, 0u, t_min, h4, t4);
        if (kCount && first_active_lane()) vs.w_node++;
        if (kCount) vs.n_box += n.pad[0];
        int32_t c4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { c4[k] = n.child[k]; if (!h4[k]) t4[k] = __builtin_inff(); }
        auto order = [&](int i, int j) {
            const bool sw = t4[j] < t4[i];
            const float ti = sw ? t4[j] : t4[i], tj = sw ? t4[i] : t4[j];
            const int32_t ci = sw ? c4[j] : c4[i], cj = sw ? c4[i] : c4[j];
            t4[i] = ti; t4[j] = tj; c4[i] = ci; c4[j] = cj;
        };
        order(0, 1); order(2, 3); order(0, 2); order(1, 3); order(1, 2);
        if (t4[3] < __builtin_inff()) { tv.sp += kStride; stack_at(tv.sp) = (uint32_t) c4[3]; }
        if (t4[2] < __builtin_inff()) { tv.sp += kStride; stack_at(tv.sp) = (uint32_t) c4[2]; }
        if (t4[1] < __builtin_inff()) { tv.sp += kStride; stack_at(tv.sp) = (uint32_t) c4[1]; }
        if (t4[0] < __builtin_inff()) tv.cur = c4[0];
        else { tv.cur = (int32_t) stack_at(tv.sp); tv.sp -= kStride; }
        }
        else {
        const uint32_t flip = __builtin_amdgcn_perm(tv.sg, tv.sg, n.n);
        bool h[4];
        slab_wide<R>(n, tv, flip, t_min, h);
        if (kCount && first_active_lane()) vs.w_node++;
        if (kCount) vs.n_box += n.pad[0];
        const bool rev = flip != 0u;
        const int32_t c0 = rev ? n.child[3] : n.child[0], c1 = rev ? n.child[2] : n.child[1], c2 = rev ? n.child[1] : n.child[2], c3 = rev ? n.child[0] : n.child[3];
        // the children behind the first one hit wait on the stack, the last to be visited pushed first
        if (h[3] && (h[0] || h[1] || h[2])) { tv.sp += kStride; stack_at(tv.sp) = (uint32_t) c3; }
        if (h[2] && (h[0] || h[1])) { tv.sp += kStride; stack_at(tv.sp) = (uint32_t) c2; }
        if (h[1] && h[0]) { tv.sp += kStride; stack_at(tv.sp) = (uint32_t) c1; }
        if (h[0] || h[1] || h[2] || h[3]) tv.cur = h[0] ? c0 : h[1] ? c1 : h[2] ? c2 : c3;
        else { tv.cur = (int32_t) stack_at(tv.sp); tv.sp -= kStride; }
        }
        RT_ASM_MARK("; wide node_step end");
    }
    else {
        const BNode n = *at32(sc.bnodes, (uint32_t) tv.cur * (uint32_t) sizeof(BNode));
        float tl, tr; bool hl, hr;
        slab_fma<R>(n.lb, tv, t_min, tl, hl);
        slab_fma<R>(n.rb, tv, t_min, tr, hr);
        if (kCount && first_active_lane()) vs.w_node++;
        if (kCount) vs.n_box += 2;
        if (hl && hr) {
            const bool left_first = tl <= tr;
            const int32_t far = left_first ? n.right : n.left;
            tv.cur = left_first ? n.left : n.right;
            tv.sp += kStride;
            stack_at(tv.sp) = (uint32_t) far;
        }
        else if (hl) tv.cur = n.left;
        else if (hr) tv.cur = n.right;
        else { tv.cur = (int32_t) stack_at(tv.sp); tv.sp -= kStride; }
    }
}

// OBJ-loader variants: finished paths add their radiance to the frame buffer with global atomics instead of to per-workgroup tile sums in
// LDS.  The sums (6 KiB in fp64, 3 in fp32) were what kept a fourth workgroup out of a CU's LDS beside a deep tree's stack: without them a
// 37-entry stack budget (39 levels, 39 KiB) fits four times, in both precisions (launch_render keeps the host's copy of this rule and clears
// the rows first unless RT_FLAG_ACCUMULATE).  Most paths end black and add nothing; the additions of a pixel come from one workgroup and stay
// in its XCD's L2.  (LDS sums were already added in whatever order the waves finished: no order is lost.)  10 M-triangle soup: fp64
// 0.91 -> 1.11 Grays/s, fp32 0.98 -> 1.22; mesh fp64 unchanged, fp32 +2 % time.
template <typename R, int kFull> constexpr bool wave_global_sums() { return class_is_obj(kFull); }
// ... and with no sums to share, nothing ties four waves into a workgroup: these kernels have a twin that runs as ONE-WAVE workgroups over
// 8 x 8 pixels (a quarter of a tile, render_wave1_kernel / render_wave1_sorted_kernel), each with its own pool of items and its own 256 B
// per level of stack.  A finished wave frees its slot and its LDS at once, and the per-launch fixed cost -- every workgroup's wait for its
// pool's longest paths, proportional to the pool -- falls (pools four times LARGER were measured the other way: +5..11 %, HISTORY).  The
// quarters of a tile land on different CUs, though, and no longer share an L1: mesh 4K fp64 16 / 64 / 128 / 256 spp 20.4 / 70.8 / 139.9 / 274.8 ms
// with four-wave workgroups, 19.9 / 71.4 / 142.3 / 279.4 with one-wave ones; 10 M-triangle soup at 16 spp 235.9 -> 223.7 ms.  launch_render
// takes the twin for launches of at most RT_ONE_WAVE_SPP (32) samples per pixel.

// waves per SIMD a variant is compiled for
template <typename R, int kFull> constexpr int wave_kernel_waves()
{ return sizeof(R) == 8 ? (kFull == 3 ? RT_WAVES_FP64_OBJ : kFull == 5 ? RT_WAVES_FP64_SPHERES : kFull == 6 ? RT_WAVES_FP64_BOXES : RT_WAVES_FP64) : kFull == 3 ? RT_WAVES_OBJ : kFull == 5 ? RT_WAVES_SPHERES : RT_WAVES_FP32; }

template <typename R, bool kCount, bool kQuant, int kFull, bool kSorted, bool kOneWave = false>
__device__ __forceinline__ void render_wave_body(const Scene<R>& sc, const RenderArgs& a, R* __restrict__ accum, unsigned long long* __restrict__ counters)
{
    // Workgroups go to the 8 XCDs round-robin by blockIdx, and each XCD has an L2 of its own: with xcd_tiles set, the image is cut
    // into super-tiles of 8 x 8 tiles, dealt to the XCDs round-robin (neighbouring super-tiles cost about the same, so the XCDs
    // stay balanced), and the tiles an XCD has in flight are two or three compact 128 x 128-pixel blocks rather than every
    // eighth tile of a strip across the image
    constexpr bool kGlobalSums = wave_global_sums<R, kFull>();
    static_assert(!kOneWave || kGlobalSums, "one-wave workgroups: kernels without tile sums only");
    constexpr uint32_t kStride = kOneWave ? 256u : 1024u;     // bytes between a lane's stack levels (4 x the workgroup's threads)
    const int tiles_x = (a.width + 15) >> 4;
    int tile_x, tile_y, sub = 0;
    uint32_t unit;                                                            // the scheduling unit this workgroup belongs to (RenderArgs::unit_order)
    if (a.xcd_tiles) {
        const int tiles_y = (a.y_end - a.y_begin + 15) >> 4;
        const int stx = (tiles_x + 7) >> 3, sty = (tiles_y + 7) >> 3;
        // (one-wave workgroups: four per tile, `sub` = which 8 x 8 quarter)
        const int s = (int) (blockIdx.x >> 3) >> (kOneWave ? 2 : 0), slot = (s >> 6) * 8 + (int) (blockIdx.x & 7u), within = s & 63;
        sub = kOneWave ? (int) ((blockIdx.x >> 3) & 3u) : 0;
        if (slot >= stx * sty) return;                                        // grid padding (whole workgroup, before any barrier)
        const int super = a.unit_order ? (int) a.unit_order[slot] : slot;
        const int sy = super / stx, sx = super - sy * stx;
        tile_x = sx * 8 + (within & 7); tile_y = sy * 8 + (within >> 3);
        if (tile_x >= tiles_x || tile_y >= tiles_y) return;                   // whole workgroup, before any barrier
        unit = (uint32_t) super;
    }
    else {
        const uint32_t tile_slot = kOneWave ? blockIdx.x >> 2 : blockIdx.x;
        sub = kOneWave ? (int) (blockIdx.x & 3u) : 0;
        unit = a.unit_order ? a.unit_order[tile_slot] : tile_slot;
        tile_x = (int) (unit % (unsigned) tiles_x); tile_y = (int) (unit / (unsigned) tiles_x);
    }
    extern __shared__ __align__(16) unsigned char smem[];
    // entry `level` of this lane: LDS address of smem + level * 1024 + lane * 4, kept as a 32-bit LDS address so that a push or
    // pop is one add and a ds instruction (no base to add, no generic pointer)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t stack_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned char*) smem + threadIdx.x * 4u;
    auto stack_at = [&](uint32_t lds_address) -> lds_u32& { return *(lds_u32*) (uintptr_t) lds_address; };
    R* tile_sum = reinterpret_cast<R*>(smem + a.lds_tile_offset);
    unsigned int* item_counter = reinterpret_cast<unsigned int*>(tile_sum + (kGlobalSums ? 0 : 768));
    if constexpr (!kGlobalSums) { tile_sum[threadIdx.x * 3] = R(0); tile_sum[threadIdx.x * 3 + 1] = R(0); tile_sum[threadIdx.x * 3 + 2] = R(0); }
    if (threadIdx.x == 0) { item_counter[0] = 0u; item_counter[1] = 0u; }     // [1]: waves of this workgroup that have finished
    stack_at(stack_base) = (uint32_t) RT_BVH_SENTINEL;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    auto pixel_of = [&](int local, int& jj, int& ii) {       // slot 0..255 of the 16x16 tile: waves are 8x8 sub-tiles (one-wave workgroups: slot 0..63 of quarter `sub`)
        const int wv = kOneWave ? sub : local >> 6, ln = local & 63;
        jj = tile_x * 16 + (wv & 1) * 8 + (ln & 7);
        ii = a.y_begin + tile_y * 16 + (wv >> 1) * 8 + (ln >> 3);
    };
    constexpr unsigned int kSlots = kOneWave ? 64u : 256u, kSlotBits = kOneWave ? 6u : 8u;
    const unsigned int n_items = kSlots * (unsigned int) (a.spp_end - a.spp_begin);
    const float t_min = consts<float>::tmin;          // box culling; primitive tests take consts<R>::tmin
    const Prim<R>* __restrict__ fprims = sc.fprims;
    const int trav_keep = (int) a.trav_keep, node_keep = (int) a.node_keep, regen_min = (int) a.regen_min;

    unsigned long long wave_rays = 0, wave_rejected = 0;   // wave-uniform (scalar registers): counted with a ballot at points where the wave has reconverged
    unsigned long long wave_untraced = 0;                  // of wave_rays: segments finished without a traversal (zero-direction rays of an OBJ-loader scene, below)
    VisitStats vs; vs.n_box = 0; vs.n_leaf = 0; vs.hash = 0; vs.w_node = vs.w_leaf = vs.w_bounce = 0; vs.n_zero = 0;

    bool alive = false, finished = a.max_depth <= 0 || n_items == 0, poisoned = false;
    int depth = 0, local = 0;
    Rng g; g.s = 0;
    Ray<R> ray;
    vec3<R> beta;          // the path's throughput.  No running radiance: a material that emits does not scatter (material.hpp:156-159,
                           // :215), so the only non-zero term of L = sum beta_i * emission_i is the last hit's
    Trav tv; tv.cur = RT_BVH_SENTINEL; tv.sp = stack_base; tv.hit = -1; tv.closest = 0.f; tv.slack = 0.f; tv.sg = 0u; tv.sel[0] = tv.sel[1] = tv.sel[2] = 0x03020100u;
    Closest<R> best;
    if constexpr (sizeof(R) == 8) best.t = R(0);

    unsigned long long census[3] = { 0, 0, 0 };
    unsigned long long cyc[4] = { 0, 0, 0, 0 }, t0 = kCount ? wall_clock64() : 0;   // diagnostic: wave time per phase (regenerate, nodes, leaves, shade)
    auto lap = [&](int phase) { if (kCount) { const unsigned long long t1 = wall_clock64(); cyc[phase] += t1 - t0; t0 = t1; } };

    for (;;) {
        // ---------------------------------------------------------------- regenerate
        const unsigned long long want = __ballot(!alive && !finished);
        const unsigned long long have = __ballot(alive);
        if (want && (have == 0 || __popcll(want) >= regen_min)) {
            RT_ASM_MARK("; regenerate begin");
            if (!alive && !finished) {
                // one LDS atomic per wave: the first fetching lane reserves a run of items for all fetching lanes; item =
                // sample * (slots of the workgroup) + slot, so simultaneous fetches are neighbouring pixels of one sample
                const int leader = __ffsll(want) - 1;
                unsigned int base = 0;
                if (lane == leader) base = atomicAdd(item_counter, (unsigned int) __popcll(want));
                base = __shfl(base, leader, 64);
                // fetching lanes below this one: v_mbcnt counts them (the mask (1 << lane) - 1 would be a 64-bit loop invariant held per lane)
                const unsigned int item = base + __builtin_amdgcn_mbcnt_hi((unsigned int) (want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int) want, 0u));
                if (item >= n_items) finished = true;
                else {
                    local = (int) (item & (kSlots - 1u));
                    const int k = a.spp_begin + (int) (item >> kSlotBits);
                    int j, i;
                    pixel_of(local, j, i);
                    if (j < a.width && i < a.y_end) {                                         // else: slot outside the image (edge tiles)
                        g.seed(a.seed, (uint32_t) (i * a.width + j), (uint32_t) k);
                        R x = (R(j) + uniform<R>(g)) / R(a.wm1);                                // raytracing.cpp:102-103
                        R y = (R(i) + uniform<R>(g)) / R(a.hm1);
                        ray = camera_emit(sc.cam, g, x, y);
                        beta = vec3<R>(R(1), R(1), R(1));
                        depth = a.max_depth;
                        poisoned = false;
                        alive = true;
                        trav_begin(tv, best, ray, stack_base);
                    }
                }
            }
        }
        RT_ASM_MARK("; regenerate end");
        if (__ballot(!finished) == 0) break;
        const int n_alive = __popcll(__ballot(alive));
        lap(0);

        // ---------------------------------------------------------------- traverse
        for (;;) {
            const int n_trav = __popcll(__ballot(tv.cur != RT_BVH_SENTINEL));
            if (n_trav == 0 || n_trav * 256 < n_alive * trav_keep) break;
            // inner nodes: while enough of the traversing lanes stand on one
            for (;;) {
                const int n_inner = __popcll(__ballot(tv.cur >= 0));
                if (n_inner == 0 || n_inner * 256 < n_trav * node_keep) break;
                if (kCount) {   // census of the lanes that sit this node step out: on a leaf / hit known / no path
                    const unsigned c0 = (unsigned) __popcll(__ballot(tv.cur < 0 && tv.cur != RT_BVH_SENTINEL));
                    const unsigned c1 = (unsigned) __popcll(__ballot(alive && tv.cur == RT_BVH_SENTINEL));
                    const unsigned c2 = (unsigned) __popcll(__ballot(!alive));
                    census[0] += c0; census[1] += c1; census[2] += c2;
                }
                if (tv.cur >= 0) node_step<R, kQuant, kCount, kSorted, kStride>(sc, tv, t_min, stack_at, vs);
            }
            lap(1);
            // leaves
            if (tv.cur < 0 && tv.cur != RT_BVH_SENTINEL) {
                RT_ASM_MARK("; leaf step begin");
                const int code = ~tv.cur, first = code >> 3, count = (code & 7) + 1;
                int k0 = 0;
                if constexpr (class_is_obj(kFull) && sizeof(R) == sizeof(float)) {
                    // triangle scenes, fp32: the first two records of the leaf (leaves hold at most two triangles there, rtcore.hip) are read
                    // together, so a lane with two triangles waits for memory once: leaf share of the wave time 18 -> 15 %, mesh 4K -1.9 % time.
                    // Not for R = double: two 96-byte records in flight cost the fp64 variant scratch and gave 42.70 -> 42.71 ms.
                    Prim<R> rec0 = *at32(fprims, (uint32_t) first * (uint32_t) sizeof(Prim<R>));
                    Prim<R> rec1 = *at32(fprims, (uint32_t) (first + (count > 1 ? 1 : 0)) * (uint32_t) sizeof(Prim<R>));
                    asm volatile("" : "+v"(rec0.g[0]), "+v"(rec0.g[4]), "+v"(rec0.g[8]), "+v"(rec0.id), "+v"(rec1.g[0]), "+v"(rec1.g[4]), "+v"(rec1.g[8]), "+v"(rec1.id));
                    R t;
                    if (kCount && first_active_lane()) vs.w_leaf++;
                    if (kCount) vs.n_leaf += (uint32_t) (count > 1 ? 2 : 1);
                    if (prim_t<R, kFull>(rec0, sc.side(), ray, consts<R>::tmin, closest_of(tv, best), t, &g) && take_hit(sc, t, closest_of(tv, best), tv.hit, rec0.id)) { set_closest(tv, best, t); tv.hit = first; }
                    if (__ballot(count > 1)) {
                        if (kCount && first_active_lane()) vs.w_leaf++;
                        if (count > 1 && prim_t<R, kFull>(rec1, sc.side(), ray, consts<R>::tmin, closest_of(tv, best), t, &g) && take_hit(sc, t, closest_of(tv, best), tv.hit, rec1.id)) { set_closest(tv, best, t); tv.hit = first + 1; }
                    }
                    k0 = 2;                                   // (a leaf longer than that -- RT_BVH_LEAF -- goes on below)
                }
                for (int k = k0; k < count; k++) {
                    R t;
                    if (kCount && first_active_lane()) vs.w_leaf++;
                    if (kCount) vs.n_leaf++;
                    // the whole 48-byte record in one round trip (three 16-byte loads issued together): left to itself the compiler
                    // sinks each field's load into the branch that first needs it -- id, then the edges, then the vertex: three
                    // dependent memory latencies per triangle (mesh workload -4.6 % time, leaf share of the wave time 21 -> 13 %)
                    Prim<R> rec = *at32(fprims, (uint32_t) (first + k) * (uint32_t) sizeof(Prim<R>));
                    asm volatile("" : "+v"(rec.g[0]), "+v"(rec.g[4]), "+v"(rec.g[8]), "+v"(rec.id));
                    if (prim_t<R, kFull>(rec, sc.side(), ray, consts<R>::tmin, closest_of(tv, best), t, &g) && take_hit(sc, t, closest_of(tv, best), tv.hit, rec.id)) { set_closest(tv, best, t); tv.hit = first + k; }
                }
                tv.cur = (int32_t) stack_at(tv.sp); tv.sp -= kStride;
                RT_ASM_MARK("; leaf step end");
            }
            lap(2);
        }

        // ---------------------------------------------------------------- shade: trace() body for lanes whose hit is known
        const bool shading = alive && tv.cur == RT_BVH_SENTINEL;
        wave_rays += (unsigned long long) __popcll(__ballot(shading));
        if (kCount && shading && ray.d.x == R(0) && ray.d.y == R(0) && ray.d.z == R(0)) vs.n_zero++;
        bool rejected_now = false, zero_ray = false;
        if (shading) {
            int path_rays = a.max_depth - depth + 1;                     // this path's segments so far, this one included
            vec3<R> L;
            if (kCount && first_active_lane()) vs.w_bounce++;
            bool done = true;
            if (tv.hit >= 0) {
                Hit<R> h;
                RT_ASM_MARK("; shade: record");
                fill_record<R, kFull>(fprims[tv.hit], sc.side(), ray, closest_of(tv, best), sc.need_uv != 0, h);
                const Mat<R>& m = sc.mats[h.mat];
                const vec3<R> texc = m.tex >= 0 ? tex_color<R, class_tex(kFull)>(sc, m.tex, h.pos, h.u, h.v) : vec3<R>();
                vec3<R> emission = mat_emit<R, class_mat(kFull)>(m, h, texc);
                Ray<R> wo; wo.time = R(0);                                                  // `ray wo;` raytracing.cpp:41
                vec3<R> att;
                bool spec = false;
                const bool fixed = (a.flags & RT_FLAG_COMPAT_FIXED) != 0;
                RT_ASM_MARK("; shade: scatter");
                if (mat_shading<R, class_mat(kFull)>(g, m, ray, h, texc, wo, att, spec, fixed)) {
                    R pdf, mpdf;
                    bool dropped = false;
                    RT_ASM_MARK("; shade: pdf");
                    if (!(fixed && spec)) {
                        if (spec) {                                                         // raytracing.cpp:51-54
                            mpdf = mat_pdf<R, class_mat(kFull)>(g, m, h, wo);
                            pdf = mpdf;
                        }
                        else {                                                              // raytracing.cpp:55-61
                            if (uniform<R>(g) < R(0.5)) { wo.o = h.pos; wo.from = h.id; wo.d = light_rand<R, class_light(kFull)>(sc, g, h.pos); wo.time = ray.time; }
                            const R wo_len = length(wo.d);                                  // one root for the material's pdf and the lights'
                            mpdf = mat_pdf<R, class_mat(kFull)>(g, m, h, wo, wo_len, fixed);
                            pdf = R(0.5) * (lights_pdf<R, class_light(kFull)>(sc, h.pos, wo.d, wo_len) + mpdf);
                        }
                        // RT_FLAG_COMPAT_FIXED: a direction of zero probability contributes nothing (the reference forms 0 * inf)
                        if (fixed && (!(pdf > R(0)) || !(mpdf > R(0)))) dropped = true;
                        R inv = R(1) / pdf;                                                 // raytracing.cpp:69, vec3.hpp:121-123
                        vec3<R> w = mpdf * att;
                        att = inv * w;
                    }                                                                       // RT_FLAG_COMPAT_FIXED: a delta lobe weighs its attenuation
                    RT_ASM_MARK("; shade: tail");
                    // `dropped` goes through a register here on purpose.  Left as a lane mask, hipcc 7.2 -O2/-O3 compiled the float
                    // spheres-and-rects variant so that paths whose earlier bounce was specular were not dropped at a later diffuse
                    // bounce (RT_FLAG_COMPAT_FIXED: +32 % rays on the fixture scene, fireflies on the metal sphere); -O1, the other
                    // variants and R = double were right.  Found by the per-pixel ray counts against the fp64 mode, which
                    // test_fast_ray_counts_match_fp64_per_pixel now also holds under RT_FLAG_COMPAT_FIXED.
                    { int dropped_reg = dropped; asm volatile("" : "+v"(dropped_reg)); dropped = dropped_reg != 0; }
                    if (!dropped) {
                    beta = beta * att;
                    // a non-finite 1/pdf or weight (also 0 * inf) poisons the sample whatever follows: the recursion multiplies it in
                    if (!rt_finite(beta.x) || !rt_finite(beta.y) || !rt_finite(beta.z)) poisoned = true;
                    ray = wo;
                    depth--;
                    done = depth <= 0;
                    if (poisoned && (a.flags & RT_FLAG_EARLY_EXIT)) done = true;
                    // phong leaves `wo` unset (material.hpp:225-229,252-254): a ray with direction (0,0,0), a fifth of the OBJ workload's rays.  The
                    // reference traces it all the same, and against triangles it can only miss (polygon.hpp:95-97: a = e1 . (0 x e2) = 0; its
                    // box walk only enters the boxes around the origin).  In a scene of triangles alone that segment is therefore finished
                    // here: counted as the ray it is, black, the path ends (rejected: its pdf is 0).  Before, the slab test's nudge of zero
                    // components made it a diagonal ray that walked the whole tree: mesh workload 152 -> 69 ms per 64 spp.
                    // (Spheres answer such a ray with NaN roots: the other variants walk on.)
                    if (class_is_obj(kFull) && !done && ray.d.x == R(0) && ray.d.y == R(0) && ray.d.z == R(0)) { zero_ray = true; done = true; path_rays++; }
                    }
                }
                else L = beta * emission;                                                   // the path ends on an emitter (or absorber)
            }
            if (done) {
                alive = false;
                bool bad = poisoned || !rt_finite(L.x) || !rt_finite(L.y) || !rt_finite(L.z);  // raytracing.cpp:108-110
                rejected_now = bad;
                if (a.flags & RT_FLAG_DEBUG_RAY_COUNTS) { L = vec3<R>(R(path_rays), bad ? R(1) : R(0), R(0)); bad = false; }
                if (!bad && (L.x != R(0) || L.y != R(0) || L.z != R(0))) {     // most paths end black (a miss, the depth limit): nothing to add
                    if constexpr (kGlobalSums) {
                        int pj, pi;
                        pixel_of(local, pj, pi);
                        R* dst = accum + ((size_t) pi * a.width + pj) * 3;
                        unsafeAtomicAdd(dst, L.x); unsafeAtomicAdd(dst + 1, L.y); unsafeAtomicAdd(dst + 2, L.z);     // global_atomic_add_f64, no return
                    }
                    else {
                    R* dst = tile_sum + local * 3;
                    __hip_atomic_fetch_add(dst, L.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 1, L.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 2, L.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            else trav_begin(tv, best, ray, stack_base);
        }
        RT_ASM_MARK("; shade: end");
        { const unsigned long long nz = (unsigned long long) __popcll(__ballot(zero_ray)); wave_rays += nz; wave_untraced += nz; }
        if (kCount && zero_ray) vs.n_zero++;
        wave_rejected += (unsigned long long) __popcll(__ballot(rejected_now));
        lap(3);
    }

    if constexpr (!class_is_obj(kFull)) {
        // A wave whose tile has no sample left for it ENDS here, without waiting for the others: its registers and its wave slot are free
        // at once, and as soon as four slots of the CU are (and LDS allows) the next workgroup starts -- instead of a tile's last long
        // paths holding four wave slots until the slowest of them is done.  The tile's sums are written by whichever wave finishes last.
        // Measured at 4K, fp64: random spheres -0.6 % (16 spp) / -1.3 % (128 spp) time, deep bounce -2.3 % / -1.6 %.  Not for the OBJ-loader
        // variants: their traversal stack fills the LDS three workgroups share, so a freed wave slot lets nobody in, and the single-wave
        // write-out cost the mesh 0.6 % at every sample count.
        const bool last_wave = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");               // this wave's additions to the tile sums come before its count
            unsigned int n = 0;
            if (lane == 0) n = atomicAdd(item_counter + 1, 1u);
            n = (unsigned int) __builtin_amdgcn_readfirstlane((int) n);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            return n == 3u;
        }();
        if (last_wave) {
#pragma unroll 1
            for (int slot = lane; slot < 256; slot += 64) {
                int j, i;
                pixel_of(slot, j, i);
                if (j < a.width && i < a.y_end) {
                    R* out = accum + ((size_t) i * a.width + j) * 3;
                    const R sx = tile_sum[slot * 3], sy = tile_sum[slot * 3 + 1], sz = tile_sum[slot * 3 + 2];
                    if (a.flags & RT_FLAG_ACCUMULATE) { out[0] += sx; out[1] += sy; out[2] += sz; }
                    else { out[0] = sx; out[1] = sy; out[2] = sz; }
                }
            }
        }
    }
    else if constexpr (!kGlobalSums) {
        __syncthreads();
        int j, i;
        pixel_of((int) threadIdx.x, j, i);
        if (j < a.width && i < a.y_end) {
            R* out = accum + ((size_t) i * a.width + j) * 3;
            const R sx = tile_sum[threadIdx.x * 3], sy = tile_sum[threadIdx.x * 3 + 1], sz = tile_sum[threadIdx.x * 3 + 2];
            if (a.flags & RT_FLAG_ACCUMULATE) { out[0] += sx; out[1] += sy; out[2] += sz; }
            else { out[0] = sx; out[1] = sy; out[2] = sz; }
        }
    }

    // wave-level reduction of the counters, one atomic pair per wave
    if (lane == 0) {
        atomicAdd(&counters[0], wave_rays);
        if (wave_rejected) atomicAdd(&counters[1], wave_rejected);
        if (wave_untraced) atomicAdd(&counters[15], wave_untraced);
        if (a.unit_cost) atomicAdd(&a.unit_cost[unit], (unsigned int) (wave_rays >> 6) + 1u);      // in units of 64 rays: a super-tile of a long frame stays below 2^32
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
            for (int k = 0; k < 3; k++) atomicAdd(&counters[12 + k], census[k]);
        }
    }
}

template <typename R, bool kCount, bool kQuant, int kFull>
__global__ void __launch_bounds__(256, (wave_kernel_waves<R, kFull>())) render_wave_kernel(Scene<R> sc, RenderArgs a, R* __restrict__ accum, unsigned long long* __restrict__ counters)
{
    render_wave_body<R, kCount, kQuant, kFull, false>(sc, a, accum, counters);
}
// the same kernel over wide nodes whose hit children are visited by entry distance (Scene::sorted_visit: sparse soups)
template <typename R, bool kCount, int kFull>
__global__ void __launch_bounds__(256, (sizeof(R) == 8 && kFull == 3 ? RT_WAVES_FP64_SORTED : wave_kernel_waves<R, kFull>())) render_wave_sorted_kernel(Scene<R> sc, RenderArgs a, R* __restrict__ accum, unsigned long long* __restrict__ counters)
{
    render_wave_body<R, kCount, true, kFull, true>(sc, a, accum, counters);
}
// ... and their one-wave-workgroup twins (kernels without tile sums: wave_global_sums), for launches of few samples per pixel
template <typename R, bool kCount, bool kQuant, int kFull>
__global__ void __launch_bounds__(64, (wave_kernel_waves<R, kFull>())) render_wave1_kernel(Scene<R> sc, RenderArgs a, R* __restrict__ accum, unsigned long long* __restrict__ counters)
{
    render_wave_body<R, kCount, kQuant, kFull, false, true>(sc, a, accum, counters);
}
template <typename R, bool kCount, int kFull>
__global__ void __launch_bounds__(64, (sizeof(R) == 8 && kFull == 3 ? RT_WAVES_FP64_SORTED : wave_kernel_waves<R, kFull>())) render_wave1_sorted_kernel(Scene<R> sc, RenderArgs a, R* __restrict__ accum, unsigned long long* __restrict__ counters)
{
    render_wave_body<R, kCount, true, kFull, true, true>(sc, a, accum, counters);
}

// Closest hit of explicit rays through the fast tree, with the render kernel's own culling code (trav_begin, node_step: slab_fma / slab_wide,
// LDS stack) and primitive tests: test hook behind rt_first_hit_tree.  R = double must return the reference walk's primitive, t bits
// and frontward flag (the closest hit does not depend on the tree), which is what pins RT_MODE_FAST_FP64's traversal.
template <typename R, bool kQuant>
__global__ void __launch_bounds__(256) first_hit_tree_kernel(Scene<R> sc, const double* __restrict__ rays, int n, int32_t* __restrict__ prim,
                                                             R* __restrict__ t_out, int32_t* __restrict__ front)
{
    extern __shared__ __align__(16) unsigned char smem[];
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t stack_base = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) unsigned char*) smem + threadIdx.x * 4u;
    auto stack_at = [&](uint32_t lds_address) -> lds_u32& { return *(lds_u32*) (uintptr_t) lds_address; };
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;                                  // no barrier below
    stack_at(stack_base) = (uint32_t) RT_BVH_SENTINEL;
    const double* p = rays + (size_t) idx * 7;
    Ray<R> r;
    r.o = vec3<R>(R(p[0]), R(p[1]), R(p[2]));
    r.d = vec3<R>(R(p[3]), R(p[4]), R(p[5]));
    r.time = R(p[6]);
    const float t_min = consts<float>::tmin;
    Trav tv; Closest<R> best;
    VisitStats unused;
    trav_begin(tv, best, r, stack_base);
    while (tv.cur != RT_BVH_SENTINEL) {
        if (tv.cur >= 0) {
            if (kQuant && sc.sorted_visit) node_step<R, kQuant, false, kQuant>(sc, tv, t_min, stack_at, unused);      // the visiting order the scene's renders use
            else node_step<R, kQuant, false, false>(sc, tv, t_min, stack_at, unused);
        }
        else {
            const int code = ~tv.cur, first = code >> 3, count = (code & 7) + 1;
            for (int k = 0; k < count; k++) {
                R t;
                if (prim_t<R, 2>(sc.fprims[first + k], sc.side(), r, consts<R>::tmin, closest_of(tv, best), t, nullptr) && take_hit(sc, t, closest_of(tv, best), tv.hit, sc.fprims[first + k].id)) { set_closest(tv, best, t); tv.hit = first + k; }
            }
            tv.cur = (int32_t) stack_at(tv.sp); tv.sp -= 1024u;
        }
    }
    int f = -1;
    if (tv.hit >= 0) {
        Hit<R> h;
        fill_record<R, 2>(sc.fprims[tv.hit], sc.side(), r, closest_of(tv, best), false, h);
        f = h.front ? 1 : 0;
    }
    prim[idx] = tv.hit >= 0 ? sc.fprims[tv.hit].id : -1;
    t_out[idx] = tv.hit >= 0 ? closest_of(tv, best) : R(0);
    front[idx] = f;
}

}  // namespace rt
