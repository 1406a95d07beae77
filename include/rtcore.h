/* include/rtcore.h -- C ABI of the MI355X path-tracing core (librtcore.so).
 *
 * Drop-in boundary for ONE hot path of ccronix/raytracing: the per-pixel bounce loop
 *   render_image pixel/sample loop   src/cpp/raytracing.cpp:96-123
 *   trace()                          src/cpp/raytracing.cpp:30-70
 *   group/bvh/aabb/primitive intersect  group.hpp:68-83, bvh.hpp:58-68, aabb.hpp:23-41,
 *                                    primitive.hpp:29-56,103-130,176-319, polygon.hpp:90-124
 *   material shading / pdf / emit    material.hpp:20-312, texture.hpp:21-133
 *   light sampling                   group.hpp:104-119, primitive.hpp:258-274, polygon.hpp:140-167
 *   camera::emit                     camera.hpp:62-68
 * The reference has no FFI: its boundary is the C++ `scene` object handed to trace().  Host code
 * (raytracing_amd/host, source-compatible with the reference's scene/object/material classes) flattens a
 * built `scene` into the plain-C `rt_scene_desc` below and calls these entry points; see INTEGRATION.md.
 *
 * Conventions: every function returns RT_OK (0) or a negative RT_E* code and never exits or throws;
 * rt_last_error() gives the message of the calling thread's last failure.  Pointers named *_dev are device
 * pointers on the current HIP device; all others are host pointers.  The library keeps no global state
 * except the per-thread error string; a scene belongs to the device that was current at rt_scene_create.
 */
#ifndef RTCORE_H
#define RTCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK          0
#define RT_EINVAL     -1   /* bad argument / malformed scene description */
#define RT_EHIP       -2   /* a HIP runtime call failed */
#define RT_ENOMEM     -3
#define RT_EUNSUPPORTED -4

/* primitive kinds: the reference's leaf object classes */
#define RT_PRIM_SPHERE   1  /* sphere   primitive.hpp:13-75    data: c[3], radius                       */
#define RT_PRIM_MSPHERE  2  /* msphere  primitive.hpp:78-155   data: c0[3], c1[3], start, end, radius   */
#define RT_PRIM_PLANEXY  3  /* planexy  primitive.hpp:158-206  data: x0,x1,y0,y1,k                      */
#define RT_PRIM_PLANEXZ  4  /* planexz  primitive.hpp:209-275  data: x0,x1,z0,z1,k                      */
#define RT_PRIM_PLANEYZ  5  /* planeyz  primitive.hpp:278-326  data: y0,y1,z0,z1,k                      */
#define RT_PRIM_TRIANGLE 6  /* triangle polygon.hpp:23-183     data: p0[3], p0p1[3], p0p2[3], n0[3], n1[3], n2[3], uv0[2], uv1[2], uv2[2] */
#define RT_PRIM_BOX      7  /* box      primitive.hpp:399-458, optionally inside rotatey and then translate (transform.hpp:9-121), as
                               cornell_box() builds it (demo.hpp:80-88): one leaf of the bvh.
                               data: minimum[3], maximum[3], sin_theta, cos_theta, target[3]                  */
#define RT_PRIM_VOLUME   8  /* volume   primitive.hpp:329-396: a constant-density medium inside a boundary object; its material must be
                               RT_MAT_FOG.  Boundary a sphere: data c[3], radius, neg_inv_d (= -1 / density).  Boundary a box
                               (RT_PRIM_FLAG_BOUNDARY_BOX), optionally inside rotatey and then translate as RT_PRIM_BOX takes them
                               (volume(translate(rotatey(box)))): data minimum[3], maximum[3], sin_theta, cos_theta, target[3], neg_inv_d.
                               The free-flight distance is drawn inside the intersection test (:373): such a primitive is never hit by
                               rt_first_hit / rt_first_hit_fast, which have no random stream, and RT_FLAG_WAVEFRONT rejects scenes that
                               contain one                                                                                              */
#define RT_PRIM_STRIDE  24  /* doubles per primitive record */
#define RT_XFORM_STRIDE  6  /* doubles per instance transform */
#define RT_PRIM_FLAG_FLIP 1 /* wrapped in flip (transform.hpp:124-149): toggles `frontward` of the leaf's own record, i.e. BEFORE the record passes
                               through the wrappers of prim_xform (whose set_face_normal calls then overwrite it, as in the reference) */
#define RT_PRIM_FLAG_ROTATEY 2   /* RT_PRIM_BOX (or a volume's boundary box): wrapped in rotatey (sin_theta, cos_theta as its constructor computes them) */
#define RT_PRIM_FLAG_TRANSLATE 4 /* RT_PRIM_BOX (or a volume's boundary box): wrapped (outermost) in translate(target) */
#define RT_PRIM_FLAG_BOUNDARY_BOX 8  /* RT_PRIM_VOLUME: the boundary is a box (see above), not a sphere */
#define RT_PRIM_FLAG_FLIP_OUTER 16   /* a flip OUTSIDE every wrapper of prim_xform (flip(translate(x)), flip(bvh) around an instance): toggles
                                        `frontward` after the record has come back out of the whole chain */
#define RT_XFORM_FLAG_CHAIN 8    /* instance transform record: the NEXT record (index + 1) continues this transform towards the object */

/* material kinds (material.hpp) and the layout of their RT_MAT_STRIDE doubles */
#define RT_MAT_LAMBERTIAN 1 /* tex = base_color                                  */
#define RT_MAT_METAL      2 /* tex = base_color; d[0] = roughness (already <= 1) */
#define RT_MAT_DIELECTRIC 3 /* d[0] = ior                                        */
#define RT_MAT_EMISSIVE   4 /* tex = emission                                    */
#define RT_MAT_PHONG      5 /* tex = diffuse_map or -1; d[1..3] Kd, d[4..6] Ks, d[7..9] Ke, d[10..12] Tr, d[13] Ns, d[14] Ni */
#define RT_MAT_FOG        6 /* tex = base_color (material.hpp:165-181); used by RT_PRIM_VOLUME */
#define RT_MAT_STRIDE    16

/* texture kinds (texture.hpp) and the layout of their RT_TEX_STRIDE doubles */
#define RT_TEX_CONSTANT 1   /* d[0..2] rgb                                                        */
#define RT_TEX_CHECKER  2   /* d[0] even texture index, d[1] odd texture index                    */
#define RT_TEX_IMAGE    3   /* d[0] width, d[1] height, d[2] channels, d[3] color_based, d[4] byte offset into tex_pixels */
#define RT_TEX_STRIDE    8

typedef struct rt_scene rt_scene;   /* opaque: a scene resident in HBM */

/* A built scene, flattened.  Caller keeps ownership; rt_scene_create copies everything it needs.
 * The BVH is the reference's own tree (bvh.hpp:22-56), node by node: child code c >= 0 is node c, c < 0 is
 * primitive ~c.  `roots` lists scene.objects' children in order (one entry after scene::build_bvh, scene.hpp:46-51).
 * A `group` used as an object inside the tree (a leaf of a bvh, a member of another group; group.hpp:68-83) is a chain of GROUP LINKS:
 * a node whose box is inverted (min[0] > max[0]; the flattener writes +inf / -inf) has no box test -- its two children are
 * intersected in order with the running t_max, which is what group::intersect's loop does; members m0..mk become
 * link(m0, link(m1, ... link(mk-1, mk))), a group of one is its member.
 * Primitive ids are insertion order into the scene, before the builder permutes the array. */
typedef struct rt_scene_desc {
    int32_t n_nodes;  const double* node_bbox;   /* n_nodes x 6: min xyz, max xyz */
                      const int32_t* node_child; /* n_nodes x 2: left, right      */
    int32_t n_roots;  const int32_t* roots;
    int32_t n_prims;  const int32_t* prim_kind; const int32_t* prim_flags; const int32_t* prim_mat;
                      const double* prim_data;   /* n_prims x RT_PRIM_STRIDE */
    int32_t n_mats;   const int32_t* mat_kind;  const int32_t* mat_tex; const double* mat_data; /* n_mats x RT_MAT_STRIDE */
    int32_t n_tex;    const int32_t* tex_kind;  const double* tex_data;                          /* n_tex x RT_TEX_STRIDE  */
    size_t n_pixel_bytes; const uint8_t* tex_pixels;  /* image texels, stb layout (row-major, interleaved channels) */
    /* scene.lights (scene.hpp:16): separate object instances, same record layout as primitives */
    int32_t n_lights; const int32_t* light_kind; const int32_t* light_flags; const int32_t* light_mat;
                      const double* light_data;  /* n_lights x RT_PRIM_STRIDE */
    /* camera after setup_camera (camera.hpp:46-60): position[3], left_bottom[3], horizontal[3], vertical[3],
     * u[3], v[3], w[3], lens_radius, shutter_start, shutter_end */
    double camera[24];
    /* instances: a bvh, a group or a single primitive wrapped in rotatey and / or translate (transform.hpp:9-121), as all_feature_test()
     * builds (demo.hpp:136-140).  Members keep their object-space geometry -- node boxes and primitive records -- and name their
     * transform; rays are taken into that frame for every test and hit records come back out through every wrapper's
     * set_face_normal call, innermost first.  One record holds translate(rotatey(.)), rotatey(.) or translate(.); any other nesting --
     * translate(translate(.)), rotatey(translate(.)), an instance inside an instance -- is a CHAIN of consecutive records, outermost
     * first, each but the last flagged RT_XFORM_FLAG_CHAIN; node_xform / prim_xform name a chain's first record.  RT_PRIM_BOX may keep
     * its innermost translate(rotatey(.)) in its own record (prim_xform -1 when that is all).  n_xforms == 0: the two index arrays may
     * be NULL. */
    int32_t n_xforms; const double* xform_data;  /* n_xforms x RT_XFORM_STRIDE: sin_theta, cos_theta, target[3], flags
                                                     (RT_PRIM_FLAG_ROTATEY | RT_PRIM_FLAG_TRANSLATE | RT_XFORM_FLAG_CHAIN; translate is the outer one) */
                      const int32_t* node_xform; /* n_nodes: transform index, -1 = world space */
                      const int32_t* prim_xform; /* n_prims: transform index, -1 = world space */
} rt_scene_desc;

/* precision / traversal mode of rt_render */
#define RT_MODE_REFERENCE_FP64 0 /* fp64, no FMA contraction, the reference's tree in the reference's visit order */
#define RT_MODE_FAST_FP32      1 /* fp32, the library's own SAH tree with ordered traversal (same closest hits, ~4x fewer
                                    tests than the reference tree); accumulation buffer is fp32 */
#define RT_MODE_FP32_REFERENCE_TREE 2 /* fp32 arithmetic over the reference's tree in the reference's visit order (A/B baseline) */
#define RT_MODE_FAST_FP64      3 /* the reference's arithmetic -- fp64, no FMA contraction, its evaluation order, for rays, primitive
                                    tests, hit records, shading and sums -- behind the library's SAH tree: boxes only cull (fp32,
                                    conservative), they never decide a hit, so closest hits, t bits and radiance are those of
                                    RT_MODE_REFERENCE_FP64.  Two surfaces at the same t (bit for bit): the reference keeps the one its
                                    walk visits last, and so does this mode (ties are broken by the primitives' rank in the reference's
                                    visit order; the one known residue: walls shared by two neighbouring `box` primitives).  Volumes draw
                                    random numbers inside the traversal, so their streams follow the visit order.  accum_dev is double;
                                    sums are added per tile (per pixel for scenes of triangles alone) in completion order, so they match the
                                    reference mode's to rounding of the additions, not bit for bit.  Zero-direction rays (phong) against spheres
                                    follow the reference's NaN slabs only in RT_MODE_REFERENCE_FP64. */

#define RT_FLAG_ACCUMULATE 1     /* add this call's sample sums into accum_dev instead of overwriting it */
#define RT_FLAG_EARLY_EXIT 2     /* stop a path as soon as its sample is known to be rejected (NaN/inf, raytracing.cpp:108-110).
                                    Same image; fewer rays than the reference traces.  Off by default. */

#define RT_FLAG_COUNT_TESTS 4    /* diagnostic: also count box and leaf tests (slower kernel variant); fills stats.box_tests / leaf_tests */
#define RT_FLAG_DEBUG_RAY_COUNTS 8 /* diagnostic: accum_dev receives per-pixel (rays, rejected samples, 0) instead of radiance sums */
#define RT_FLAG_LEGACY_LOOP 16   /* diagnostic (A/B): RT_MODE_FAST_FP32 with the lane-synchronous bounce loop (every lane traverses to completion,
                                    then every lane shades) instead of the per-wave traverse / shade / regenerate state machine */
#define RT_FLAG_WAVEFRONT 32     /* RT_MODE_FAST_FP32 as a multi-kernel wavefront pipeline over a path pool in HBM (rt_wavefront.h) instead of
                                    one kernel; blocks until the samples are done.  Slower than the default on the measured workloads. */

#define RT_FLAG_QUEUE 64         /* A/B: the fast modes through the decoupled kernel (rt_queue.h: per-wave path pools in global memory, lanes hand
                                    finished traversals to full-width shade passes and take queued rays at once).  Same samples and sums as the
                                    default kernel up to the order of the tile additions.  Scenes with boxes, volumes or instances ignore it.
                                    The pools are allocated on a scene's first such render and freed by rt_scene_destroy: 2048 pools x 1024 slots
                                    x 64 B = 128 MiB for RT_MODE_FAST_FP32, x 128 B = 256 MiB for RT_MODE_FAST_FP64. */

#define RT_FLAG_COMPAT_FIXED 128 /* No counterpart in the reference (SURVEY 7 / 9 "compat=fixed"): the reference's estimator with three defects repaired,
                                    for pictures rather than parity.  (1) A specular lobe is a delta and weighs its attenuation: the reference divides by
                                    material::pdf, which is 0 for a dielectric and for a metal reflection below the surface, so every glass sample
                                    is 0 * inf and rejected; a specular direction below the surface is absorbed.  (2) phong sets wo: mirror
                                    reflection for its specular lobe, a cosine lobe for its diffuse one, whose pdf is not re-rolled.  (3) A
                                    diffuse direction of zero probability ends the path with nothing instead of 0/0.  Checked against the oracle's
                                    restatement of the same rules on the same streams (orc_set_compat_fixed).  Wave kernel and reference-order kernels;
                                    RT_FLAG_QUEUE / RT_FLAG_WAVEFRONT / RT_FLAG_LEGACY_LOOP with it are refused. */

typedef struct rt_render_params {
    int32_t width, height;       /* image size: pixel (j,i) -> sample position as raytracing.cpp:102-103 */
    int32_t y_begin, y_end;      /* rows rendered by this call (tile split); 0, height for the whole frame */
    int32_t spp_begin, spp_end;  /* global sample indices rendered by this call (spp split)  */
    int32_t max_depth;           /* raytracing.cpp:90 */
    int32_t mode;                /* RT_MODE_* */
    int32_t flags;               /* RT_FLAG_* */
    uint64_t seed;               /* key of the counter-based RNG: stream = f(seed, pixel, sample) */
    void* stream;                /* hipStream_t, NULL = default stream */
} rt_render_params;

typedef struct rt_render_stats {
    uint64_t rays;               /* trace() invocations past the depth check (raytracing.cpp:32-37): what the reference counts as a ray */
    uint64_t rejected_samples;   /* samples zeroed by the NaN/inf test (raytracing.cpp:108-110) */
    double kernel_ms;            /* HIP-event time of the render kernel on params.stream */
    uint64_t box_tests;          /* aabb::intersect calls (aabb.hpp:23-41), only with RT_FLAG_COUNT_TESTS */
    uint64_t leaf_tests;         /* primitive intersect calls, only with RT_FLAG_COUNT_TESTS */
    uint64_t wave_node_steps;    /* RT_FLAG_COUNT_TESTS, fast mode: wave-level iterations of the node loop (x64 lanes = issue slots) */
    uint64_t wave_leaf_steps;    /* ... of the leaf-test loop */
    uint64_t wave_bounces;       /* ... of the bounce loop */
    uint64_t phase_ticks[4];     /* RT_FLAG_COUNT_TESTS, fast mode: wave time (100 MHz ticks, summed over waves) in the regenerate, node, leaf and shade phases */
    uint64_t zero_direction_rays; /* RT_FLAG_COUNT_TESTS: of `rays`, those traced with direction (0,0,0): phong leaves `wo` unset (material.hpp:225-229,
                                    252-254), the reference traces that ray all the same (it misses everything) and so counts it */
    uint64_t untraced_rays;      /* always counted: of `rays`, the segments the fast kernels finished WITHOUT a BVH traversal -- the zero-direction
                                    rays above in a scene of triangles alone, where such a ray can only miss (polygon.hpp:95-97).  rays - untraced_rays
                                    is the number of rays that walked the tree (SURVEY 8d's metric).  0 in the reference-order modes, which walk them */
} rt_render_stats;

/* Copies the scene to HBM on the current device and builds the device layouts.  Replaces the implicit
 * `const scene&` argument of trace() (raytracing.cpp:30). */
int rt_scene_create(const rt_scene_desc* desc, rt_scene** scene_out);
int rt_scene_destroy(rt_scene* scene);

/* Renders samples [spp_begin, spp_end) of rows [y_begin, y_end): per pixel the SUM over those samples of
 * trace(emit(x, y)) with rejected samples zeroed -- the body of the k-loop at raytracing.cpp:101-113.
 * accum_dev: width*height*3 sums, row i = image row i as in the reference (row 0 is the bottom of the PNG),
 * double for RT_MODE_REFERENCE_FP64 and RT_MODE_FAST_FP64, float for the fp32 modes.  It must be ordinary device memory (hipMalloc / a torch
 * CUDA tensor), not host-mapped or fine-grained memory: for scenes of triangles alone the fast modes add each finished path into it with
 * the GPU's floating-point atomics (the rows are cleared first unless RT_FLAG_ACCUMULATE).  One kernel, asynchronous on params->stream unless
 * `stats` is non-NULL, in which case the call synchronises the stream and fills *stats (= rt_render_wait).  The scene's device must be
 * the current one.  A scene has ONE set of counters and events: a second rt_render before the first one's rt_render_wait is refused
 * (RT_EINVAL).  With RT_FLAG_WAVEFRONT the call
 * runs a loop of extend / shade kernels over a path pool (84 B per slot, RT_WF_SLOTS slots, default 2 Mi; allocated on the
 * scene's first such call, freed by rt_scene_destroy), polls a device flag and returns when the samples are done. */
int rt_render(rt_scene* scene, const rt_render_params* params, void* accum_dev, rt_render_stats* stats);

/* Waits for the scene's outstanding rt_render (one launched with stats == NULL) and fills *stats (may be NULL: wait only).  Lets one
 * host thread start renders on several devices and collect them afterwards. */
int rt_render_wait(rt_scene* scene, rt_render_stats* stats);

/* The name (as rocprofv3 lists it, without the argument list) of the kernel an rt_render of this scene with this mode and flags launches:
 * the render kernels are compiled in variants by scene content and node format, and a profile is only comparable with a run of the same
 * variant.  (Scenes of triangles alone: a launch of at most 32 samples per pixel takes that kernel's one-wave-workgroup twin, named
 * render_wave1_... instead of render_wave_...)  No counterpart in the reference.  Host only. */
int rt_render_kernel_name(rt_scene* scene, int mode, int flags, char* out, size_t out_bytes);

/* One process, several GPUs (no counterpart in the reference, which knows one device): the spp split of SURVEY 8e at the C boundary.
 * Per device: rt_device_set(d), rt_scene_create (the scene is replicated), a buffer, rt_render of that device's share of the global sample
 * indices with stats == NULL; then rt_render_wait on each and ONE rt_frame_reduce: dst_dev (on dst_device) += every src_dev[i] (on
 * src_device[i]), n_values = width*height*3 sums.  The sources reach the destination by peer copies, each on a stream of its own so that
 * the (up to 7) copies run together, every one over its source's own xGMI link (the links are point to point); peer access is enabled
 * once per device pair where the platform offers it (hipDeviceCanAccessPeer), otherwise the copy goes through the host.  The additions
 * follow on one stream as the copies land.  The call waits for all work on the devices involved before it starts and returns when the
 * sum is complete.  Sources on the destination's own device are allowed.  Staging buffers (one per source in flight, n_values each),
 * streams and events are kept per destination device between calls -- a frame is reduced once per frame -- until
 * rt_frame_reduce_release() frees them (safe to call at any time; the next reduce allocates again).
 * (Across processes -- one rank per GPU -- bench.py sums the same buffers with an RCCL reduce instead: raytracing_amd/parallel.py.) */
int rt_device_count(int* count_out);
int rt_device_set(int device);
int rt_frame_reduce(void* dst_dev, int dst_device, void* const* src_dev, const int* src_device, int n_src, int accum_is_fp64, size_t n_values);
int rt_frame_reduce_release(void);

/* mean, gamma 0.45, clamp, truncate to 8 bit: raytracing.cpp:115-121.  rgb_dev: n_pixels*3 bytes. */
int rt_resolve(const void* accum_dev, int accum_is_fp64, size_t n_pixels, int spp, uint8_t* rgb_dev, void* stream);

/* Parity / debug entry: closest hit of explicit rays (origin[3], direction[3], time; t_min 1e-3, t_max inf as
 * ray.hpp:12-13) through scene.objects.intersect (group.hpp:68-83 -> bvh.hpp:58-68), fp64, reference visit order.
 * Outputs (host, any may be NULL): primitive id or -1, t, frontward flag, number of box tests, number of leaf
 * tests, FNV-1a hash of the visited pre-order entry indices.  Synchronous.  Like rt_render, the first-hit entry points need the
 * scene's device to be the current one (RT_EINVAL otherwise). */
int rt_first_hit(rt_scene* scene, const double* rays, size_t n_rays, int32_t* prim, double* t, int32_t* front,
                 uint32_t* n_box, uint32_t* n_leaf, uint64_t* visit_hash);

/* rt_first_hit for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc / hipFree).  No
 * counterpart in the reference.  Same arguments, same kernel and bit-identical outputs; the device-side staging of rays and results
 * (88 bytes per ray) lives in a static device array of the library (RT_NOALLOC_MAX_RAYS rays = 2.75 MiB per GPU the library runs on,
 * held until the process ends) instead of hipMalloc'd memory: the call allocates and frees nothing.  It is SYNCHRONOUS like
 * rt_first_hit -- blocking copies from and to the caller's host arrays, the kernel on the null stream, a device synchronise -- so it
 * is not for use under stream capture.  More than RT_NOALLOC_MAX_RAYS rays -> RT_EINVAL.  The scene's device must be the current one.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two host threads) race; serialise them. */
#define RT_NOALLOC_MAX_RAYS 32768
int rt_first_hit_noalloc(rt_scene* scene, const double* rays, size_t n_rays, int32_t* prim, double* t, int32_t* front,
                         uint32_t* n_box, uint32_t* n_leaf, uint64_t* visit_hash);

/* Closest hit of the same rays through RT_MODE_FAST_FP32's tree (fp32): primitive id or -1, and t.  No counterpart in
 * the reference; lets tests check that the fast tree returns the reference's hits. */
int rt_first_hit_fast(rt_scene* scene, const double* rays, size_t n_rays, int32_t* prim, float* t);

/* Closest hit of the same rays through the fast tree with the render kernels' own culling code: fp64 != 0 is RT_MODE_FAST_FP64's
 * traversal (primitive, t bits and frontward flag must equal rt_first_hit's), fp64 == 0 RT_MODE_FAST_FP32's (t widened to double).
 * Walks the quantised four-wide nodes when the scene's renders do.  No counterpart in the reference.  Synchronous. */
int rt_first_hit_tree(rt_scene* scene, const double* rays, size_t n_rays, int fp64, int32_t* prim, double* t, int32_t* front);

/* Device-buffer helpers for hosts that do not otherwise touch HIP (the C++ driver). */
int rt_buffer_alloc(size_t bytes, void** dev_out);
int rt_buffer_free(void* dev);
int rt_buffer_download(void* host_dst, const void* dev_src, size_t bytes);
int rt_buffer_upload(void* dev_dst, const void* host_src, size_t bytes);

/* Binary scene cache (no counterpart in the reference, whose only scene source is OBJ text through tinyobj and
 * per-triangle `new`, loader.hpp:27,124): writes / reads a flattened scene verbatim, so a 10 M-triangle scene is parsed and its
 * reference BVH built once.  rt_scene_desc_load returns a description whose arrays live in ONE allocation owned by the
 * library; free it with rt_scene_desc_free (after rt_scene_create, which copies what it needs).  Host only: no GPU needed. */
int rt_scene_desc_save(const rt_scene_desc* desc, const char* path);
int rt_scene_desc_load(const char* path, rt_scene_desc** desc_out);
int rt_scene_desc_free(rt_scene_desc* desc);

/* Progressive-accumulation checkpoint: the sample sums of a frame and how far it got.  A render is resumed by
 * loading the sums and calling rt_render with spp_begin = *spp_done and RT_FLAG_ACCUMULATE; because RNG streams are keyed by
 * the global sample index, the resumed frame equals the uninterrupted one (bit for bit in RT_MODE_REFERENCE_FP64).
 * rt_checkpoint_load fails with RT_EINVAL when the file's geometry (width, height, precision) differs from the arguments. */
int rt_checkpoint_save(const char* path, const void* accum_dev, int accum_is_fp64, int32_t width, int32_t height, int32_t spp_done, uint64_t seed);
int rt_checkpoint_load(const char* path, void* accum_dev, int accum_is_fp64, int32_t width, int32_t height, int32_t* spp_done, uint64_t* seed);

/* Test tooling for the fp64-vs-CPU comparison (tools/first_divergence.py; no counterpart in the reference).  rt_debug_path: the path of ONE
 * stream (pixel row i, column j, global sample k) through the reference-order walk with the render kernel's own functions -- per segment a row
 * of 9 doubles: ray origin, direction, time, hit distance (0: miss), primitive id (-1); *n_out = segments traced (rows beyond `cap` are not
 * stored).  rt_debug_libm: the device library's results for n calls (op 1 sin, 2 cos, 3 acos, 4 atan2(x, y), 5 pow(x, y), 6 log; 7: the sign of sin(x) as the checker texture takes it -- -1, 0, +1, or 2 for not-a-number; 8: x^5 as the dielectric's reflectance takes it), so that a
 * CPU path's libm calls can be replayed on the GPU and compared bit for bit.  Both synchronous, both allocate. */
int rt_debug_path(rt_scene* scene, int32_t width, int32_t height, int32_t i, int32_t j, int32_t k, uint64_t seed, int32_t max_depth, int32_t flags,
                  double* rows9_host, int32_t cap, int32_t* n_out_host);
int rt_debug_libm(const int32_t* op, const double* x, const double* y, int32_t n, double* out_host);

/* n draws of the counter-based stream (seed, pixel, sample), computed on the device (test hook for the RNG) */
int rt_debug_rng(uint64_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out_host);

const char* rt_last_error(void);
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif
