// raytracing_amd/csrc/rtcore.hip -- C ABI (include/rtcore.h): scene upload, launches, resolve.
//
// HBM layout built here from rt_scene_desc (all arrays are flat, index-linked; no pointers inside records):
//   ops   : the reference BVH in pre-order, one record per bvh node / leaf reference (Op<R>, 32 B fp32 / 56 B fp64)
//   prims : hot intersection records by primitive id (Prim<R>, 48 B fp32 / 96 B fp64)
//   tris  : cold per-triangle shading records (normals + uvs), only when the scene has triangles
//   mats, texs, pixels, lights, light_tris
// Both precisions are resident so one scene serves the parity entry points and the fast renderer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rt_bvh_build.h"
#include "rt_launch.h"
#include "rtcore.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg)
{
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return fail(RT_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

template <typename R> struct DeviceScene {
    rt::Scene<R> view{};
    std::vector<void*> owned;
    uint32_t ops_bytes = 0, prims_bytes = 0;   // padded to 16
    std::vector<uint32_t> leaf_order;          // fp32 scene, during rt_scene_create only: primitive ids in the fast tree's leaf order
};

template <typename T> int upload(std::vector<void*>& owned, const std::vector<T>& host, const T** dev, size_t pad_to = 16)
{
    size_t bytes = host.size() * sizeof(T);
    size_t padded = ((bytes + pad_to - 1) / pad_to) * pad_to;
    if (padded == 0) padded = pad_to;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, padded));
    owned.push_back(p);
    HIP_TRY(hipMemset(p, 0, padded));
    if (bytes) HIP_TRY(hipMemcpy(p, host.data(), bytes, hipMemcpyHostToDevice));
    *dev = static_cast<const T*>(p);
    return RT_OK;
}

inline float down(double v) { float f = (float) v; return (double) f > v ? std::nextafterf(f, -INFINITY) : f; }
inline float up(double v) { float f = (float) v; return (double) f < v ? std::nextafterf(f, INFINITY) : f; }
template <typename R> R box_lo(double v);
template <typename R> R box_hi(double v);
template <> double box_lo<double>(double v) { return v; }
template <> double box_hi<double>(double v) { return v; }
template <> float box_lo<float>(double v) { return down(v); }   // fp32 boxes contain the fp64 boxes
template <> float box_hi<float>(double v) { return up(v); }

}  // namespace

// ---- rt_first_hit_noalloc: device-side staging that lives in the code object instead of in hipMalloc'd memory.
// Layout: n rays x 7 doubles, then the six output arrays in rt_first_hit's order of decreasing alignment (t, hash, prim, front,
// n_box, n_leaf).  In host code the array's NAME is only a handle for hipGetSymbolAddress: the address a kernel may use is the
// one that call returns, per device (every GPU the library runs on holds its own copy of the array until the process ends).
#define RT_NOALLOC_BYTES_PER_RAY (7 * 8 + 8 + 8 + 4 + 4 + 4 + 4)
static __device__ __align__(256) unsigned char scratch[(size_t) RT_NOALLOC_MAX_RAYS * RT_NOALLOC_BYTES_PER_RAY];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = { nullptr };

// device address of `scratch` on the CURRENT device, resolved once per device
int scratch_address(int device, void** out)
{
    if (device < 0 || device >= kMaxDevices) return fail(RT_EUNSUPPORTED, "device index beyond the library's per-device table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail(RT_EHIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}
}  // namespace

struct rt_scene {
    int device = 0;
    DeviceScene<double> d64;
    DeviceScene<float> d32;
    int32_t n_lights = 0, n_prims = 0, n_ops = 0;
    bool has_volume = false;                  // RT_PRIM_VOLUME present: needs the path's random stream during traversal
    bool has_box = false;                     // RT_PRIM_BOX present
    bool only_spheres = true;                 // no triangle, no phong, no image texture, rect lights only (fog only with volumes): the lean variants 5 / 6 / 7
    bool only_triangles = true;               // every primitive is an RT_PRIM_TRIANGLE, every material an RT_MAT_PHONG, no checker texture (an OBJ-loader scene)
    unsigned long long* counters = nullptr;   // [0] rays, [1] rejected samples, [2] box tests, [3] leaf tests
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool pending = false;                     // an rt_render without stats has been launched and not yet waited for
    // rt_queue.h: path pools (per precision) and their locks, allocated on the scene's first decoupled render
    void* q_pools[2] = { nullptr, nullptr };
    unsigned int* q_locks[2] = { nullptr, nullptr };
    uint32_t q_n_pools = 2048;                // >= workgroups the chip can hold at once: 256 CUs x at most 8 workgroups of 256 threads (8 waves per SIMD),
                                              // so a workgroup's try-lock walk over the pools always finds a free one
    uint32_t queue = 0, q_slots = 192, q_handoff_min = 16, q_shade_starved = 16;   // RT_QUEUE / RT_Q_SLOTS / RT_Q_HANDOFF / RT_Q_STARVED
    rt::WfBuffers wf{};                       // RT_MODE_FAST_FP32 path pool, allocated on first use
    std::vector<void*> wf_owned;
    uint32_t wf_slots = 2u << 20;             // RT_WF_SLOTS overrides
    uint32_t trav_keep = 96, node_keep = 128, regen_min = 16;
    uint32_t xcd_tiles = 0;                   // super-tiles dealt to the XCDs (rt_wave.h): on when tree + records exceed one XCD's L2; RT_XCD_TILES overrides
    uint32_t quantised = 0;                   // fast mode walks the quantised four-wide nodes (RT_QNODES overrides)   // rt_wave.h phase thresholds (RT_TRAV_KEEP / RT_NODE_KEEP / RT_REGEN_MIN override)
    // cost-ordered launch of the wave kernel (rt_device.h RenderArgs::unit_order): per scheduling unit (super-tile or tile) the rays of the
    // first render of a frame geometry, then the units sorted by them, dearest first; kept for every later render of that geometry
    uint32_t* unit_cost = nullptr; uint32_t* unit_order = nullptr;
    uint32_t n_units = 0;
    int32_t order_key[5] = { 0, 0, 0, 0, 0 };  // width, height, y_begin, y_end, xcd_tiles the buffers were made for
    int order_state = 0;                      // 0: nothing yet, 1: the outstanding (or last waited) render is collecting costs, 2: unit_order is valid
    bool tile_order = true;                   // RT_TILE_ORDER=0: launch in image order (A/B)
    uint32_t stack_pad = 0;                   // RT_STACK_PAD: stack levels reserved beyond what the tree needs (test hook for the > 48 KiB LDS launches)
    size_t lds_limit = 0;   // reference-order program is read from global memory unless RT_LDS_LIMIT allows staging it in LDS
};

namespace {

struct Flat {   // validated, precision-independent view of the description
    std::vector<int32_t> op_prim, op_skip;
    std::vector<double> op_box;   // 6 per op
};

int validate(const rt_scene_desc* d)
{
    if (!d) return fail(RT_EINVAL, "desc is NULL");
    if (d->n_nodes < 0 || d->n_prims <= 0 || d->n_mats <= 0 || d->n_tex < 0 || d->n_lights < 0 || d->n_roots <= 0)
        return fail(RT_EINVAL, "scene needs at least one primitive, material and root");
    if ((d->n_nodes && (!d->node_bbox || !d->node_child)) || !d->roots || !d->prim_kind || !d->prim_flags || !d->prim_mat || !d->prim_data ||
        !d->mat_kind || !d->mat_tex || !d->mat_data || (d->n_tex && (!d->tex_kind || !d->tex_data)) ||
        (d->n_lights && (!d->light_kind || !d->light_flags || !d->light_data)))
        return fail(RT_EINVAL, "a required array of the scene description is NULL");
    auto code_ok = [&](int32_t c) { return c >= 0 ? c < d->n_nodes : (~c) < d->n_prims; };
    for (int i = 0; i < d->n_roots; i++) if (!code_ok(d->roots[i])) return fail(RT_EINVAL, "root code out of range");
    for (int i = 0; i < 2 * d->n_nodes; i++) if (!code_ok(d->node_child[i])) return fail(RT_EINVAL, "bvh child code out of range");
    for (int i = 0; i < d->n_prims; i++) {
        if (d->prim_kind[i] < RT_PRIM_SPHERE || d->prim_kind[i] > RT_PRIM_VOLUME) return fail(RT_EINVAL, "unknown primitive kind");
        if (d->prim_mat[i] < 0 || d->prim_mat[i] >= d->n_mats) return fail(RT_EINVAL, "primitive material index out of range");
        const int fl = d->prim_flags[i];
        if (fl & ~(RT_PRIM_FLAG_FLIP | RT_PRIM_FLAG_ROTATEY | RT_PRIM_FLAG_TRANSLATE | RT_PRIM_FLAG_BOUNDARY_BOX | RT_PRIM_FLAG_FLIP_OUTER)) return fail(RT_EINVAL, "unknown primitive flag");
        if (d->prim_kind[i] == RT_PRIM_VOLUME) {
            if (d->mat_kind[d->prim_mat[i]] != RT_MAT_FOG) return fail(RT_EINVAL, "a volume takes a fog material (primitive.hpp:332-344)");
            if ((fl & (RT_PRIM_FLAG_ROTATEY | RT_PRIM_FLAG_TRANSLATE)) && !(fl & RT_PRIM_FLAG_BOUNDARY_BOX))
                return fail(RT_EINVAL, "a volume's rotatey / translate flags describe its boundary BOX (RT_PRIM_FLAG_BOUNDARY_BOX)");
        }
        else if (fl & RT_PRIM_FLAG_BOUNDARY_BOX) return fail(RT_EINVAL, "RT_PRIM_FLAG_BOUNDARY_BOX is a flag of RT_PRIM_VOLUME");
        if ((fl & (RT_PRIM_FLAG_ROTATEY | RT_PRIM_FLAG_TRANSLATE)) && d->prim_kind[i] != RT_PRIM_BOX && d->prim_kind[i] != RT_PRIM_VOLUME)
            return fail(RT_EINVAL, "only a box (or a volume's boundary box) carries rotatey / translate in its own record: use prim_xform");
    }
    if (d->n_xforms < 0 || d->n_xforms > 65534 || (d->n_xforms && (!d->xform_data || !d->prim_xform || (d->n_nodes && !d->node_xform))))
        return fail(RT_EINVAL, "instance transforms: 0..65534 of them, with xform_data, node_xform and prim_xform present");
    for (int i = 0; i < d->n_xforms; i++) {
        const int fl = (int) d->xform_data[(size_t) i * RT_XFORM_STRIDE + 5];
        if ((fl & (RT_PRIM_FLAG_ROTATEY | RT_PRIM_FLAG_TRANSLATE)) == 0 || (fl & ~(RT_PRIM_FLAG_ROTATEY | RT_PRIM_FLAG_TRANSLATE | RT_XFORM_FLAG_CHAIN)))
            return fail(RT_EINVAL, "instance transform: wrappers must be ROTATEY and / or TRANSLATE (plus RT_XFORM_FLAG_CHAIN)");
        if ((fl & RT_XFORM_FLAG_CHAIN) && i + 1 == d->n_xforms) return fail(RT_EINVAL, "instance transform: the last record cannot continue (RT_XFORM_FLAG_CHAIN)");
    }
    if (d->n_xforms) {
        for (int i = 0; i < d->n_nodes; i++) if (d->node_xform[i] < -1 || d->node_xform[i] >= d->n_xforms) return fail(RT_EINVAL, "node_xform out of range");
        for (int i = 0; i < d->n_prims; i++) {
            if (d->prim_xform[i] < -1 || d->prim_xform[i] >= d->n_xforms) return fail(RT_EINVAL, "prim_xform out of range");
        }
    }
    for (int i = 0; i < d->n_lights; i++) {
        if (d->light_kind[i] < RT_PRIM_SPHERE || d->light_kind[i] > RT_PRIM_BOX) return fail(RT_EINVAL, "unknown light kind (a volume cannot be a light)");
        if (d->light_mat && (d->light_mat[i] < 0 || d->light_mat[i] >= d->n_mats)) return fail(RT_EINVAL, "light material index out of range");
    }
    if (d->n_pixel_bytes >= (1ull << 32)) return fail(RT_EUNSUPPORTED, "tex_pixels beyond 4 GiB (texel offsets are 32-bit on the device)");
    for (int i = 0; i < d->n_mats; i++) {
        int k = d->mat_kind[i], t = d->mat_tex[i];
        if (k < RT_MAT_LAMBERTIAN || k > RT_MAT_FOG) return fail(RT_EINVAL, "unknown material kind");
        bool needs_tex = k == RT_MAT_LAMBERTIAN || k == RT_MAT_METAL || k == RT_MAT_EMISSIVE || k == RT_MAT_FOG;
        if (t >= d->n_tex || (needs_tex && t < 0)) return fail(RT_EINVAL, "material texture index out of range");
    }
    for (int i = 0; i < d->n_tex; i++) {
        int k = d->tex_kind[i];
        const double* td = d->tex_data + (size_t) i * RT_TEX_STRIDE;
        if (k == RT_TEX_CHECKER) {
            // !(x >= 0 && x < n) also catches NaN; a fractional value would truncate to some other texture
            for (int k = 0; k < 2; k++)
                if (!(td[k] >= 0 && td[k] < d->n_tex) || td[k] != std::floor(td[k])) return fail(RT_EINVAL, "checker texture index out of range or not an integer");
        }
        else if (k == RT_TEX_IMAGE) {
            double w = td[0], h = td[1], ch = td[2], off = td[4];
            // 1 or 2 channels (a grey PNG as stb hands it over): the reference reads base[0..2] whatever `channels` is (texture.hpp:115-122),
            // i.e. the texel and its right-hand neighbours; so does the device, over a pool padded so the last texel's read stays inside
            if (w > 0 && !(ch >= 1 && ch <= 4 && ch == std::floor(ch))) return fail(RT_EINVAL, "image texture: 1..4 channels expected");
            if (w > 0 && (!(h > 0) || !(off >= 0) || w != std::floor(w) || h != std::floor(h) || off != std::floor(off) ||
                          off + w * h * ch > (double) d->n_pixel_bytes || !d->tex_pixels))
                return fail(RT_EINVAL, "image texture exceeds tex_pixels");
        }
        else if (k != RT_TEX_CONSTANT) return fail(RT_EINVAL, "unknown texture kind");
    }
    return RT_OK;
}

// pre-order program of the reference tree; iterative so a degenerate tree cannot overflow the host stack
int linearize(const rt_scene_desc* d, Flat& f)
{
    struct Frame { int32_t code; int stage; int op; };
    std::vector<Frame> stack;
    // a span-1 node names the same child twice (bvh.hpp:31-34) and the program repeats that sub-tree, as the reference walks it twice
    size_t guard = (size_t) d->n_nodes * 16 + (size_t) d->n_roots + 16;
    for (int r = 0; r < d->n_roots; r++) {
        stack.push_back({ d->roots[r], 0, -1 });
        while (!stack.empty()) {
            Frame& fr = stack.back();
            if (fr.code < 0) {
                f.op_prim.push_back(~fr.code); f.op_skip.push_back(0);
                for (int k = 0; k < 6; k++) f.op_box.push_back(0.0);
                stack.pop_back();
                continue;
            }
            // a group link (rtcore.h: inverted box) has no box test and therefore no entry: its children follow one another, the running t_max
            // of the walk is group::intersect's `closest` (group.hpp:72-80)
            const bool link = d->node_bbox[(size_t) fr.code * 6] > d->node_bbox[(size_t) fr.code * 6 + 3];
            if (fr.stage == 0) {
                fr.op = link ? -1 : (int) f.op_prim.size();
                if (!link) {
                    f.op_prim.push_back(d->n_xforms && d->node_xform[fr.code] >= 0 ? -2 - d->node_xform[fr.code] : -1); f.op_skip.push_back(0);   // see rt_device.h closest_hit
                    for (int k = 0; k < 6; k++) f.op_box.push_back(d->node_bbox[(size_t) fr.code * 6 + k]);
                }
                fr.stage = 1;
                int32_t child = d->node_child[(size_t) fr.code * 2];
                stack.push_back({ child, 0, -1 });
            }
            else if (fr.stage == 1) {
                fr.stage = 2;
                int32_t child = d->node_child[(size_t) fr.code * 2 + 1];
                stack.push_back({ child, 0, -1 });
            }
            else {
                if (fr.op >= 0) f.op_skip[fr.op] = (int32_t) f.op_prim.size();
                stack.pop_back();
            }
            if (f.op_prim.size() > guard) return fail(RT_EINVAL, "bvh links are not a tree (shared node or cycle)");
        }
    }
    return RT_OK;
}

template <typename R> void fill_prim(rt::Prim<R>& p, rt::TriShade<R>* tri, int kind, int flags, int mat, const double* g, int id, int xform = -1)
{
    std::memset(&p, 0, sizeof p);
    p.kf = (int32_t) ((uint32_t) kind | ((uint32_t) (flags & 0xff) << 8) | ((uint32_t) (xform + 1) << 16)); p.mat = mat; p.id = id;
    const bool boxed = kind == RT_PRIM_BOX || (kind == RT_PRIM_VOLUME && (flags & RT_PRIM_FLAG_BOUNDARY_BOX));
    int n = kind == RT_PRIM_SPHERE ? 4 : kind == RT_PRIM_VOLUME ? 5 : kind == RT_PRIM_MSPHERE ? 9 : kind == RT_PRIM_TRIANGLE ? 9 : 5;
    if (boxed) n = 8;
    for (int k = 0; k < n; k++) p.g[k] = (R) g[k];
    if (boxed && !(flags & RT_PRIM_FLAG_ROTATEY)) { p.g[6] = R(0); p.g[7] = R(1); }
    if (kind == RT_PRIM_VOLUME && boxed) p.g[8] = (R) g[11];                             // neg_inv_d behind the boundary box's fields
    if (tri) {
        std::memset(tri, 0, sizeof *tri);
        if (boxed) for (int k = 0; k < 3; k++) tri->n[k] = (flags & RT_PRIM_FLAG_TRANSLATE) ? (R) g[8 + k] : R(0);   // translate target
        if (kind == RT_PRIM_TRIANGLE) {
            for (int k = 0; k < 9; k++) tri->n[k] = (R) g[9 + k];
            for (int k = 0; k < 6; k++) tri->uv[k] = (R) g[18 + k];
        }
    }
}

// bounds of a primitive over the time span its rays can carry (camera shutter, hull with the primitive's own span)
void prim_bounds(const rt_scene_desc* d, int i, rtb::Box& b)
{
    const double* g = d->prim_data + (size_t) i * RT_PRIM_STRIDE;
    auto sphere = [&](const double c[3], double r) { double lo[3] = { c[0] - r, c[1] - r, c[2] - r }, hi[3] = { c[0] + r, c[1] + r, c[2] + r }; b.grow(lo); b.grow(hi); };
    switch (d->prim_kind[i]) {
    case RT_PRIM_VOLUME: if (d->prim_flags[i] & RT_PRIM_FLAG_BOUNDARY_BOX) goto as_box;
        // fall through
    case RT_PRIM_SPHERE: sphere(g, std::fabs(g[3])); break;
    case RT_PRIM_MSPHERE: {
        double t0 = std::min(std::min(d->camera[22], d->camera[23]), std::min(g[6], 0.0));
        double t1 = std::max(std::max(d->camera[22], d->camera[23]), std::max(g[7], 0.0));
        for (double t : { t0, t1 }) {
            double f = (t - g[6]) / (g[7] - g[6]);
            if (!std::isfinite(f)) f = 0;
            double c[3] = { g[0] + f * (g[3] - g[0]), g[1] + f * (g[4] - g[1]), g[2] + f * (g[5] - g[2]) };
            sphere(c, std::fabs(g[8]));
        }
        break;
    }
    case RT_PRIM_BOX: as_box: {   // the eight corners, out of the box's frame: rotatey (transform.hpp:60-66) then translate
        const int fl = d->prim_flags[i];
        const double sn = (fl & RT_PRIM_FLAG_ROTATEY) ? g[6] : 0.0, cs = (fl & RT_PRIM_FLAG_ROTATEY) ? g[7] : 1.0;
        for (int c = 0; c < 8; c++) {
            const double x = (c & 1) ? g[3] : g[0], y = (c & 2) ? g[4] : g[1], z = (c & 4) ? g[5] : g[2];
            double p[3] = { cs * x + sn * z, y, -sn * x + cs * z };
            if (fl & RT_PRIM_FLAG_TRANSLATE) for (int k = 0; k < 3; k++) p[k] += g[8 + k];
            b.grow(p);
        }
        break;
    }
    case RT_PRIM_TRIANGLE:
        for (int v = 0; v < 3; v++) {
            double p[3];
            for (int k = 0; k < 3; k++) p[k] = g[k] + (v == 1 ? g[3 + k] : v == 2 ? g[6 + k] : 0.0);
            b.grow(p);
        }
        break;
    default: {
        int ax = d->prim_kind[i] == RT_PRIM_PLANEYZ ? 1 : 0, bx = d->prim_kind[i] == RT_PRIM_PLANEXY ? 1 : 2;
        int kx = 3 - ax - bx;
        double lo[3], hi[3];
        lo[ax] = g[0]; hi[ax] = g[1]; lo[bx] = g[2]; hi[bx] = g[3]; lo[kx] = g[4]; hi[kx] = g[4];
        b.grow(lo); b.grow(hi);
        break;
    }
    }
    if (d->n_xforms && d->prim_xform[i] >= 0) {   // member of an instance: the eight corners of its own box, out of the instance's frame --
        // through every record of the chain, innermost (the last) first
        int first = d->prim_xform[i], last = first;
        while ((int) d->xform_data[(size_t) last * RT_XFORM_STRIDE + 5] & RT_XFORM_FLAG_CHAIN) last++;
        for (int xi = last; xi >= first; xi--) {
            const double* x = d->xform_data + (size_t) xi * RT_XFORM_STRIDE;
            const int fl = (int) x[5];
            const double sn = (fl & RT_PRIM_FLAG_ROTATEY) ? x[0] : 0.0, cs = (fl & RT_PRIM_FLAG_ROTATEY) ? x[1] : 1.0;
            rtb::Box w;
            for (int c = 0; c < 8; c++) {
                const double px = (c & 1) ? b.hi[0] : b.lo[0], py = (c & 2) ? b.hi[1] : b.lo[1], pz = (c & 4) ? b.hi[2] : b.lo[2];
                double p[3] = { cs * px + sn * pz, py, -sn * px + cs * pz };
                if (fl & RT_PRIM_FLAG_TRANSLATE) for (int k = 0; k < 3; k++) p[k] += x[2 + k];
                w.grow(p);
            }
            b = w;
        }
    }
    // Pad flat boxes (rects, axis-aligned triangles: zero extent along one axis).  The boxes only cull and are rounded outward to fp32
    // afterwards, so the pad need not be the reference's (primitive.hpp:203 0.001, polygon.hpp:128 1e-7): it was 1e-4, a twentieth of an
    // edge of the 1.3 M-triangle mesh on every side -- 5 % more leaf tests and 3.1 % more frame time than with 1e-6 (RT_BOX_PAD, 4K fp64:
    // 43.82 ms at 1e-4, 42.49 at 1e-5, 42.45 at 1e-6, 42.91 at 0).
    for (int k = 0; k < 3; k++) {
        static const double pad_abs = std::getenv("RT_BOX_PAD") ? std::atof(std::getenv("RT_BOX_PAD")) : 1e-6;
        double pad = pad_abs + 1e-6 * std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k]));
        b.lo[k] -= pad; b.hi[k] += pad;
    }
}

// the record behind the wide nodes' unused child slots: a triangle of zero area, which no ray hits (polygon.hpp:95-97: a = 0); material and
// id 0 exist in every scene, so even a NaN ray that "hits" it (the reference's range test passes NaN) reads valid memory
template <typename R> rt::Prim<R> dummy_record()
{
    rt::Prim<R> p;
    std::memset(&p, 0, sizeof p);
    p.kf = RT_PRIM_TRIANGLE; p.mat = 0; p.id = 0;
    return p;
}

template <typename R> int build_fast_tree(const rt_scene_desc*, const Flat&, const std::vector<rt::Prim<R>>&, DeviceScene<R>& out, const DeviceScene<float>* tree);
// fp64 (RT_MODE_FAST_FP64): the fp32 scene's tree -- nodes only cull, in fp32, for both precisions -- with the fp64 primitive
// records in its leaf order.  Node arrays stay owned by the fp32 scene.
template <> int build_fast_tree<double>(const rt_scene_desc*, const Flat&, const std::vector<rt::Prim<double>>& prims, DeviceScene<double>& out, const DeviceScene<float>* tree)
{
    out.view.bnodes = nullptr; out.view.wnodes = nullptr; out.view.wide_stack = 0; out.view.fprims = nullptr; out.view.n_bnodes = 0; out.view.bvh_depth = 0; out.view.sorted_visit = 0;
    if (!tree || !tree->view.bnodes) return RT_OK;
    if ((tree->leaf_order.size() + 1) * sizeof(rt::Prim<double>) >= (1ull << 32))
        return fail(RT_EUNSUPPORTED, "fast-mode fp64 record array beyond 4 GiB (the kernel addresses records by 32-bit byte offsets)");
    std::vector<rt::Prim<double>> ordered(tree->leaf_order.size() + 1);
    for (size_t i = 0; i < tree->leaf_order.size(); i++) ordered[i] = prims[tree->leaf_order[i]];
    ordered.back() = dummy_record<double>();
    int rc;
    if ((rc = upload(out.owned, ordered, &out.view.fprims))) return rc;
    out.view.bnodes = tree->view.bnodes; out.view.wnodes = tree->view.wnodes; out.view.wide_stack = tree->view.wide_stack; out.view.n_bnodes = tree->view.n_bnodes; out.view.bvh_depth = tree->view.bvh_depth;
    out.view.sorted_visit = tree->view.sorted_visit;
    return RT_OK;
}
template <> int build_fast_tree<float>(const rt_scene_desc* d, const Flat& f, const std::vector<rt::Prim<float>>& prims, DeviceScene<float>& out, const DeviceScene<float>*)
{
    std::vector<rtb::Item> items(d->n_prims);
    for (int i = 0; i < d->n_prims; i++) {
        prim_bounds(d, i, items[i].box);
        for (int k = 0; k < 3; k++) items[i].c[k] = 0.5 * (items[i].box.lo[k] + items[i].box.hi[k]);
        items[i].id = i;
    }
    // A volume draws its free-flight distance at every test.  Where the reference tree names a volume more than once (both
    // children of a span-1 node, bvh.hpp:31-34: all_feature_test() under scene::build_bvh) the reference tests it that often per
    // ray, and so must this tree, or the medium would scatter less: such a volume enters the builder once per occurrence.
    {
        std::vector<int> seen(d->n_prims, 0);
        for (int32_t p : f.op_prim)
            if (p >= 0 && d->prim_kind[p] == RT_PRIM_VOLUME && seen[p]++ > 0) items.push_back(items[p]);
    }
    // cost of a primitive test relative to a node step, measured (4K sweeps): triangles 1.0 (mesh workload -2.8 % time against 1.5),
    // scenes with spheres 1.5 (their test solves the quadratic in fp64).  Depth limit 26: from the binary 32-byte nodes of round 2
    // (a 28-level stack was the deepest with which five workgroups shared a CU's LDS); kept, lower limits cost time (22: +4 %).
    bool all_triangles = true;
    for (int i = 0; i < d->n_prims; i++) all_triangles &= d->prim_kind[i] == RT_PRIM_TRIANGLE;
    // SAH bins per axis: 64 for triangle meshes (-5 % frame time against 16 on the 0.3 / 1.3 / 5.2 M-triangle meshes alike; 128 and more
    // gained nothing), 16 for scenes with spheres (64: +4 %)
    // leaves of at most two triangles: the same box and leaf tests per ray as with four (18.2 / 2.07 on the 1.3 M-triangle mesh), but the leaf
    // step's record loop ends for all lanes after two rounds: 4K fp64 42.56 ms at 4, 42.42 at 3, 41.84 at 2
    int leaf_max = all_triangles ? 2 : 4, depth_limit = 26, bins = all_triangles ? 64 : 16; double c_prim = all_triangles ? 1.0 : 1.5;
    if (const char* e = std::getenv("RT_BVH_BINS")) bins = std::atoi(e);
    if (const char* e = std::getenv("RT_BVH_LEAF")) leaf_max = std::atoi(e);
    if (const char* e = std::getenv("RT_BVH_CPRIM")) c_prim = std::atof(e);
    if (const char* e = std::getenv("RT_BVH_DEPTH")) depth_limit = std::max(8, std::atoi(e));
    rtb::Tree tree = rtb::Builder(items, leaf_max, c_prim, depth_limit, bins).build();
    if (std::getenv("RT_DEBUG")) fprintf(stderr, "[rtcore] fast tree: %zu nodes, depth %d (limit %d), leaf_max %d, c_prim %.2f\n", tree.nodes.size(), tree.depth, depth_limit, leaf_max, c_prim);
    if (tree.depth + 2 > RT_BVH_STACK_MAX) return fail(RT_EUNSUPPORTED, "fast-mode tree deeper than the traversal stack");
    if (tree.nodes.size() * sizeof(rt::BNode) >= (1ull << 32) || (tree.order.size() + 1) * sizeof(rt::Prim<float>) >= (1ull << 32) || tree.order.size() >= (1u << 28))
        return fail(RT_EUNSUPPORTED, "fast-mode tree or record array beyond 4 GiB (the kernel addresses records by 32-bit byte offsets)");
    std::vector<rt::BNode> nodes(tree.nodes.size());
    for (size_t i = 0; i < nodes.size(); i++) {
        std::memset(&nodes[i], 0, sizeof(rt::BNode));
        for (int k = 0; k < 3; k++) {
            nodes[i].lb[k] = down(tree.nodes[i].lbox.lo[k]); nodes[i].lb[3 + k] = up(tree.nodes[i].lbox.hi[k]);
            nodes[i].rb[k] = down(tree.nodes[i].rbox.lo[k]); nodes[i].rb[3 + k] = up(tree.nodes[i].rbox.hi[k]);
        }
        nodes[i].left = tree.nodes[i].left; nodes[i].right = tree.nodes[i].right;
    }
    // the same tree collapsed to quantised four-wide nodes (rt_device.h WNode): decoded boxes must contain the fp32 boxes above
    double radius = 0;
    for (const rt::BNode& n : nodes) for (int k = 0; k < 6; k++) { if (std::isfinite(n.lb[k])) radius = std::max(radius, (double) std::fabs(n.lb[k])); if (std::isfinite(n.rb[k])) radius = std::max(radius, (double) std::fabs(n.rb[k])); }
    const double pad = 8 * 1.2e-7 * std::max(radius, 1e-30);       // fp32 decode error of origin*idir - o*idir, in world units
    // stack budget: the LDS stack of a workgroup is one KiB per level, and four workgroups per CU (four waves per SIMD) must fit it -- 37
    // levels (35 entries) beside the 3 KiB of tile sums of the fp32 kernels; the OBJ-loader kernels keep no sums in LDS (rt_wave.h
    // wave_global_sums) and take 39 levels (37 entries).  The mesh workload's tree (binary depth 24) collapses freely into 34 entries; a
    // tighter budget keeps the tallest paths narrower (31: 27 % more nodes, +1.3 % time at equal occupancy -- and +27 % time on the
    // 10 M-triangle soup's deep tree, which still gains 6 % from 35 -> 37).
    int budget = all_triangles ? 37 : 35;
    if (const char* e = std::getenv("RT_WIDE_STACK")) budget = std::max(1, std::atoi(e));
    const rtb::WideTree wt = rtb::widen(tree, budget);
    const int32_t dummy_leaf = ~(int32_t) ((uint32_t) tree.order.size() * 8u);      // leaf of one record: the one after the last real record
    auto encode = [&](const rtb::WideTree& t) {
        std::vector<rt::WNode> enc(t.nodes.size());
        for (size_t i = 0; i < enc.size(); i++) {
            const rtb::WideNode& w = t.nodes[i];
            std::memset(&enc[i], 0, sizeof(rt::WNode));
            for (int a = 0; a < 3; a++) {
                double lo[4], hi[4];
                for (int k = 0; k < w.n; k++) { lo[k] = (double) down(w.box[k].lo[a]) - pad; hi[k] = (double) up(w.box[k].hi[a]) + pad; }
                rtb::quantise_axis_wide(lo, hi, w.n, enc[i].o[a], enc[i].lo[a], enc[i].hi[a]);
            }
            // an unused slot names the dummy leaf (one record that no ray can hit, appended to the record arrays below): its box decodes
            // inside out (lo 255, hi 0) and is missed, but should rounding ever let a ray into it, the ray tests a triangle of zero area
            // and walks on -- so the visit needs no validity test per child
            for (int k = 0; k < 4; k++) enc[i].child[k] = k < w.n ? w.child[k] : dummy_leaf;
            enc[i].n = (uint32_t) w.axis * 0x01010101u;      // as a v_perm_b32 selector: byte `axis` of its first operand, four times
            enc[i].pad[0] = (uint32_t) w.n;
        }
        return enc;
    };
    const std::vector<rt::WNode> wn = encode(wt);
    // visiting order of the wide nodes (rt_bvh_build.h visit_order_census; scenes of triangles alone -- the ones that walk these nodes): slot
    // order unless it costs this tree's incoherent rays more than 5 % more box tests than the distance order does.  The two measured points
    // -- mesh +1.8 % tests, -0.6 % time without the sort; 10 M-triangle soup +27 % tests, +9.6 % time -- put the break-even near +3 %
    // of the kernel's own rays; the census's rays are all bounced ones (no coherent camera rays), where slot order fares worst, hence 5.  RT_SORTED_VISIT=0/1 overrides.
    out.view.sorted_visit = 0;
    if (all_triangles && !wt.nodes.empty()) {
        double tests[2];
        auto hit = [&](int at, const double* o, const double* dir, double t_max, double* t_out) {
            const double* g = d->prim_data + (size_t) tree.order[at] * RT_PRIM_STRIDE;       // p0, p0p1, p0p2 (polygon.hpp:93-115)
            const double qx = dir[1] * g[8] - dir[2] * g[7], qy = dir[2] * g[6] - dir[0] * g[8], qz = dir[0] * g[7] - dir[1] * g[6];
            const double a = g[3] * qx + g[4] * qy + g[5] * qz;
            if (std::fabs(a) < 1e-7) return false;
            const double f = 1.0 / a, sx = o[0] - g[0], sy = o[1] - g[1], sz = o[2] - g[2];
            const double u = f * (sx * qx + sy * qy + sz * qz);
            if (u < 0) return false;
            const double kx = sy * g[5] - sz * g[4], ky = sz * g[3] - sx * g[5], kz = sx * g[4] - sy * g[3];
            const double v = f * (dir[0] * kx + dir[1] * ky + dir[2] * kz);
            if (v < 0 || u + v > 1) return false;
            const double tt = f * (g[6] * kx + g[7] * ky + g[8] * kz);
            if (tt < 1e-3 || tt > t_max) return false;
            *t_out = tt;
            return true;
        };
        auto start = [&](double u, double* o) {                // the centre of a primitive's box: where a bounced ray starts, give or take its size
            const rtb::Item& it = items[std::min((size_t) (u * (double) d->n_prims), (size_t) d->n_prims - 1)];
            for (int k = 0; k < 3; k++) o[k] = it.c[k];
        };
        rtb::visit_order_census(wt, 2048, start, hit, tests);
        out.view.sorted_visit = tests[0] > 1.05 * tests[1];
        if (std::getenv("RT_DEBUG")) fprintf(stderr, "[rtcore] visit order census: %.1f box tests per ray in slot order, %.1f by distance -> %s\n", tests[0], tests[1], out.view.sorted_visit ? "sorted" : "slot order");
    }
    if (const char* e = std::getenv("RT_SORTED_VISIT")) out.view.sorted_visit = std::atoi(e) != 0;
    if (std::getenv("RT_DEBUG")) fprintf(stderr, "[rtcore] wide tree: %zu nodes of 64 B, stack need %d entries\n", wn.size(), wt.stack_need);
    if (wt.stack_need + 2 > RT_BVH_STACK_MAX) return fail(RT_EUNSUPPORTED, "fast-mode wide tree needs a deeper traversal stack than RT_BVH_STACK_MAX");
    std::vector<rt::Prim<float>> ordered(tree.order.size() + 1);
    for (size_t i = 0; i < tree.order.size(); i++) ordered[i] = prims[tree.order[i]];
    ordered.back() = dummy_record<float>();
    out.leaf_order.assign(tree.order.begin(), tree.order.end());
    int rc;
    if ((rc = upload(out.owned, nodes, &out.view.bnodes))) return rc;
    if ((rc = upload(out.owned, wn, &out.view.wnodes))) return rc;
    out.view.wide_stack = wt.stack_need;
    if ((rc = upload(out.owned, ordered, &out.view.fprims))) return rc;
    out.view.n_bnodes = (int32_t) nodes.size();
    out.view.bvh_depth = tree.depth;
    return RT_OK;
}

template <typename R> int build_device(const rt_scene_desc* d, const Flat& f, DeviceScene<R>& out, const DeviceScene<float>* tree = nullptr)
{
    const int n_ops = (int) f.op_prim.size();
    std::vector<rt::Op<R>> ops(n_ops);
    for (int i = 0; i < n_ops; i++) {
        for (int k = 0; k < 3; k++) { ops[i].mn[k] = box_lo<R>(f.op_box[(size_t) i * 6 + k]); ops[i].mx[k] = box_hi<R>(f.op_box[(size_t) i * 6 + 3 + k]); }
        ops[i].skip = f.op_skip[i]; ops[i].prim = f.op_prim[i];
    }
    bool has_tri = false;
    for (int i = 0; i < d->n_prims; i++) has_tri |= d->prim_kind[i] == RT_PRIM_TRIANGLE || d->prim_kind[i] == RT_PRIM_BOX ||
                                                    (d->prim_kind[i] == RT_PRIM_VOLUME && (d->prim_flags[i] & RT_PRIM_FLAG_BOUNDARY_BOX));   // kinds with a side record
    bool light_tri = false;
    for (int i = 0; i < d->n_lights; i++) light_tri |= d->light_kind[i] == RT_PRIM_TRIANGLE || d->light_kind[i] == RT_PRIM_BOX;

    std::vector<rt::Prim<R>> prims(d->n_prims);
    std::vector<rt::TriShade<R>> tris(has_tri ? d->n_prims : 0);
    for (int i = 0; i < d->n_prims; i++)
        fill_prim<R>(prims[i], has_tri ? &tris[i] : nullptr, d->prim_kind[i], d->prim_flags[i], d->prim_mat[i], d->prim_data + (size_t) i * RT_PRIM_STRIDE, i,
                     d->n_xforms ? d->prim_xform[i] : -1);
    std::vector<rt::Prim<R>> lights(d->n_lights);
    std::vector<rt::TriShade<R>> ltris(light_tri ? d->n_lights : 0);
    for (int i = 0; i < d->n_lights; i++)
        fill_prim<R>(lights[i], light_tri ? &ltris[i] : nullptr, d->light_kind[i], d->light_flags[i], d->light_mat ? d->light_mat[i] : 0,
                     d->light_data + (size_t) i * RT_PRIM_STRIDE, i);
    // triangle::pdf's area (polygon.hpp:146) is a constant of the light: taken here, once, with the expressions the kernels used to
    // evaluate per shaded hit (products and sums unfused, IEEE root and division: the same bits)
    for (int i = 0; light_tri && i < d->n_lights; i++) {
#pragma clang fp contract(off)
        if (lights[i].kind() != RT_PRIM_TRIANGLE) continue;
        const rt::Prim<R>& p = lights[i];
        ltris[i].area = rt::length(rt::cross(rt::vec3<R>(p.g[3], p.g[4], p.g[5]), rt::vec3<R>(p.g[6], p.g[7], p.g[8]))) / R(2);
    }

    std::vector<rt::Mat<R>> mats(d->n_mats);
    for (int i = 0; i < d->n_mats; i++) {
        mats[i].kind = d->mat_kind[i]; mats[i].tex = d->mat_tex[i];
        for (int k = 0; k < RT_MAT_STRIDE; k++) mats[i].d[k] = (R) d->mat_data[(size_t) i * RT_MAT_STRIDE + k];
    }
    bool need_uv = false;
    std::vector<rt::Tex<R>> texs(d->n_tex);
    for (int i = 0; i < d->n_tex; i++) {
        const double* td = d->tex_data + (size_t) i * RT_TEX_STRIDE;
        rt::Tex<R>& t = texs[i];
        std::memset(&t, 0, sizeof t);
        t.kind = d->tex_kind[i];
        if (t.kind == RT_TEX_CONSTANT) { t.rgb[0] = (R) td[0]; t.rgb[1] = (R) td[1]; t.rgb[2] = (R) td[2]; }
        else if (t.kind == RT_TEX_CHECKER) { t.a = (int) td[0]; t.b = (int) td[1]; }
        else if (td[0] <= 0) { t.kind = RT_TEX_CONSTANT; t.rgb[0] = R(0); t.rgb[1] = R(1); t.rgb[2] = R(1); }   // an image that failed to load is cyan everywhere (texture.hpp:97-99)
        else { t.a = (int) td[0]; t.b = (int) td[1]; t.ch = (int) td[2]; t.color_based = td[3] != 0; t.offset = (uint32_t) td[4]; need_uv = true; }
    }
    std::vector<rt::Xform<R>> xforms(d->n_xforms);
    for (int i = 0; i < d->n_xforms; i++) {
        const double* x = d->xform_data + (size_t) i * RT_XFORM_STRIDE;
        xforms[i].flags = (int32_t) x[5];
        xforms[i].sn = (xforms[i].flags & RT_PRIM_FLAG_ROTATEY) ? (R) x[0] : R(0); xforms[i].cs = (xforms[i].flags & RT_PRIM_FLAG_ROTATEY) ? (R) x[1] : R(1);
        for (int k = 0; k < 3; k++) xforms[i].t[k] = (xforms[i].flags & RT_PRIM_FLAG_TRANSLATE) ? (R) x[2 + k] : R(0);
    }
    // texel bytes, behind a table of the 256 values a colour texel can decode to (texture.hpp:117-121: pow(byte / 255, 2.2) per channel):
    // computed HERE, with the host's pow -- the library the reference itself calls -- so a textured hit costs three loads instead of three
    // device pow() (~600 vector instructions a shade pass), and carries the CPU path's bits exactly.  Scene::pixels points behind the table.
    std::vector<uint8_t> pixels(256 * sizeof(R));
    {
        R lut[256];
        const R scale = R(1) / R(255);
        for (int b = 0; b < 256; b++) lut[b] = (R) std::pow(scale * R(b), R(2.2));
        std::memcpy(pixels.data(), lut, sizeof lut);
    }
    if (d->tex_pixels) pixels.insert(pixels.end(), d->tex_pixels, d->tex_pixels + d->n_pixel_bytes);
    pixels.insert(pixels.end(), 4, (uint8_t) 0);    // a 1- or 2-channel image's last texel is read as three bytes (validate())

    rt::Scene<R>& v = out.view;
    int rc;
    if ((rc = upload(out.owned, ops, &v.ops))) return rc;
    if ((rc = upload(out.owned, prims, &v.prims))) return rc;
    if (has_tri) { if ((rc = upload(out.owned, tris, &v.tris))) return rc; } else v.tris = nullptr;
    if ((rc = upload(out.owned, mats, &v.mats))) return rc;
    if ((rc = upload(out.owned, texs, &v.texs))) return rc;
    if ((rc = upload(out.owned, pixels, &v.pixels))) return rc;
    v.pixels += 256 * sizeof(R);                    // behind the decode table (the allocation stays in out.owned)
    if ((rc = upload(out.owned, lights, &v.lights))) return rc;
    if (d->n_xforms) { if ((rc = upload(out.owned, xforms, &v.xforms))) return rc; } else v.xforms = nullptr;
    if (light_tri) { if ((rc = upload(out.owned, ltris, &v.light_tris))) return rc; } else v.light_tris = nullptr;
    {   // when the reference's walk visits each primitive last (rt_device.h Scene::ref_rank)
        std::vector<uint32_t> rank(d->n_prims, 0u);
        for (int i = 0; i < n_ops; i++) if (f.op_prim[i] >= 0) rank[f.op_prim[i]] = (uint32_t) i;
        if ((rc = upload(out.owned, rank, &v.ref_rank))) return rc;
    }
    v.n_ops = n_ops; v.n_prims = d->n_prims; v.n_lights = d->n_lights; v.need_uv = need_uv;
    v.n_lights_r = R(d->n_lights); v.inv_n_lights = R(1) / R(d->n_lights);        // (no lights: inf, as the reference's 1.0 / 0; rt_render refuses such a scene)
    const double* c = d->camera;
    auto v3 = [&](int o) { return rt::vec3<R>((R) c[o], (R) c[o + 1], (R) c[o + 2]); };
    v.cam.pos = v3(0); v.cam.left_bottom = v3(3); v.cam.horizontal = v3(6); v.cam.vertical = v3(9); v.cam.u = v3(12); v.cam.v = v3(15);
    v.cam.lens_radius = (R) c[21]; v.cam.s0 = (R) c[22]; v.cam.s1 = (R) c[23];
    if ((rc = build_fast_tree<R>(d, f, prims, out, tree))) return rc;
    out.ops_bytes = (uint32_t) (((size_t) n_ops * sizeof(rt::Op<R>) + 15) / 16 * 16);
    out.prims_bytes = (uint32_t) (((size_t) d->n_prims * sizeof(rt::Prim<R>) + 15) / 16 * 16);
    return RT_OK;
}

template <typename R> void release(DeviceScene<R>& s)
{
    for (void* p : s.owned) (void) hipFree(p);
    s.owned.clear();
}

__global__ void resolve_kernel(const void* accum, int is_fp64, size_t n_values, double inv_spp, uint8_t* rgb)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    double c = is_fp64 ? static_cast<const double*>(accum)[i] : (double) static_cast<const float*>(accum)[i];
    c = inv_spp * c;                                  // color /= spp  (vec3.hpp:56-58)   raytracing.cpp:115
    c = pow(c, 0.45);                                 // gamma         raytracing.cpp:116
    c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);          // clamp         raytracing.cpp:117 (NaN falls through, as in util.hpp:119-130)
    rgb[i] = (uint8_t) (int) (c * 255);               // truncation    raytracing.cpp:119-121
}

template <typename T> __global__ void add_kernel(T* __restrict__ dst, const T* __restrict__ src, size_t n)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

__global__ void rng_kernel(uint64_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out)
{
    if (blockIdx.x || threadIdx.x) return;
    rt::Rng g; g.seed(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = g.next();
}

// path pool of the wavefront renderer (rt_wavefront.h): allocated once per scene
int wf_prepare(rt_scene* s)
{
    if (s->wf.pool.ctl) return RT_OK;
    rt::WfBuffers b{};
    const uint32_t cap = std::max(256u, s->wf_slots / 256u * 256u);
    auto dev_alloc = [&](void** ptr, size_t bytes) -> bool {
        if (hipMalloc(ptr, bytes) != hipSuccess) return false;
        s->wf_owned.push_back(*ptr);
        return true;
    };
    b.capacity = cap;
    b.bvh_depth = s->d32.view.bvh_depth;
    b.pool.lds_levels = (uint32_t) std::min(b.bvh_depth + 2, RT_WF_LDS_LEVELS_MAX);
    int grid = 0;
    if (rt::wf_extend_grid((int) b.pool.lds_levels, &grid) != hipSuccess) return fail(RT_EHIP, "occupancy query for wf_extend failed");
    b.pool.extend_threads = (uint32_t) grid * 256u;
    if (std::getenv("RT_DEBUG")) fprintf(stderr, "[rtcore] wavefront pool: %u slots, tree depth %d, %u stack levels in LDS, wf_extend grid %d x 256\n", cap, b.bvh_depth, b.pool.lds_levels, grid);
    const size_t spill_levels = (size_t) std::max(0, b.bvh_depth + 2 - (int) b.pool.lds_levels);
    bool ok = dev_alloc((void**) &b.pool.ray_o, (size_t) cap * 16) && dev_alloc((void**) &b.pool.ray_d, (size_t) cap * 16) &&
              dev_alloc((void**) &b.pool.hit, (size_t) cap * 8) && dev_alloc((void**) &b.pool.beta, (size_t) cap * 16) &&
              dev_alloc((void**) &b.pool.rad, (size_t) cap * 16) && dev_alloc((void**) &b.pool.rng, (size_t) cap * 8) &&
              dev_alloc((void**) &b.pool.item, (size_t) cap * 4) && dev_alloc((void**) &b.pool.wg_rays, (size_t) cap / 256 * 8) &&
              dev_alloc((void**) &b.pool.wg_rejected, (size_t) cap / 256 * 8) &&
              dev_alloc((void**) &b.pool.spill, std::max<size_t>(16, spill_levels * b.pool.extend_threads * 4)) &&
              dev_alloc((void**) &b.pool.ctl, sizeof(rt::WfCtl));
    if (ok && hipHostMalloc(&b.host_pinned, sizeof(rt::WfCtl) + 256) != hipSuccess) ok = false;
    if (!ok) {
        for (void* q : s->wf_owned) (void) hipFree(q);
        s->wf_owned.clear();
        return fail(RT_ENOMEM, "allocating the wavefront path pool failed");
    }
    s->wf = b;
    return RT_OK;
}

}  // namespace

// the wave kernel's variant by scene content (rt_device.h class_*): 2 instances, 1 boxes or volumes, 3 an OBJ-loader scene, 5 / 6 / 7 the
// spheres-and-rects twins of 0 / 1 / 2, 0 the rest
static uint32_t scene_kinds(const rt_scene* s)
{
    return s->d32.view.xforms != nullptr ? (s->only_spheres ? 7 : 2) : ((s->has_box || s->has_volume) ? (s->only_spheres ? 6 : 1) : (s->only_triangles ? 3 : (s->only_spheres && !s->quantised ? 5 : 0)));
}

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }
const char* rt_version(void) { return "rtcore-mi355x 0.1 (gfx950)"; }

int rt_scene_create(const rt_scene_desc* desc, rt_scene** scene_out)
{
    if (!scene_out) return fail(RT_EINVAL, "scene_out is NULL");
    *scene_out = nullptr;
    int rc = validate(desc);
    if (rc) return rc;
    Flat f;
    if ((rc = linearize(desc, f))) return rc;
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ENOMEM, "out of host memory");
    hipError_t e = hipGetDevice(&s->device);
    if (e != hipSuccess) { delete s; return fail(RT_EHIP, std::string("hipGetDevice: ") + hipGetErrorString(e) + " (no HIP device: the HIP path is the only path)"); }
    if ((rc = build_device<float>(desc, f, s->d32)) || (rc = build_device<double>(desc, f, s->d64, &s->d32))) { rt_scene_destroy(s); return rc; }
    std::vector<uint32_t>().swap(s->d32.leaf_order);
    s->n_lights = desc->n_lights; s->n_prims = desc->n_prims; s->n_ops = (int32_t) f.op_prim.size();
    for (int i = 0; i < desc->n_prims; i++) { s->has_volume |= desc->prim_kind[i] == RT_PRIM_VOLUME; s->has_box |= desc->prim_kind[i] == RT_PRIM_BOX; s->only_triangles &= desc->prim_kind[i] == RT_PRIM_TRIANGLE; }
    for (int i = 0; i < desc->n_mats; i++) s->only_triangles &= desc->mat_kind[i] == RT_MAT_PHONG;
    for (int i = 0; i < desc->n_tex; i++) s->only_triangles &= desc->tex_kind[i] != RT_TEX_CHECKER;      // (the OBJ loader makes none; a description that has one takes the general variant)
    auto is_rect = [](int k) { return k == RT_PRIM_PLANEXY || k == RT_PRIM_PLANEYZ || k == RT_PRIM_PLANEXZ; };
    for (int i = 0; i < desc->n_prims; i++) s->only_spheres &= desc->prim_kind[i] != RT_PRIM_TRIANGLE;
    for (int i = 0; i < desc->n_lights; i++) s->only_spheres &= is_rect(desc->light_kind[i]);
    for (int i = 0; i < desc->n_mats; i++) s->only_spheres &= desc->mat_kind[i] != RT_MAT_PHONG && (desc->mat_kind[i] != RT_MAT_FOG || s->has_volume);
    for (int i = 0; i < desc->n_tex; i++)       // (an image texture without texels is uploaded as the constant it evaluates to)
        s->only_spheres &= desc->tex_kind[i] == RT_TEX_CONSTANT || desc->tex_kind[i] == RT_TEX_CHECKER || (desc->tex_kind[i] == RT_TEX_IMAGE && desc->tex_data[(size_t) i * RT_TEX_STRIDE] <= 0);
    if (const char* env = std::getenv("RT_LEAN_SPHERES")) s->only_spheres &= std::atoi(env) != 0;      // 0: the general variant (A/B)
    if (hipMalloc((void**) &s->counters, 16 * sizeof(unsigned long long)) != hipSuccess || hipEventCreate(&s->ev0) != hipSuccess ||
        hipEventCreate(&s->ev1) != hipSuccess) { rt_scene_destroy(s); return fail(RT_EHIP, "allocating counters/events failed"); }
    if (const char* env = std::getenv("RT_LDS_LIMIT")) s->lds_limit = (size_t) std::strtoull(env, nullptr, 10);
    if (const char* env = std::getenv("RT_WF_SLOTS")) s->wf_slots = (uint32_t) std::min(64ll << 20, std::max(256ll, std::atoll(env)));
    // measured optima (tools/sweep_wave.py, 4K): small scenes spend as much time shading as traversing and want fuller shading
    // passes (64/64); big meshes are traversal-bound and want fuller node steps (96/128)
    if (desc->n_prims < 4096) { s->trav_keep = 64; s->node_keep = 64; }
    // quantised four-wide nodes halve the dependent node visits of a ray for ~5 % more box tests: mesh workload -6 % time (fp32), -9 %
    // (fp64), 5.2 M triangles -9 % (fp64); 485 spheres: no gain (boxes loosened by the quantisation cost what the fewer visits save)
    s->quantised = s->only_triangles;
    if (const char* env = std::getenv("RT_QNODES")) s->quantised = std::atoi(env) != 0;
    // measured at 4K: 1.3 M triangles -1.4 % time, 5.2 M triangles -2.3 %; 485 spheres (everything L1/L2-resident anyway) +1..4 %
    s->xcd_tiles = (size_t) s->d32.view.n_bnodes * sizeof(rt::BNode) + (size_t) desc->n_prims * sizeof(rt::Prim<float>) > (4u << 20);
    if (const char* env = std::getenv("RT_XCD_TILES")) s->xcd_tiles = std::atoi(env) != 0;
    if (const char* env = std::getenv("RT_TRAV_KEEP")) s->trav_keep = (uint32_t) std::min(256, std::max(1, std::atoi(env)));
    if (const char* env = std::getenv("RT_NODE_KEEP")) s->node_keep = (uint32_t) std::min(256, std::max(1, std::atoi(env)));
    if (const char* env = std::getenv("RT_REGEN_MIN")) s->regen_min = (uint32_t) std::min(64, std::max(1, std::atoi(env)));
    if (const char* env = std::getenv("RT_TILE_ORDER")) s->tile_order = std::atoi(env) != 0;
    if (const char* env = std::getenv("RT_STACK_PAD")) s->stack_pad = (uint32_t) std::min(RT_BVH_STACK_MAX, std::max(0, std::atoi(env)));
    if (const char* env = std::getenv("RT_QUEUE")) s->queue = std::atoi(env) != 0;
    if (const char* env = std::getenv("RT_Q_SLOTS")) s->q_slots = (uint32_t) std::min(RT_Q_SLOTS_MAX, std::max(64, std::atoi(env)));
    if (const char* env = std::getenv("RT_Q_HANDOFF")) s->q_handoff_min = (uint32_t) std::min(64, std::max(1, std::atoi(env)));
    if (const char* env = std::getenv("RT_Q_STARVED")) s->q_shade_starved = (uint32_t) std::min(64, std::max(1, std::atoi(env)));
    *scene_out = s;
    return RT_OK;
}

int rt_scene_destroy(rt_scene* s)
{
    if (!s) return RT_OK;
    release(s->d64); release(s->d32);
    for (void* q : s->wf_owned) (void) hipFree(q);
    if (s->wf.host_pinned) (void) hipHostFree(s->wf.host_pinned);
    for (int k = 0; k < 2; k++) { if (s->q_pools[k]) (void) hipFree(s->q_pools[k]); if (s->q_locks[k]) (void) hipFree(s->q_locks[k]); }
    if (s->counters) (void) hipFree(s->counters);
    if (s->unit_cost) (void) hipFree(s->unit_cost);
    if (s->unit_order) (void) hipFree(s->unit_order);
    if (s->ev0) (void) hipEventDestroy(s->ev0);
    if (s->ev1) (void) hipEventDestroy(s->ev1);
    delete s;
    return RT_OK;
}

int rt_render(rt_scene* s, const rt_render_params* p, void* accum_dev, rt_render_stats* stats)
{
    if (!s || !p || !accum_dev) return fail(RT_EINVAL, "scene, params and accum_dev are required");
    if (p->width < 2 || p->height < 2) return fail(RT_EINVAL, "width and height must be >= 2 (raytracing.cpp:102 divides by width-1)");
    if (p->y_begin < 0 || p->y_end > p->height || p->y_begin > p->y_end) return fail(RT_EINVAL, "row range outside the image");
    if (p->spp_begin < 0 || p->spp_end < p->spp_begin) return fail(RT_EINVAL, "bad sample range");
    if (p->max_depth < 0) return fail(RT_EINVAL, "max_depth < 0");
    if ((long long) p->spp_end - p->spp_begin >= (1 << 23)) return fail(RT_EINVAL, "more than 2^23 - 1 samples per pixel in one call (a tile's items are counted in 32 bits): split the range");
    if ((p->flags & RT_FLAG_WAVEFRONT) && p->max_depth > 65535) return fail(RT_EINVAL, "RT_FLAG_WAVEFRONT keeps the depth in 16 bits: max_depth <= 65535");
    if ((long long) p->width * p->height >= (1LL << 31) / 3) return fail(RT_EINVAL, "image too large");
    if (p->mode != RT_MODE_REFERENCE_FP64 && p->mode != RT_MODE_FAST_FP32 && p->mode != RT_MODE_FP32_REFERENCE_TREE && p->mode != RT_MODE_FAST_FP64) return fail(RT_EINVAL, "unknown mode");
    if ((p->flags & RT_FLAG_COMPAT_FIXED) && ((p->flags & (RT_FLAG_QUEUE | RT_FLAG_WAVEFRONT)) || s->queue || (p->mode == RT_MODE_FAST_FP32 && (p->flags & RT_FLAG_LEGACY_LOOP))))
        return fail(RT_EINVAL, "RT_FLAG_COMPAT_FIXED is implemented by the wave kernel and the reference-order kernels only");
    if (p->mode == RT_MODE_FAST_FP64 && (p->flags & (RT_FLAG_LEGACY_LOOP | RT_FLAG_WAVEFRONT)))
        return fail(RT_EINVAL, "RT_MODE_FAST_FP64 runs the wave kernel only: RT_FLAG_LEGACY_LOOP and RT_FLAG_WAVEFRONT are fp32 A/B variants");
    if (s->n_lights <= 0)
        return fail(RT_EINVAL, "scene.lights is empty: the reference indexes a null light list on the first diffuse hit (group.hpp:115-119)");
    if (s->pending) return fail(RT_EINVAL, "a render of this scene is outstanding (its events and counters are in use): call rt_render_wait first");
    int current = -1;
    HIP_TRY(hipGetDevice(&current));
    if (current != s->device) return fail(RT_EINVAL, "the scene lives on device " + std::to_string(s->device) + " but device " + std::to_string(current) + " is current: call rt_device_set first");
    hipStream_t st = static_cast<hipStream_t>(p->stream);
    rt::RenderArgs a{};
    a.width = p->width; a.height = p->height; a.y_begin = p->y_begin; a.y_end = p->y_end;
    a.wm1 = (double) (p->width - 1); a.hm1 = (double) (p->height - 1);
    a.spp_begin = p->spp_begin; a.spp_end = p->spp_end; a.max_depth = p->max_depth; a.flags = p->flags; a.seed = p->seed;
    a.trav_keep = s->trav_keep; a.node_keep = s->node_keep; a.regen_min = s->regen_min; a.quantised = s->quantised; a.xcd_tiles = s->xcd_tiles; a.kinds = scene_kinds(s);
    if (p->mode == RT_MODE_FAST_FP32 && (p->flags & RT_FLAG_WAVEFRONT)) {
        if (s->has_volume) return fail(RT_EINVAL, "RT_FLAG_WAVEFRONT: the extend kernel has no random stream, which a volume's intersection test draws from");
        int rc = wf_prepare(s);
        if (rc) return rc;
        unsigned long long c[8] = { 0 };
        HIP_TRY(hipEventRecord(s->ev0, st));
        HIP_TRY(rt::wf_render(s->d32.view, a, s->wf, static_cast<float*>(accum_dev), c, st));
        HIP_TRY(hipEventRecord(s->ev1, st));
        HIP_TRY(hipEventSynchronize(s->ev1));
        if (stats) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
            stats->rays = c[0]; stats->rejected_samples = c[1]; stats->kernel_ms = ms; stats->box_tests = c[2]; stats->leaf_tests = c[3];
            stats->wave_node_steps = c[4]; stats->wave_leaf_steps = c[5]; stats->wave_bounces = c[6];
            for (int k = 0; k < 4; k++) stats->phase_ticks[k] = 0;
            stats->zero_direction_rays = 0; stats->untraced_rays = 0;
        }
        return RT_OK;
    }
    // the decoupled kernel (rt_queue.h) serves the fast modes of scenes without boxes, volumes (their free-flight draw needs the path's
    // stream inside the traversal) or instances.  Its pools are allocated -- both or neither -- before anything is queued on the stream.
    const bool fast_mode = p->mode == RT_MODE_FAST_FP64 || p->mode == RT_MODE_FAST_FP32;
    void* q_pools = nullptr; unsigned int* q_locks = nullptr;
    if ((s->queue || (p->flags & RT_FLAG_QUEUE)) && a.kinds == 5) a.kinds = 0;       // the decoupled kernel has no spheres-and-rects twin: its general variant
    const bool decoupled = fast_mode && (s->queue || (p->flags & RT_FLAG_QUEUE)) && !(p->flags & RT_FLAG_LEGACY_LOOP) && (a.kinds == 0 || a.kinds == 3) && p->max_depth < (1 << 21);
    const int qk = p->mode == RT_MODE_FAST_FP64 ? 1 : 0;
    if (decoupled && !s->q_pools[qk]) {
        const size_t bytes = (size_t) s->q_n_pools * 4 * RT_Q_SLOTS_MAX * (qk ? rt::queue_slot_bytes(0.0) : rt::queue_slot_bytes(0.0f));
        void* pools = nullptr; unsigned int* locks = nullptr;
        HIP_TRY(hipMalloc(&pools, bytes));
        const hipError_t e = hipMalloc((void**) &locks, s->q_n_pools * sizeof(unsigned int));
        if (e != hipSuccess) { (void) hipFree(pools); return fail(RT_EHIP, std::string("hipMalloc of the decoupled kernel's pool locks: ") + hipGetErrorString(e)); }
        s->q_pools[qk] = pools; s->q_locks[qk] = locks;
    }
    // cost-ordered launch (wave kernel only): the first render of a frame geometry counts rays per scheduling unit, later ones launch the
    // units dearest first.  Scheduling only: every tile still adds up its own samples, so sums and counts do not change.
    // Measured at 4K, fp64 (tools/drain_ab.py, profiles/r03_*_drain_ab_*): random spheres -4 % time at 16 and 128 spp, Cornell box -6 %, the
    // deep-bounce scene -11 % (16 spp) and -18 % (128 spp): its dear tiles -- 200-bounce paths through glass -- are few, and in image
    // order the last of them ran alone.  Not under the XCD super-tile order (scenes that outgrow an L2): there image order keeps the chip
    // on one compact region of the image and of the tree; sorted by cost the 1.3 M-triangle mesh ran 3-4 % slower at 16, 128 and 512 spp
    // alike, and moving only its cheap super-tiles to the end changed nothing (its fixed ~4 ms per launch is every workgroup's own
    // drain, not the launch's tail).
    if (s->tile_order && !s->xcd_tiles && fast_mode && !decoupled && !(p->flags & (RT_FLAG_LEGACY_LOOP | RT_FLAG_WAVEFRONT)) && p->y_end > p->y_begin) {
        const int tiles_x = (p->width + 15) / 16, tiles_y = (p->y_end - p->y_begin + 15) / 16;
        const uint32_t units = s->xcd_tiles ? (uint32_t) (((tiles_x + 7) / 8) * ((tiles_y + 7) / 8)) : (uint32_t) (tiles_x * tiles_y);
        const int32_t key[5] = { p->width, p->height, p->y_begin, p->y_end, (int32_t) s->xcd_tiles };
        if (std::memcmp(key, s->order_key, sizeof key) != 0 || !s->unit_cost) {
            if (s->unit_cost) (void) hipFree(s->unit_cost);
            if (s->unit_order) (void) hipFree(s->unit_order);
            s->unit_cost = s->unit_order = nullptr; s->order_state = 0; s->n_units = 0;
            std::memset(s->order_key, 0, sizeof s->order_key);
            uint32_t* cost = nullptr; uint32_t* order = nullptr;
            HIP_TRY(hipMalloc((void**) &cost, (size_t) units * sizeof(uint32_t)));
            const hipError_t e = hipMalloc((void**) &order, (size_t) units * sizeof(uint32_t));
            if (e != hipSuccess) { (void) hipFree(cost); return fail(RT_EHIP, std::string("hipMalloc of the launch order: ") + hipGetErrorString(e)); }
            s->unit_cost = cost; s->unit_order = order; s->n_units = units;
            std::memcpy(s->order_key, key, sizeof key);
        }
        if (s->order_state == 2) a.unit_order = s->unit_order;
        else {
            HIP_TRY(hipMemsetAsync(s->unit_cost, 0, (size_t) units * sizeof(uint32_t), st));
            a.unit_cost = s->unit_cost;
            s->order_state = 1;
        }
    }
    HIP_TRY(hipMemsetAsync(s->counters, 0, 16 * sizeof(unsigned long long), st));
    HIP_TRY(hipEventRecord(s->ev0, st));
    if (decoupled) {
        HIP_TRY(hipMemsetAsync(s->q_locks[qk], 0, s->q_n_pools * sizeof(unsigned int), st));
        q_pools = s->q_pools[qk]; q_locks = s->q_locks[qk];
        a.q_slots = s->q_slots; a.q_pools = s->q_n_pools; a.q_handoff_min = s->q_handoff_min; a.q_shade_starved = s->q_shade_starved;
    }
    if (q_pools && a.kinds == 0) a.quantised = 0;              // the decoupled kernel's generic variant walks the binary nodes only
    // the kernels launch_render gives the wide nodes to (rt_kernels.inl): wave and decoupled kernel of the variants 0 and 3
    const bool wide = a.quantised != 0 && (a.kinds == 0 || a.kinds == 3) && !(p->flags & RT_FLAG_LEGACY_LOOP);
    if (p->mode == RT_MODE_FAST_FP64) {
        a.bvh_stack_levels = (uint32_t) (wide ? s->d64.view.wide_stack : s->d64.view.bvh_depth) + 2 + s->stack_pad;
        HIP_TRY(rt::launch_render(s->d64.view, a, static_cast<double*>(accum_dev), s->counters, st, q_pools, q_locks));
    }
    else if (p->mode == RT_MODE_REFERENCE_FP64) {
        if ((size_t) s->d64.ops_bytes + s->d64.prims_bytes <= s->lds_limit) { a.lds_ops_bytes = s->d64.ops_bytes; a.lds_prims_bytes = s->d64.prims_bytes; }
        HIP_TRY(rt::launch_render(s->d64.view, a, static_cast<double*>(accum_dev), s->counters, st));
    }
    else {
        if (p->mode == RT_MODE_FAST_FP32) a.bvh_stack_levels = (uint32_t) (wide ? s->d32.view.wide_stack : s->d32.view.bvh_depth) + 2 + s->stack_pad;
        else if ((size_t) s->d32.ops_bytes + s->d32.prims_bytes <= s->lds_limit) { a.lds_ops_bytes = s->d32.ops_bytes; a.lds_prims_bytes = s->d32.prims_bytes; }
        HIP_TRY(rt::launch_render(s->d32.view, a, static_cast<float*>(accum_dev), s->counters, st, q_pools, q_locks));
    }
    HIP_TRY(hipEventRecord(s->ev1, st));
    s->pending = true;
    if (stats) return rt_render_wait(s, stats);
    return RT_OK;
}

int rt_render_kernel_name(rt_scene* s, int mode, int flags, char* out, size_t n)
{
    if (!s || !out || n == 0) return fail(RT_EINVAL, "scene and an output buffer are required");
    const bool fp64 = mode == RT_MODE_FAST_FP64 || mode == RT_MODE_REFERENCE_FP64;
    const bool count = (flags & RT_FLAG_COUNT_TESTS) != 0;
    std::string name;
    if ((mode == RT_MODE_FAST_FP64 || mode == RT_MODE_FAST_FP32) && !(flags & (RT_FLAG_LEGACY_LOOP | RT_FLAG_WAVEFRONT))) {
        uint32_t kinds = scene_kinds(s);
        const bool queue = (s->queue || (flags & RT_FLAG_QUEUE)) && (kinds == 0 || kinds == 3 || kinds == 5);
        if (queue && kinds == 5) kinds = 0;
        const bool wide = s->quantised != 0 && (kinds == 0 || kinds == 3) && !(queue && kinds == 0);
        const bool sorted = wide && !queue && (fp64 ? s->d64.view.sorted_visit : s->d32.view.sorted_visit) != 0;     // rt_wave.h render_wave_sorted_kernel
        if (sorted) name = std::string("rt::render_wave_sorted_kernel<") + (fp64 ? "double" : "float") + ", " + (count ? "true" : "false") + ", " + std::to_string(kinds) + ">";
        else name = std::string(queue ? "rt::render_queue_kernel<" : "rt::render_wave_kernel<") + (fp64 ? "double" : "float") + ", " + (count ? "true" : "false") + ", " +
               (wide ? "true" : "false") + ", " + std::to_string(kinds) + ">";
    }
    else if (mode == RT_MODE_FAST_FP32 && (flags & RT_FLAG_WAVEFRONT)) name = "rt::wf_extend_kernel";
    else name = std::string("rt::render_kernel<") + (fp64 ? "double" : "float") + ", " + (mode == RT_MODE_FAST_FP32 ? "2" : ((size_t) (fp64 ? s->d64.ops_bytes + s->d64.prims_bytes : s->d32.ops_bytes + s->d32.prims_bytes) <= s->lds_limit ? "1" : "0")) + ", " + (count ? "true" : "false") + ">";
    std::snprintf(out, n, "%s", name.c_str());
    return RT_OK;
}

int rt_render_wait(rt_scene* s, rt_render_stats* stats)
{
    if (!s) return fail(RT_EINVAL, "scene is required");
    if (!s->pending) return fail(RT_EINVAL, "rt_render_wait: no render of this scene is outstanding");
    // events and counters belong to the scene's device; the wait itself needs no particular current device, the read-back does
    int current = -1;
    HIP_TRY(hipGetDevice(&current));
    if (current != s->device) HIP_TRY(hipSetDevice(s->device));
    int rc = RT_OK;
    auto chk = [&](hipError_t err, const char* what) { if (err != hipSuccess && rc == RT_OK) rc = fail(RT_EHIP, std::string(what) + ": " + hipGetErrorString(err)); };
    chk(hipEventSynchronize(s->ev1), "hipEventSynchronize");
    s->pending = false;
    if (rc == RT_OK && s->order_state == 1 && s->unit_cost && s->n_units) {       // the render just waited for counted rays per unit: sort once
        std::vector<uint32_t> cost(s->n_units), order(s->n_units);
        chk(hipMemcpy(cost.data(), s->unit_cost, cost.size() * sizeof(uint32_t), hipMemcpyDeviceToHost), "launch-order costs");
        for (uint32_t i = 0; i < s->n_units; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
        chk(hipMemcpy(s->unit_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "launch order");
        s->order_state = rc == RT_OK ? 2 : 0;
    }
    if (rc == RT_OK && stats) {
        float ms = 0;
        chk(hipEventElapsedTime(&ms, s->ev0, s->ev1), "hipEventElapsedTime");
        unsigned long long c[16] = { 0 };
        chk(hipMemcpy(c, s->counters, sizeof c, hipMemcpyDeviceToHost), "counter read-back");
        for (int k = 0; k < 4; k++) stats->phase_ticks[k] = c[8 + k];
        if (getenv("RT_DEBUG") && c[4]) fprintf(stderr, "[rtcore] node-step census per wave step: inner %.2f leaf %.2f hit-known %.2f no-path %.2f\n", (double) c[2] / 2 / c[4], (double) c[12] / c[4], (double) c[13] / c[4], (double) c[14] / c[4]);
        stats->rays = c[0]; stats->rejected_samples = c[1]; stats->kernel_ms = ms; stats->box_tests = c[2]; stats->leaf_tests = c[3]; stats->wave_node_steps = c[4]; stats->wave_leaf_steps = c[5]; stats->wave_bounces = c[6];
        stats->zero_direction_rays = c[7]; stats->untraced_rays = c[15];
    }
    if (current != s->device) (void) hipSetDevice(current);
    return rc;
}

int rt_device_count(int* count_out)
{
    if (!count_out) return fail(RT_EINVAL, "count_out is NULL");
    HIP_TRY(hipGetDeviceCount(count_out));
    return RT_OK;
}

int rt_device_set(int device)
{
    HIP_TRY(hipSetDevice(device));
    return RT_OK;
}

// ---- rt_frame_reduce: per-destination-device context (staging buffers, streams, events), created on first use and kept until
// rt_frame_reduce_release: a frame is reduced once per rendered frame, and two hipMallocs of 199 MB per call were most of its cost.
namespace {
struct ReduceCtx {
    std::vector<void*> stage;               // one staging buffer per source in flight (<= kReduceLanes), each `bytes` large
    size_t bytes = 0;
    std::vector<hipStream_t> copy_stream;   // one per staging buffer: the peer copies of different sources run together, each over its
                                            // source's own xGMI link (the links are point to point)
    std::vector<hipEvent_t> copied, added;
    hipStream_t add_stream = nullptr;
};
constexpr int kReduceLanes = 8;             // sources copied at once: 7 peers of an 8-GPU node + one for a source on the destination itself
std::mutex g_reduce_mutex;
ReduceCtx g_reduce[kMaxDevices];
bool g_peer_enabled[kMaxDevices][kMaxDevices] = { { false } };

void reduce_release(ReduceCtx& c)
{
    for (void* p : c.stage) if (p) (void) hipFree(p);
    for (hipStream_t st : c.copy_stream) if (st) (void) hipStreamDestroy(st);
    for (hipEvent_t e : c.copied) if (e) (void) hipEventDestroy(e);
    for (hipEvent_t e : c.added) if (e) (void) hipEventDestroy(e);
    if (c.add_stream) (void) hipStreamDestroy(c.add_stream);
    c = ReduceCtx();
}

// with dst_device current: `lanes` staging buffers of `bytes` each, with their streams and events
int reduce_prepare(ReduceCtx& c, int lanes, size_t bytes)
{
    if (c.bytes < bytes) {                  // grow: drop the old buffers first (a frame geometry change, rare)
        for (void*& p : c.stage) { if (p) (void) hipFree(p); p = nullptr; }
        c.stage.clear();
        c.bytes = bytes;
    }
    if (!c.add_stream) HIP_TRY(hipStreamCreateWithFlags(&c.add_stream, hipStreamNonBlocking));
    while ((int) c.stage.size() < lanes) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, c.bytes));
        c.stage.push_back(p);
    }
    while ((int) c.copy_stream.size() < lanes) {
        hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        c.copy_stream.push_back(st);
        HIP_TRY(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
        c.copied.push_back(e0);
        HIP_TRY(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
        c.added.push_back(e1);
    }
    return RT_OK;
}

// with dst current: let dst's DMA engines and kernels read src's memory directly (xGMI), once per ordered pair.  Where the platform
// has no peer path hipMemcpyPeerAsync still works (it stages through the host), so a refusal is not an error.
void enable_peer(int dst, int src)
{
    if (dst == src || dst < 0 || src < 0 || dst >= kMaxDevices || src >= kMaxDevices || g_peer_enabled[dst][src]) return;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dst, src) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(src, 0);
        if (e != hipSuccess) (void) hipGetLastError();       // hipErrorPeerAccessAlreadyEnabled (someone else enabled it) is fine: clear it
    }
    g_peer_enabled[dst][src] = true;                          // asked once, whatever the answer
}
}  // namespace

int rt_frame_reduce(void* dst_dev, int dst_device, void* const* src_dev, const int* src_device, int n_src, int accum_is_fp64, size_t n_values)
{
    if (!dst_dev || n_src < 0 || (n_src && (!src_dev || !src_device))) return fail(RT_EINVAL, "dst_dev and n_src source buffers with their devices are required");
    if (n_src == 0 || n_values == 0) return RT_OK;
    if (n_values > (size_t) 0x7fffffff * 256) return fail(RT_EINVAL, "rt_frame_reduce: buffer too large");
    if (dst_device < 0 || dst_device >= kMaxDevices) return fail(RT_EUNSUPPORTED, "device index beyond the library's per-device table");
    for (int i = 0; i < n_src; i++) {
        if (!src_dev[i]) return fail(RT_EINVAL, "a source buffer is NULL");
        if (src_device[i] < 0 || src_device[i] >= kMaxDevices) return fail(RT_EUNSUPPORTED, "device index beyond the library's per-device table");
    }
    const size_t bytes = n_values * (accum_is_fp64 ? 8 : 4);
    int current = -1;
    HIP_TRY(hipGetDevice(&current));
    std::lock_guard<std::mutex> lock(g_reduce_mutex);
    int rc = RT_OK;
    auto chk = [&](hipError_t err, const char* what) { if (err != hipSuccess && rc == RT_OK) rc = fail(RT_EHIP, std::string(what) + ": " + hipGetErrorString(err)); };
    // whatever wrote the buffers (renders on the streams of their devices) must be done before they are read or added to
    for (int i = 0; i < n_src && rc == RT_OK; i++) {
        chk(hipSetDevice(src_device[i]), "hipSetDevice");
        chk(hipDeviceSynchronize(), "hipDeviceSynchronize (source)");
    }
    chk(hipSetDevice(dst_device), "hipSetDevice");
    chk(hipDeviceSynchronize(), "hipDeviceSynchronize (destination)");
    ReduceCtx& c = g_reduce[dst_device];
    const int lanes = std::min(n_src, kReduceLanes);
    if (rc == RT_OK) rc = reduce_prepare(c, lanes, bytes);
    if (rc == RT_OK) for (int i = 0; i < n_src; i++) enable_peer(dst_device, src_device[i]);
    const unsigned blocks = (unsigned) ((n_values + 255) / 256);
    // every source's copy is queued on its lane's stream at once (lanes run together); the additions follow on one stream in source
    // order, each waiting for its copy; a lane's buffer is reused by source i + kReduceLanes once source i has been added
    for (int i = 0; i < n_src && rc == RT_OK; i++) {
        const int b = i % lanes;
        if (i >= lanes) chk(hipStreamWaitEvent(c.copy_stream[b], c.added[b], 0), "hipStreamWaitEvent");
        chk(hipMemcpyPeerAsync(c.stage[b], dst_device, src_dev[i], src_device[i], bytes, c.copy_stream[b]), "hipMemcpyPeerAsync");
        chk(hipEventRecord(c.copied[b], c.copy_stream[b]), "hipEventRecord");
        chk(hipStreamWaitEvent(c.add_stream, c.copied[b], 0), "hipStreamWaitEvent");
        if (rc != RT_OK) break;
        if (accum_is_fp64) hipLaunchKernelGGL(add_kernel<double>, dim3(blocks), dim3(256), 0, c.add_stream, static_cast<double*>(dst_dev), static_cast<const double*>(c.stage[b]), n_values);
        else hipLaunchKernelGGL(add_kernel<float>, dim3(blocks), dim3(256), 0, c.add_stream, static_cast<float*>(dst_dev), static_cast<const float*>(c.stage[b]), n_values);
        chk(hipGetLastError(), "add kernel launch");
        chk(hipEventRecord(c.added[b], c.add_stream), "hipEventRecord");
    }
    // also on failure: nothing of this call may still be in flight when the caller frees its buffers
    if (c.add_stream) chk(hipStreamSynchronize(c.add_stream), "hipStreamSynchronize");
    for (hipStream_t st : c.copy_stream) chk(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (current != dst_device) (void) hipSetDevice(current);
    return rc;
}

int rt_frame_reduce_release(void)
{
    int current = -1;
    HIP_TRY(hipGetDevice(&current));
    std::lock_guard<std::mutex> lock(g_reduce_mutex);
    for (int d = 0; d < kMaxDevices; d++) {
        if (g_reduce[d].stage.empty() && !g_reduce[d].add_stream) continue;
        if (hipSetDevice(d) == hipSuccess) reduce_release(g_reduce[d]);
    }
    (void) hipSetDevice(current);
    return RT_OK;
}

int rt_resolve(const void* accum_dev, int accum_is_fp64, size_t n_pixels, int spp, uint8_t* rgb_dev, void* stream)
{
    if (!accum_dev || !rgb_dev || spp <= 0) return fail(RT_EINVAL, "accum_dev, rgb_dev and spp > 0 are required");
    size_t n = n_pixels * 3;
    if (n == 0) return RT_OK;
    if (n > (size_t) 0x7fffffff * 256) return fail(RT_EINVAL, "image too large");
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), accum_dev,
                       accum_is_fp64, n, 1 / (double) spp, rgb_dev);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// the first-hit entry points stage their rays on the current device and launch there: the scene must live on it (as rt_render requires)
static int require_scene_device(const rt_scene* s)
{
    int current = -1;
    HIP_TRY(hipGetDevice(&current));
    if (current != s->device) return fail(RT_EINVAL, "the scene lives on device " + std::to_string(s->device) + " but device " + std::to_string(current) + " is current: call rt_device_set first");
    return RT_OK;
}

int rt_first_hit(rt_scene* s, const double* rays, size_t n, int32_t* prim, double* t, int32_t* front, uint32_t* n_box, uint32_t* n_leaf,
                 uint64_t* visit_hash)
{
    if (!s || (!rays && n)) return fail(RT_EINVAL, "scene and rays are required");
    if (n == 0) return RT_OK;
    if (n > (size_t) 1 << 28) return fail(RT_EINVAL, "too many rays in one call");
    if (int rc_dev = require_scene_device(s)) return rc_dev;
    double* d_rays = nullptr; void* d_out = nullptr;
    const size_t per = 4 + 8 + 4 + 4 + 4 + 8;
    HIP_TRY(hipMalloc((void**) &d_rays, n * 7 * sizeof(double)));
    hipError_t e = hipMalloc(&d_out, n * per);
    if (e != hipSuccess) { (void) hipFree(d_rays); return fail(RT_EHIP, "hipMalloc failed"); }
    char* b = static_cast<char*>(d_out);
    double* d_t = reinterpret_cast<double*>(b);
    unsigned long long* d_hash = reinterpret_cast<unsigned long long*>(b + n * 8);
    int32_t* d_prim = reinterpret_cast<int32_t*>(b + n * 16);
    int32_t* d_front = reinterpret_cast<int32_t*>(b + n * 20);
    uint32_t* d_box = reinterpret_cast<uint32_t*>(b + n * 24);
    uint32_t* d_leaf = reinterpret_cast<uint32_t*>(b + n * 28);
    int rc = RT_OK;
    auto chk = [&](hipError_t err, const char* what) { if (err != hipSuccess && rc == RT_OK) rc = fail(RT_EHIP, std::string(what) + ": " + hipGetErrorString(err)); };
    chk(hipMemcpy(d_rays, rays, n * 7 * sizeof(double), hipMemcpyHostToDevice), "upload rays");
    if (rc == RT_OK) chk(rt::launch_first_hit(s->d64.view, d_rays, (int) n, d_prim, d_t, d_front, d_box, d_leaf, d_hash, nullptr), "first_hit launch");
    if (rc == RT_OK) chk(hipDeviceSynchronize(), "first_hit kernel");
    if (rc == RT_OK && prim) chk(hipMemcpy(prim, d_prim, n * 4, hipMemcpyDeviceToHost), "download");
    if (rc == RT_OK && t) chk(hipMemcpy(t, d_t, n * 8, hipMemcpyDeviceToHost), "download");
    if (rc == RT_OK && front) chk(hipMemcpy(front, d_front, n * 4, hipMemcpyDeviceToHost), "download");
    if (rc == RT_OK && n_box) chk(hipMemcpy(n_box, d_box, n * 4, hipMemcpyDeviceToHost), "download");
    if (rc == RT_OK && n_leaf) chk(hipMemcpy(n_leaf, d_leaf, n * 4, hipMemcpyDeviceToHost), "download");
    if (rc == RT_OK && visit_hash) chk(hipMemcpy(visit_hash, d_hash, n * 8, hipMemcpyDeviceToHost), "download");
    (void) hipFree(d_rays); (void) hipFree(d_out);
    return rc;
}

int rt_first_hit_noalloc(rt_scene* s, const double* rays, size_t n, int32_t* prim, double* t, int32_t* front, uint32_t* n_box, uint32_t* n_leaf,
                         uint64_t* visit_hash)
{
    if (!s || (!rays && n)) return fail(RT_EINVAL, "scene and rays are required");
    if (n > (size_t) RT_NOALLOC_MAX_RAYS)
        return fail(RT_EINVAL, "rt_first_hit_noalloc: at most RT_NOALLOC_MAX_RAYS (" + std::to_string(RT_NOALLOC_MAX_RAYS) + ") rays per call, got " + std::to_string(n) +
                                   "; split the batch or use rt_first_hit");
    if (n == 0) return RT_OK;
    int device = -1;
    HIP_TRY(hipGetDevice(&device));
    if (device != s->device) return fail(RT_EINVAL, "the scene lives on another device than the current one");
    void* base = nullptr;
    int rc = scratch_address(device, &base);
    if (rc) return rc;
    // every byte a call reads it has written first (rays by the upload, outputs by the kernel for all idx < n): nothing to clear
    char* b = static_cast<char*>(base);
    double* d_rays = reinterpret_cast<double*>(b);
    char* o = b + n * 7 * sizeof(double);
    double* d_t = reinterpret_cast<double*>(o);
    unsigned long long* d_hash = reinterpret_cast<unsigned long long*>(o + n * 8);
    int32_t* d_prim = reinterpret_cast<int32_t*>(o + n * 16);
    int32_t* d_front = reinterpret_cast<int32_t*>(o + n * 20);
    uint32_t* d_box = reinterpret_cast<uint32_t*>(o + n * 24);
    uint32_t* d_leaf = reinterpret_cast<uint32_t*>(o + n * 28);
    HIP_TRY(hipMemcpy(d_rays, rays, n * 7 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(rt::launch_first_hit(s->d64.view, d_rays, (int) n, d_prim, d_t, d_front, d_box, d_leaf, d_hash, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    if (prim) HIP_TRY(hipMemcpy(prim, d_prim, n * 4, hipMemcpyDeviceToHost));
    if (t) HIP_TRY(hipMemcpy(t, d_t, n * 8, hipMemcpyDeviceToHost));
    if (front) HIP_TRY(hipMemcpy(front, d_front, n * 4, hipMemcpyDeviceToHost));
    if (n_box) HIP_TRY(hipMemcpy(n_box, d_box, n * 4, hipMemcpyDeviceToHost));
    if (n_leaf) HIP_TRY(hipMemcpy(n_leaf, d_leaf, n * 4, hipMemcpyDeviceToHost));
    if (visit_hash) HIP_TRY(hipMemcpy(visit_hash, d_hash, n * 8, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_first_hit_fast(rt_scene* s, const double* rays, size_t n, int32_t* prim, float* t)
{
    if (!s || (!rays && n)) return fail(RT_EINVAL, "scene and rays are required");
    if (n == 0) return RT_OK;
    if (n > (size_t) 1 << 28) return fail(RT_EINVAL, "too many rays in one call");
    if (int rc_dev = require_scene_device(s)) return rc_dev;
    double* d_rays = nullptr; int32_t* d_prim = nullptr; float* d_t = nullptr;
    int rc = RT_OK;
    auto chk = [&](hipError_t err, const char* what) { if (err != hipSuccess && rc == RT_OK) rc = fail(RT_EHIP, std::string(what) + ": " + hipGetErrorString(err)); };
    chk(hipMalloc((void**) &d_rays, n * 7 * sizeof(double)), "hipMalloc");
    chk(hipMalloc((void**) &d_prim, n * 4), "hipMalloc");
    chk(hipMalloc((void**) &d_t, n * 4), "hipMalloc");
    if (rc == RT_OK) chk(hipMemcpy(d_rays, rays, n * 7 * sizeof(double), hipMemcpyHostToDevice), "upload rays");
    if (rc == RT_OK) chk(rt::launch_first_hit_fast(s->d32.view, d_rays, (int) n, d_prim, d_t, nullptr), "first_hit_fast launch");
    if (rc == RT_OK) chk(hipDeviceSynchronize(), "first_hit_fast kernel");
    if (rc == RT_OK && prim) chk(hipMemcpy(prim, d_prim, n * 4, hipMemcpyDeviceToHost), "download");
    if (rc == RT_OK && t) chk(hipMemcpy(t, d_t, n * 4, hipMemcpyDeviceToHost), "download");
    (void) hipFree(d_rays); (void) hipFree(d_prim); (void) hipFree(d_t);
    return rc;
}

int rt_first_hit_tree(rt_scene* s, const double* rays, size_t n, int fp64, int32_t* prim, double* t, int32_t* front)
{
    if (!s || (!rays && n)) return fail(RT_EINVAL, "scene and rays are required");
    if (n == 0) return RT_OK;
    if (n > (size_t) 1 << 28) return fail(RT_EINVAL, "too many rays in one call");
    if (int rc_dev = require_scene_device(s)) return rc_dev;
    double* d_rays = nullptr; int32_t* d_prim = nullptr; int32_t* d_front = nullptr; void* d_t = nullptr;
    int rc = RT_OK;
    auto chk = [&](hipError_t err, const char* what) { if (err != hipSuccess && rc == RT_OK) rc = fail(RT_EHIP, std::string(what) + ": " + hipGetErrorString(err)); };
    chk(hipMalloc((void**) &d_rays, n * 7 * sizeof(double)), "hipMalloc");
    chk(hipMalloc((void**) &d_prim, n * 4), "hipMalloc");
    chk(hipMalloc((void**) &d_front, n * 4), "hipMalloc");
    chk(hipMalloc(&d_t, n * 8), "hipMalloc");
    if (rc == RT_OK) chk(hipMemcpy(d_rays, rays, n * 7 * sizeof(double), hipMemcpyHostToDevice), "upload rays");
    if (rc == RT_OK) {
        if (fp64) chk(rt::launch_first_hit_tree(s->d64.view, d_rays, (int) n, (int) s->quantised, d_prim, static_cast<double*>(d_t), d_front, nullptr), "first_hit_tree launch");
        else chk(rt::launch_first_hit_tree(s->d32.view, d_rays, (int) n, (int) s->quantised, d_prim, static_cast<float*>(d_t), d_front, nullptr), "first_hit_tree launch");
    }
    if (rc == RT_OK) chk(hipDeviceSynchronize(), "first_hit_tree kernel");
    if (rc == RT_OK && prim) chk(hipMemcpy(prim, d_prim, n * 4, hipMemcpyDeviceToHost), "download");
    if (rc == RT_OK && front) chk(hipMemcpy(front, d_front, n * 4, hipMemcpyDeviceToHost), "download");
    if (rc == RT_OK && t) {
        if (fp64) chk(hipMemcpy(t, d_t, n * 8, hipMemcpyDeviceToHost), "download");
        else {
            std::vector<float> tf(n);
            chk(hipMemcpy(tf.data(), d_t, n * 4, hipMemcpyDeviceToHost), "download");
            for (size_t i = 0; i < n; i++) t[i] = tf[i];
        }
    }
    (void) hipFree(d_rays); (void) hipFree(d_prim); (void) hipFree(d_front); (void) hipFree(d_t);
    return rc;
}

int rt_buffer_alloc(size_t bytes, void** dev_out)
{
    if (!dev_out) return fail(RT_EINVAL, "dev_out is NULL");
    HIP_TRY(hipMalloc(dev_out, bytes ? bytes : 16));
    return RT_OK;
}
int rt_buffer_free(void* dev) { if (dev) HIP_TRY(hipFree(dev)); return RT_OK; }
int rt_buffer_download(void* host_dst, const void* dev_src, size_t bytes) { HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost)); return RT_OK; }
int rt_buffer_upload(void* dev_dst, const void* host_src, size_t bytes) { HIP_TRY(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice)); return RT_OK; }

// ---- binary scene cache: header, then the arrays in declaration order, each padded to 8 bytes
namespace {
struct CacheHeader {
    char magic[8];                      // "RTSCENE2"
    int32_t n_nodes, n_roots, n_prims, n_mats, n_tex, n_lights, n_xforms, pad;
    uint64_t n_pixel_bytes;
    double camera[24];
};
struct CacheArray { const void* src; size_t bytes; };
size_t pad8(size_t n) { return (n + 7) & ~(size_t) 7; }
enum { kCacheArrays = 20 };
void cache_arrays(const rt_scene_desc* d, CacheArray a[kCacheArrays])
{
    const size_t nn = (size_t) d->n_nodes, np = (size_t) d->n_prims, nm = (size_t) d->n_mats, nt = (size_t) d->n_tex, nl = (size_t) d->n_lights;
    const size_t nx = (size_t) d->n_xforms;
    const CacheArray v[kCacheArrays] = {
        { d->node_bbox, nn * 6 * 8 }, { d->node_child, nn * 2 * 4 }, { d->roots, (size_t) d->n_roots * 4 },
        { d->prim_kind, np * 4 }, { d->prim_flags, np * 4 }, { d->prim_mat, np * 4 }, { d->prim_data, np * RT_PRIM_STRIDE * 8 },
        { d->mat_kind, nm * 4 }, { d->mat_tex, nm * 4 }, { d->mat_data, nm * RT_MAT_STRIDE * 8 },
        { d->tex_kind, nt * 4 }, { d->tex_data, nt * RT_TEX_STRIDE * 8 }, { d->tex_pixels, d->n_pixel_bytes },
        { d->light_kind, nl * 4 }, { d->light_flags, nl * 4 }, { d->light_mat, nl * 4 }, { d->light_data, nl * RT_PRIM_STRIDE * 8 },
        { d->xform_data, nx * RT_XFORM_STRIDE * 8 }, { d->node_xform, nx ? nn * 4 : 0 }, { d->prim_xform, nx ? np * 4 : 0 } };
    for (int i = 0; i < kCacheArrays; i++) a[i] = v[i];
}
struct CheckpointHeader { char magic[8]; int32_t width, height, spp_done, is_fp64; uint64_t seed; };   // "RTCKPT01"
}  // namespace

int rt_scene_desc_save(const rt_scene_desc* d, const char* path)
{
    if (!path) return fail(RT_EINVAL, "path is NULL");
    int rc = validate(d);
    if (rc) return rc;
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(RT_EINVAL, std::string("cannot write ") + path);
    CacheHeader h{};
    std::memcpy(h.magic, "RTSCENE2", 8);
    h.n_nodes = d->n_nodes; h.n_roots = d->n_roots; h.n_prims = d->n_prims; h.n_mats = d->n_mats; h.n_tex = d->n_tex; h.n_lights = d->n_lights; h.n_xforms = d->n_xforms;
    h.n_pixel_bytes = d->n_pixel_bytes;
    std::memcpy(h.camera, d->camera, sizeof h.camera);
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    CacheArray a[kCacheArrays];
    cache_arrays(d, a);
    const char zeros[8] = { 0 };
    for (int i = 0; i < kCacheArrays && ok; i++) {
        if (a[i].bytes && a[i].src) ok = std::fwrite(a[i].src, 1, a[i].bytes, f) == a[i].bytes;
        else if (a[i].bytes && a[i].src == nullptr && i == 15) {        // light_mat is optional (rt_scene_create reads a NULL one as zeros): so it is here
            const std::vector<char> z(a[i].bytes, 0);
            ok = std::fwrite(z.data(), 1, z.size(), f) == z.size();
        }
        else if (a[i].bytes) { std::fclose(f); return fail(RT_EINVAL, "a required array of the scene description is NULL"); }
        const size_t padding = pad8(a[i].bytes) - a[i].bytes;
        if (ok && padding) ok = std::fwrite(zeros, 1, padding, f) == padding;
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : fail(RT_EINVAL, std::string("short write to ") + path);
}

int rt_scene_desc_load(const char* path, rt_scene_desc** out)
{
    if (!path || !out) return fail(RT_EINVAL, "path and desc_out are required");
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(RT_EINVAL, std::string("cannot read ") + path);
    CacheHeader h{};
    if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, "RTSCENE2", 8) != 0 || h.n_nodes < 0 || h.n_roots <= 0 || h.n_prims <= 0 ||
        h.n_mats <= 0 || h.n_tex < 0 || h.n_lights < 0 || h.n_xforms < 0 || h.n_xforms > 65534) { std::fclose(f); return fail(RT_EINVAL, std::string(path) + " is not a scene cache"); }
    rt_scene_desc tmp{};
    tmp.n_nodes = h.n_nodes; tmp.n_roots = h.n_roots; tmp.n_prims = h.n_prims; tmp.n_mats = h.n_mats; tmp.n_tex = h.n_tex; tmp.n_lights = h.n_lights; tmp.n_xforms = h.n_xforms;
    tmp.n_pixel_bytes = (size_t) h.n_pixel_bytes;
    CacheArray a[kCacheArrays];
    cache_arrays(&tmp, a);
    size_t total = pad8(sizeof(rt_scene_desc));
    for (int i = 0; i < kCacheArrays; i++) total += pad8(a[i].bytes);
    // the file must hold exactly what the header promises (a truncated or foreign file must not drive a huge allocation)
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, (long) sizeof h, SEEK_SET);
    if (size < 0 || (size_t) size != sizeof h + total - pad8(sizeof(rt_scene_desc))) { std::fclose(f); return fail(RT_EINVAL, std::string(path) + ": size does not match its header"); }
    char* block = static_cast<char*>(std::malloc(total));
    if (!block) { std::fclose(f); return fail(RT_ENOMEM, "out of host memory"); }
    rt_scene_desc* d = reinterpret_cast<rt_scene_desc*>(block);
    *d = tmp;
    std::memcpy(d->camera, h.camera, sizeof h.camera);
    char* p = block + pad8(sizeof(rt_scene_desc));
    const void* ptrs[kCacheArrays];
    bool ok = true;
    for (int i = 0; i < kCacheArrays && ok; i++) {
        ptrs[i] = a[i].bytes ? p : nullptr;
        const size_t n = pad8(a[i].bytes);
        if (n) ok = std::fread(p, 1, n, f) == n;
        p += n;
    }
    std::fclose(f);
    if (!ok) { std::free(block); return fail(RT_EINVAL, std::string("short read from ") + path); }
    d->node_bbox = (const double*) ptrs[0]; d->node_child = (const int32_t*) ptrs[1]; d->roots = (const int32_t*) ptrs[2];
    d->prim_kind = (const int32_t*) ptrs[3]; d->prim_flags = (const int32_t*) ptrs[4]; d->prim_mat = (const int32_t*) ptrs[5]; d->prim_data = (const double*) ptrs[6];
    d->mat_kind = (const int32_t*) ptrs[7]; d->mat_tex = (const int32_t*) ptrs[8]; d->mat_data = (const double*) ptrs[9];
    d->tex_kind = (const int32_t*) ptrs[10]; d->tex_data = (const double*) ptrs[11]; d->tex_pixels = (const uint8_t*) ptrs[12];
    d->light_kind = (const int32_t*) ptrs[13]; d->light_flags = (const int32_t*) ptrs[14]; d->light_mat = (const int32_t*) ptrs[15]; d->light_data = (const double*) ptrs[16];
    d->xform_data = (const double*) ptrs[17]; d->node_xform = (const int32_t*) ptrs[18]; d->prim_xform = (const int32_t*) ptrs[19];
    const int rc = validate(d);                 // a cache is input like any other: index ranges are checked before anything walks it
    if (rc) { std::free(block); return rc; }
    *out = d;
    return RT_OK;
}

int rt_scene_desc_free(rt_scene_desc* d) { std::free(d); return RT_OK; }

int rt_checkpoint_save(const char* path, const void* accum_dev, int is_fp64, int32_t width, int32_t height, int32_t spp_done, uint64_t seed)
{
    if (!path || !accum_dev || width <= 0 || height <= 0 || spp_done < 0) return fail(RT_EINVAL, "path, accum_dev, a positive size and spp_done >= 0 are required");
    const size_t bytes = (size_t) width * height * 3 * (is_fp64 ? 8 : 4);
    std::vector<char> host(bytes);
    HIP_TRY(hipMemcpy(host.data(), accum_dev, bytes, hipMemcpyDeviceToHost));
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(RT_EINVAL, std::string("cannot write ") + path);
    CheckpointHeader h{};
    std::memcpy(h.magic, "RTCKPT01", 8);
    h.width = width; h.height = height; h.spp_done = spp_done; h.is_fp64 = is_fp64 ? 1 : 0; h.seed = seed;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && std::fwrite(host.data(), 1, bytes, f) == bytes;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : fail(RT_EINVAL, std::string("short write to ") + path);
}

int rt_checkpoint_load(const char* path, void* accum_dev, int is_fp64, int32_t width, int32_t height, int32_t* spp_done, uint64_t* seed)
{
    if (!path || !accum_dev || !spp_done) return fail(RT_EINVAL, "path, accum_dev and spp_done are required");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(RT_EINVAL, std::string("cannot read ") + path);
    CheckpointHeader h{};
    const bool head = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "RTCKPT01", 8) == 0;
    if (!head || h.width != width || h.height != height || h.is_fp64 != (is_fp64 ? 1 : 0) || h.spp_done < 0) {
        std::fclose(f);
        return fail(RT_EINVAL, std::string(path) + (head ? ": checkpoint of a different frame geometry or precision" : " is not a checkpoint"));
    }
    const size_t bytes = (size_t) width * height * 3 * (is_fp64 ? 8 : 4);
    std::vector<char> host(bytes);
    const bool ok = std::fread(host.data(), 1, bytes, f) == bytes;
    std::fclose(f);
    if (!ok) return fail(RT_EINVAL, std::string("short read from ") + path);
    HIP_TRY(hipMemcpy(accum_dev, host.data(), bytes, hipMemcpyHostToDevice));
    *spp_done = h.spp_done;
    if (seed) *seed = h.seed;
    return RT_OK;
}

int rt_debug_path(rt_scene* s, int32_t width, int32_t height, int32_t i, int32_t j, int32_t k, uint64_t seed, int32_t max_depth, int32_t flags,
                  double* rows9_host, int32_t cap, int32_t* n_out_host)
{
    if (!s || !rows9_host || !n_out_host || cap <= 0) return fail(RT_EINVAL, "scene, output rows and their capacity are required");
    if (width < 2 || height < 2 || i < 0 || i >= height || j < 0 || j >= width || k < 0 || max_depth < 0) return fail(RT_EINVAL, "pixel outside the image, or a negative sample index or depth");
    if (int rc_dev = require_scene_device(s)) return rc_dev;
    rt::RenderArgs a{};
    a.width = width; a.height = height; a.max_depth = max_depth; a.flags = flags; a.seed = seed;
    double* d_rows = nullptr; int* d_n = nullptr;
    int rc = RT_OK;
    auto chk = [&](hipError_t err, const char* what) { if (err != hipSuccess && rc == RT_OK) rc = fail(RT_EHIP, std::string(what) + ": " + hipGetErrorString(err)); };
    chk(hipMalloc((void**) &d_rows, (size_t) cap * 9 * sizeof(double)), "hipMalloc");
    chk(hipMalloc((void**) &d_n, sizeof(int)), "hipMalloc");
    if (rc == RT_OK) chk(rt::launch_debug_path(s->d64.view, a, i, j, k, d_rows, cap, d_n, nullptr), "path kernel launch");
    if (rc == RT_OK) chk(hipDeviceSynchronize(), "path kernel");
    int n = 0;
    if (rc == RT_OK) chk(hipMemcpy(&n, d_n, sizeof(int), hipMemcpyDeviceToHost), "download");
    if (rc == RT_OK) chk(hipMemcpy(rows9_host, d_rows, (size_t) std::min(n, cap) * 9 * sizeof(double), hipMemcpyDeviceToHost), "download");
    *n_out_host = n;
    (void) hipFree(d_rows); (void) hipFree(d_n);
    return rc;
}

int rt_debug_libm(const int32_t* op, const double* x, const double* y, int32_t n, double* out_host)
{
    if (n < 0 || (n && (!op || !x || !y || !out_host))) return fail(RT_EINVAL, "n >= 0 and the four arrays are required");
    if (n == 0) return RT_OK;
    int32_t* d_op = nullptr; double* d_v = nullptr;
    int rc = RT_OK;
    auto chk = [&](hipError_t err, const char* what) { if (err != hipSuccess && rc == RT_OK) rc = fail(RT_EHIP, std::string(what) + ": " + hipGetErrorString(err)); };
    chk(hipMalloc((void**) &d_op, (size_t) n * 4), "hipMalloc");
    chk(hipMalloc((void**) &d_v, (size_t) n * 3 * 8), "hipMalloc");
    if (rc == RT_OK) chk(hipMemcpy(d_op, op, (size_t) n * 4, hipMemcpyHostToDevice), "upload");
    if (rc == RT_OK) chk(hipMemcpy(d_v, x, (size_t) n * 8, hipMemcpyHostToDevice), "upload");
    if (rc == RT_OK) chk(hipMemcpy(d_v + n, y, (size_t) n * 8, hipMemcpyHostToDevice), "upload");
    if (rc == RT_OK) chk(rt::launch_debug_libm(d_op, d_v, d_v + n, n, d_v + 2 * (size_t) n, nullptr), "libm kernel launch");
    if (rc == RT_OK) chk(hipDeviceSynchronize(), "libm kernel");
    if (rc == RT_OK) chk(hipMemcpy(out_host, d_v + 2 * (size_t) n, (size_t) n * 8, hipMemcpyDeviceToHost), "download");
    (void) hipFree(d_op); (void) hipFree(d_v);
    return rc;
}

int rt_debug_rng(uint64_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out_host)
{
    if (n <= 0 || !out_host) return fail(RT_EINVAL, "n > 0 and out_host are required");
    uint32_t* d = nullptr;
    HIP_TRY(hipMalloc((void**) &d, (size_t) n * 4));
    hipLaunchKernelGGL(rng_kernel, dim3(1), dim3(64), 0, nullptr, seed, pixel, sample, n, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out_host, d, (size_t) n * 4, hipMemcpyDeviceToHost);
    (void) hipFree(d);
    if (e != hipSuccess) return fail(RT_EHIP, hipGetErrorString(e));
    return RT_OK;
}

}  // extern "C"
