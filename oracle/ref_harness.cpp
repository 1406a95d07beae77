// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, dev container only.
//
// Drives the *real* reference CPU renderer (ccronix/raytracing, src/cpp) to
// produce golden fixtures.  The reference sources are NOT copied: this TU
// #includes them where they lie (-I /root/reference/src/cpp -I /root/reference/include)
// and the binary goes to oracle/_ref/ (git-ignored).  See oracle/Makefile.
//
// What comes from the reference unmodified: every class (scene, group, bvh,
// aabb, sphere, msphere, planexz, flip, triangle, materials, textures, camera,
// loader), `random_scene()`, and the integrator `trace()` (raytracing.cpp:30-70).
// What this harness adds: scene set-up for the BASELINE configs (the reference
// hard-codes the Cornell box), a pixel loop with the same statements as
// raytracing.cpp:96-123 but returning the fp64 radiance as well, dump code, and
// an instrumented re-walk of bvh::intersect (bvh.hpp:58-68) that calls the
// reference's own aabb::intersect / leaf intersect and records the visit order.

#include <omp.h>
#include <cassert>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <climits>
#include <limits>
#include <algorithm>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include <emmintrin.h>

// the reference keeps triangle / camera / flip fields private; the dump code
// needs to read them.  All std headers are already included above.
#define private public
#define main reference_main
#include "raytracing.cpp"
#undef main
#undef private

// ---------------------------------------------------------------- output
static std::string g_outdir = ".";

static void dump(const char* name, const char* dtype, const void* data, size_t count, size_t elem)
{
    std::string path = g_outdir + "/" + name + ".bin";
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    fwrite(data, elem, count, f);
    fclose(f);
    printf("FIXTURE %s %s %zu\n", name, dtype, count);
}
static void dump_f64(const char* n, const std::vector<double>& v) { dump(n, "f8", v.data(), v.size(), 8); }
static void dump_i32(const char* n, const std::vector<int32_t>& v) { dump(n, "i4", v.data(), v.size(), 4); }
static void dump_u64(const char* n, const std::vector<uint64_t>& v) { dump(n, "u8", v.data(), v.size(), 8); }
static void dump_u8(const char* n, const std::vector<uint8_t>& v) { dump(n, "u1", v.data(), v.size(), 1); }

// private generator for harness-made inputs (never rand(): the reference owns that stream)
static uint64_t g_sm = 0x1234abcdULL;
static uint64_t sm_next() { uint64_t z = (g_sm += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static double sm_unit() { return (sm_next() >> 11) * (1.0 / 9007199254740992.0); }
static double sm_range(double a, double b) { return a + (b - a) * sm_unit(); }

// ---------------------------------------------------------------- tree dump
enum { K_NODE = 0, K_SPHERE = 1, K_MSPHERE = 2, K_PLANEXY = 3, K_PLANEXZ = 4, K_PLANEYZ = 5, K_TRIANGLE = 6, K_BOX = 7, K_VOLUME = 8 };
enum { ROW = 56 };   // doubles per pre-order entry
enum { K_GROUP = 9 };  // a group used as an object (group.hpp:12): one entry, like an instance
// the `nested` fixture also dumps every entry's wrapper chain (row[28]: base-4 digits outermost first, 1 flip 2 rotatey 3 translate;
// row[29]: a position-weighted sum of the wrappers' parameters) and a volume's boundary (g[5], g[6..11]); the older fixtures' rows
// stay as they were made
static bool g_wrapper_rows = false;

struct pentry {
    const object* o; const object* base; int flipped; int left; int right;
    int xf = 0; double sin_t = 0, cos_t = 1; vec3d target;     // translate(rotatey(box)) wrappers: bit 1 rotatey, bit 2 translate
};
// strips flip / translate / rotatey wrappers (transform.hpp), outermost first
static void unwrap(pentry& e)
{
    for (;;) {
        if (const flip* f = dynamic_cast<const flip*>(e.base)) { e.base = f->obj; e.flipped ^= 1; }
        else if (const translate* t = dynamic_cast<const translate*>(e.base)) { e.base = t->obj; e.target = t->target; e.xf |= 4; }
        else if (const rotatey* r = dynamic_cast<const rotatey*>(e.base)) { e.base = r->obj; e.sin_t = r->sin_theta; e.cos_t = r->cos_theta; e.xf |= 2; }
        else break;
    }
}

struct tree_dump {
    std::vector<pentry> pre;
    std::map<const object*, int> first;   // object -> first pre-order index

    int walk(const object* o)
    {
        int idx = (int) pre.size();
        pentry e; e.o = o; e.base = o; e.flipped = 0; e.left = e.right = -1;
        unwrap(e);
        pre.push_back(e);
        if (!first.count(o)) first[o] = idx;
        if (const bvh* b = dynamic_cast<const bvh*>(e.base)) {          // also a bvh under translate / rotatey (an instance)
            int l = walk(b->left);
            int r = walk(b->right);
            pre[idx].left = l;
            pre[idx].right = r;
        }
        return idx;
    }

    static void put_tex(const texture* t, double* row)   // row[0]=tkind, row[1..]
    {
        if (const constant* c = dynamic_cast<const constant*>(t)) {
            row[0] = 1; row[1] = c->base_color.x(); row[2] = c->base_color.y(); row[3] = c->base_color.z();
        } else if (const checker* c = dynamic_cast<const checker*>(t)) {
            row[0] = 2;
            const constant* e = dynamic_cast<const constant*>(c->even);
            const constant* o = dynamic_cast<const constant*>(c->odds);
            row[1] = e->base_color.x(); row[2] = e->base_color.y(); row[3] = e->base_color.z();
            row[4] = o->base_color.x(); row[5] = o->base_color.y(); row[6] = o->base_color.z();
        } else if (const image* im = dynamic_cast<const image*>(t)) {
            row[0] = 3; row[1] = im->width; row[2] = im->height; row[3] = im->channels; row[4] = im->color_based;
        } else {
            row[0] = 0;
        }
    }

    static void put_mat(const material* m, double* row)  // row[0]=mkind, row[1]=tkind..., layout below
    {
        // [0] mkind [1] roughness/ior [2..8] texture [9..] phong
        if (const lambertian* l = dynamic_cast<const lambertian*>(m)) { row[0] = 1; put_tex(l->base_color, row + 2); }
        else if (const metal* l = dynamic_cast<const metal*>(m)) { row[0] = 2; row[1] = l->roughness; put_tex(l->base_color, row + 2); }
        else if (const dielectric* l = dynamic_cast<const dielectric*>(m)) { row[0] = 3; row[1] = l->ior; }
        else if (const emissive* l = dynamic_cast<const emissive*>(m)) { row[0] = 4; put_tex(l->emission, row + 2); }
        else if (const fog* l = dynamic_cast<const fog*>(m)) { row[0] = 6; put_tex(l->base_color, row + 2); }
        else if (const phong* p = dynamic_cast<const phong*>(m)) {
            row[0] = 5;
            if (p->diffuse_map) put_tex(p->diffuse_map, row + 2);
            for (int i = 0; i < 3; i++) { row[9 + i] = p->Kd[i]; row[12 + i] = p->Ks[i]; row[15 + i] = p->Ke[i]; row[18 + i] = p->Tr[i]; }
            row[21] = p->Ns; row[22] = p->Ni;
        }
    }

    std::vector<double> rows() const
    {
        std::vector<double> out(pre.size() * ROW, 0.0);
        for (size_t i = 0; i < pre.size(); i++) {
            double* r = &out[i * ROW];
            const pentry& e = pre[i];
            r[1] = e.flipped;
            r[2] = first.at(e.o);              // first occurrence of this object in pre-order
            double* g = r + 4;                 // geometry block, 24 doubles
            double* m = r + 30;                // material block
            const material* mat = nullptr;
            if (g_wrapper_rows) {
                double code = 0, sum = 0; int k = 1;
                for (const object* w = e.o; ; k++) {
                    if (const flip* f = dynamic_cast<const flip*>(w)) { code = code * 4 + 1; w = f->obj; }
                    else if (const rotatey* ro = dynamic_cast<const rotatey*>(w)) { code = code * 4 + 2; sum += k * (ro->sin_theta + 2 * ro->cos_theta); w = ro->obj; }
                    else if (const translate* tr = dynamic_cast<const translate*>(w)) { code = code * 4 + 3; sum += k * (tr->target.x() + 2 * tr->target.y() + 3 * tr->target.z()); w = tr->obj; }
                    else break;
                }
                r[28] = code; r[29] = sum;
            }
            if (const group* gr = dynamic_cast<const group*>(e.base)) { r[0] = K_GROUP; g[0] = gr->size(); continue; }
            if (const bvh* b = dynamic_cast<const bvh*>(e.base)) {
                r[0] = K_NODE;
                for (int k = 0; k < 3; k++) { g[k] = b->bbox.min()[k]; g[3 + k] = b->bbox.max()[k]; }
                g[6] = e.left; g[7] = e.right;
                if (e.xf) { r[3] = e.xf; g[8] = e.sin_t; g[9] = e.cos_t; for (int k = 0; k < 3; k++) g[10 + k] = e.target[k]; }
            } else if (const sphere* s = dynamic_cast<const sphere*>(e.base)) {
                r[0] = K_SPHERE; for (int k = 0; k < 3; k++) g[k] = s->center[k]; g[3] = s->radius; mat = s->mat;
            } else if (const msphere* s = dynamic_cast<const msphere*>(e.base)) {
                r[0] = K_MSPHERE; for (int k = 0; k < 3; k++) { g[k] = s->center0[k]; g[3 + k] = s->center1[k]; }
                g[6] = s->start; g[7] = s->end; g[8] = s->radius; mat = s->mat;
            } else if (const planexy* p = dynamic_cast<const planexy*>(e.base)) {
                r[0] = K_PLANEXY; g[0] = p->x0; g[1] = p->x1; g[2] = p->y0; g[3] = p->y1; g[4] = p->k; mat = p->mat;
            } else if (const planexz* p = dynamic_cast<const planexz*>(e.base)) {
                r[0] = K_PLANEXZ; g[0] = p->x0; g[1] = p->x1; g[2] = p->z0; g[3] = p->z1; g[4] = p->k; mat = p->mat;
            } else if (const planeyz* p = dynamic_cast<const planeyz*>(e.base)) {
                r[0] = K_PLANEYZ; g[0] = p->y0; g[1] = p->y1; g[2] = p->z0; g[3] = p->z1; g[4] = p->k; mat = p->mat;
            } else if (const triangle* t = dynamic_cast<const triangle*>(e.base)) {
                r[0] = K_TRIANGLE;
                for (int k = 0; k < 3; k++) {
                    g[k] = t->vertices[0].position[k]; g[3 + k] = t->p0p1[k]; g[6 + k] = t->p0p2[k];
                    g[9 + k] = t->vertices[0].normal[k]; g[12 + k] = t->vertices[1].normal[k]; g[15 + k] = t->vertices[2].normal[k];
                }
                for (int v = 0; v < 3; v++) { g[18 + 2 * v] = t->vertices[v].uv_coord.content[0]; g[19 + 2 * v] = t->vertices[v].uv_coord.content[1]; }
                mat = t->mat;
            } else if (const box* bx = dynamic_cast<const box*>(e.base)) {
                // box (primitive.hpp:399-458) under translate(rotatey(.)) (transform.hpp:9-121): one leaf of the bvh
                r[0] = K_BOX; r[3] = e.xf;
                for (int k = 0; k < 3; k++) { g[k] = bx->minimum[k]; g[3 + k] = bx->maximum[k]; g[8 + k] = e.target[k]; }
                g[6] = e.sin_t; g[7] = e.cos_t;
                aabb bb; e.o->bounding_box(0, 0, bb);          // the wrapped object's box, as the bvh builder sees it
                for (int k = 0; k < 3; k++) { g[11 + k] = bb.min()[k]; g[14 + k] = bb.max()[k]; }
                mat = dynamic_cast<const planexy*>(bx->xy0)->mat;
            } else if (const volume* vo = dynamic_cast<const volume*>(e.base)) {
                // volume (primitive.hpp:329-396) around a sphere boundary: one leaf; the boundary's own material is not used
                const sphere* sb = dynamic_cast<const sphere*>(vo->surface);
                r[0] = K_VOLUME;
                if (sb) { for (int k = 0; k < 3; k++) g[k] = sb->center[k]; g[3] = sb->radius; }
                g[4] = vo->neg_inv_d; mat = vo->mat;
                if (g_wrapper_rows) {                                   // the boundary: wrappers (same digits), then sphere (1) / box (7) and a box's corners
                    double code = 0;
                    const object* w = vo->surface;
                    for (;;) {
                        if (const flip* f = dynamic_cast<const flip*>(w)) { code = code * 4 + 1; w = f->obj; }
                        else if (const rotatey* ro = dynamic_cast<const rotatey*>(w)) { code = code * 4 + 2; w = ro->obj; }
                        else if (const translate* tr = dynamic_cast<const translate*>(w)) { code = code * 4 + 3; w = tr->obj; }
                        else break;
                    }
                    const box* bb = dynamic_cast<const box*>(w);
                    g[5] = code * 16 + (dynamic_cast<const sphere*>(w) ? 1 : bb ? 7 : 0);
                    if (bb) for (int k = 0; k < 3; k++) { g[6 + k] = bb->minimum[k]; g[9 + k] = bb->maximum[k]; }
                }
            } else {
                r[0] = -1;
            }
            if (mat) put_mat(mat, m);
        }
        return out;
    }
};

// ---------------------------------------------------------------- instrumented re-walk of bvh::intersect
struct walk_stats { uint32_t nbox, nleaf; uint64_t hash; int hit_entry; };

static inline void fnv(uint64_t& h, uint32_t v) { h = (h ^ (uint64_t) v) * 0x100000001B3ULL; }

static bool rewalk(const tree_dump& td, int idx, const ray& r, intersection& rec, walk_stats& st)
{
    const pentry& e = td.pre[idx];
    fnv(st.hash, (uint32_t) idx);
    if (const bvh* b = dynamic_cast<const bvh*>(e.o)) {
        st.nbox++;
        if (!b->bbox.intersect(r)) return false;                     // bvh.hpp:60
        bool hl = rewalk(td, e.left, r, rec, st);                    // bvh.hpp:64
        double tmr = hl ? rec.t : r.t_max;                           // bvh.hpp:65
        bool hr = rewalk(td, e.right, ray(r.origin, r.direction, r.time, r.t_min, tmr), rec, st);
        return hl || hr;
    }
    st.nleaf++;
    bool hit = e.o->intersect(r, rec);
    if (hit) st.hit_entry = idx;
    return hit;
}

// ---------------------------------------------------------------- scenes
struct built_scene { scene* scn; tree_dump td; const object* root; };

static camera c0_camera(int w, int h)
{
    return camera(vec3d(13, 2, 3), vec3d(0, 0, 0), vec3d(0, 1, 0), 20, double(w) / h, 0.1, 10, 0, 1);
}

// BASELINE config 0/1 scene: random_scene() (demo.hpp:17-60) + one overhead area light, because the
// reference has no sky and crashes on an empty light list.  Light geometry is added to
// the objects AND, as a second instance, to the lights, exactly as raytracing.cpp:81-85 does.
static built_scene* build_c0(int w, int h)
{
    built_scene* b = new built_scene();
    b->scn = new scene(w, h);
    b->scn->add_object(random_scene());
    b->scn->add_object(new flip(new planexz(-20, 20, -20, 20, 30, new emissive(vec3d(4, 4, 4)))));
    b->scn->add_light(new flip(new planexz(-20, 20, -20, 20, 30, new emissive(vec3d(4, 4, 4)))));
    b->scn->set_camera(c0_camera(w, h));
    b->scn->build_bvh();
    b->root = b->scn->objects.content()[0];
    b->td.walk(b->root);
    return b;
}

// small hand-made scene: every * primitive and material type in one bvh, two lights (rect + triangle)
static built_scene* build_mixed(int w, int h)
{
    built_scene* b = new built_scene();
    b->scn = new scene(w, h);
    group grp;
    texture* chk = new checker(vec3d(0.2, 0.3, 0.1), vec3d(0.9, 0.9, 0.9));
    grp.add(new sphere(vec3d(0, -100.5, -1), 100, new lambertian(chk)));
    grp.add(new sphere(vec3d(0, 0, -1), 0.5, new lambertian(vec3d(0.1, 0.2, 0.5))));
    grp.add(new sphere(vec3d(-1.05, 0, -1), 0.5, new dielectric(1.5)));
    grp.add(new sphere(vec3d(1.05, 0, -1), 0.5, new metal(vec3d(0.8, 0.6, 0.2), 0.3)));
    grp.add(new msphere(vec3d(0.3, 0.7, -0.6), vec3d(0.3, 0.9, -0.6), 0, 1, 0.15, new lambertian(vec3d(0.7, 0.2, 0.2))));
    grp.add(new sphere(vec3d(-0.4, -0.3, -0.2), 0.2, new metal(vec3d(0.9, 0.9, 0.9), 0.0)));
    material* tri_mat = new lambertian(vec3d(0.3, 0.7, 0.3));
    grp.add(new triangle(vec3d(-2, -0.5, -2.5), vec3d(2, -0.5, -2.5), vec3d(0, 2.0, -2.8), tri_mat));
    grp.add(new triangle(vec3d(1.2, -0.5, 0.2), vec3d(1.9, -0.5, -0.4), vec3d(1.5, 0.6, -0.1), new metal(vec3d(0.7, 0.7, 0.9), 0.1)));
    phong* ph = new phong(vec3d(0.6, 0.5, 0.2));
    grp.add(new triangle(vec3d(-1.9, -0.5, 0.3), vec3d(-1.2, -0.5, 0.1), vec3d(-1.6, 0.5, 0.0), ph));
    grp.add(new planexy(-3, 3, -0.5, 3, -3.2, new lambertian(vec3d(0.6, 0.6, 0.7))));
    grp.add(new planeyz(-0.5, 3, -3.2, 1, -3, new lambertian(vec3d(0.7, 0.3, 0.3))));
    grp.add(new flip(new planexz(-1, 1, -2, 0, 3, new emissive(vec3d(6, 6, 6)))));
    phong* lamp = new phong(vec3d(0, 0, 0));
    lamp->Ke = vec3d(5, 4, 3);
    triangle* lamp_tri = new triangle(vec3d(2.5, 2.5, -1.5), vec3d(2.5, 0.5, -0.5), vec3d(2.5, 0.5, -2.5), lamp);
    grp.add(lamp_tri);
    b->scn->add_object(grp);
    b->scn->add_light(new flip(new planexz(-1, 1, -2, 0, 3, new emissive(vec3d(6, 6, 6)))));
    b->scn->add_light(lamp_tri);    // aliased instance, as loader.hpp:124-128 does for emissive OBJ faces
    b->scn->set_camera(camera(vec3d(-0.5, 1.2, 3.5), vec3d(0, 0.2, -1), vec3d(0, 1, 0), 45, double(w) / h, 0.05, 4.5, 0, 1));
    b->scn->build_bvh();
    b->root = b->scn->objects.content()[0];
    b->td.walk(b->root);
    return b;
}

// the scene of the reference's own main(): raytracing.cpp:77-88 (cornell_box(), demo.hpp:63-91)
static built_scene* build_cornell(int w, int h)
{
    built_scene* b = new built_scene();
    camera cam = camera(vec3d(278, 278, -800), vec3d(278, 278, 0), vec3d(0, 1, 0), 40, 1, 0, 1, 0, 1);
    object* light = new flip(new planexz(213, 343, 227, 332, 554, new emissive(vec3d(30, 30, 30))));
    b->scn = new scene(w, h);
    b->scn->add_object(cornell_box());
    b->scn->add_light(light);
    b->scn->set_camera(cam);
    b->scn->build_bvh();
    b->root = b->scn->objects.content()[0];
    b->td.walk(b->root);
    return b;
}

// volume / fog (primitive.hpp:329-396, material.hpp:165-181): the sphere-bounded media of all_feature_test() (demo.hpp:117-134:
// global haze in a radius-5000 sphere, a glass ball filled with blue fog, moving / glass / fuzzy-metal / mirror spheres) over a ground
// rect, plus a free-standing smoke ball; the light is flipped so that it shines downwards (the demo's is not, and lights nothing)
static built_scene* build_smoke(int w, int h)
{
    built_scene* b = new built_scene();
    b->scn = new scene(w, h);
    group grp;
    grp.add(new planexz(-1000, 1000, -1000, 1000, 0, new lambertian(vec3d(0.48, 0.83, 0.53))));
    grp.add(new flip(new planexz(123, 423, 147, 412, 554, new emissive(vec3d(7, 7, 7)))));
    grp.add(new msphere(vec3d(400, 400, 200), vec3d(430, 400, 200), 0, 1, 50, new lambertian(vec3d(0.7, 0.3, 0.1))));
    grp.add(new sphere(vec3d(260, 150, 45), 50, new dielectric(1.5)));
    grp.add(new sphere(vec3d(0, 150, 145), 50, new metal(vec3d(0.8, 0.8, 0.9), 1)));
    grp.add(new volume(new sphere(vec3d(0, 0, 0), 5000, new dielectric(1.5)), 0.0002, vec3d(1, 1, 1)));
    object* ball = new sphere(vec3d(360, 150, 145), 70, new dielectric(1.5));
    grp.add(ball);
    grp.add(new volume(ball, 0.2, vec3d(0.2, 0.4, 0.9)));
    grp.add(new sphere(vec3d(400, 200, 400), 100, new lambertian(new checker(vec3d(0.2, 0.3, 0.1), vec3d(0.9, 0.9, 0.9)))));
    grp.add(new sphere(vec3d(220, 280, 300), 50, new metal(vec3d(1, 1, 1), 0)));
    grp.add(new volume(new sphere(vec3d(150, 100, 250), 80, new dielectric(1.5)), 0.01, vec3d(0.9, 0.9, 0.9)));
    b->scn->add_object(grp);
    b->scn->add_light(new flip(new planexz(123, 423, 147, 412, 554, new emissive(vec3d(7, 7, 7)))));
    b->scn->set_camera(camera(vec3d(478, 278, -600), vec3d(278, 278, 0), vec3d(0, 1, 0), 40, double(w) / h, 0, 10, 0, 1));
    b->scn->build_bvh();
    b->root = b->scn->objects.content()[0];
    b->td.walk(b->root);
    return b;
}

// all_feature_test() (demo.hpp:94-143) as the reference builds it, under a light and a camera of the harness' choosing (the
// reference's main() never renders this scene).  scene::build_bvh() wraps the single object in a span-1 bvh: both children
// are the same subtree, and the reference walks it twice.
static built_scene* build_features(int w, int h)
{
    built_scene* b = new built_scene();
    b->scn = new scene(w, h);
    b->scn->add_object(all_feature_test());
    b->scn->add_light(new planexz(123, 423, 147, 412, 554, new emissive(vec3d(7, 7, 7))));
    b->scn->set_camera(camera(vec3d(478, 278, -600), vec3d(278, 278, 0), vec3d(0, 1, 0), 40, double(w) / h, 0, 10, 0, 1));
    b->scn->build_bvh();
    b->root = b->scn->objects.content()[0];
    b->td.walk(b->root);
    return b;
}

// Everything the reference's object model composes and the demos never do: wrappers nested every way (transform.hpp:9-149), volumes
// around boxes (primitive.hpp:332-343 takes any object*), groups used as objects (group.hpp:12,68-102), an instance inside an
// instance.  Every leaf gets a material of its own, registered in construction order: the hit record's `mat` then names the LEAF
// that was hit, also inside a wrapper, a group or an instance (fixture array <tag>_fh_prim).
static std::map<const material*, int> g_leaf_of_mat;
static material* leaf_mat(material* m) { int n = (int) g_leaf_of_mat.size(); g_leaf_of_mat[m] = n; return m; }
static material* leaf_lam(double r, double g, double b) { return leaf_mat(new lambertian(vec3d(r, g, b))); }
static object* leaf_volume(object* surface, double density, const vec3d& color) { volume* v = new volume(surface, density, color); leaf_mat(v->mat); return v; }

static built_scene* build_nested(int w, int h)
{
    built_scene* b = new built_scene();
    b->scn = new scene(w, h);
    group grp;
    // 0 ground, 1 lamp
    grp.add(new planexz(-40, 40, -40, 40, 0, leaf_lam(0.5, 0.5, 0.5)));
    grp.add(new flip(new planexz(-4, 4, -4, 4, 14, leaf_mat(new emissive(vec3d(8, 8, 8))))));
    // 2 translate(translate(sphere))
    grp.add(new translate(new translate(new sphere(vec3d(0, 1, 0), 1, leaf_lam(0.8, 0.3, 0.3)), vec3d(-6, 0, 0)), vec3d(0, 0, -3)));
    // 3 translate(rotatey(rotatey(box)))
    grp.add(new translate(new rotatey(new rotatey(new box(vec3d(-1, 0, -1), vec3d(1, 2, 1), leaf_lam(0.3, 0.8, 0.3)), 20), 25), vec3d(-2, 0, 4)));
    // 4 translate(rotatey(translate(rotatey(sphere)))), glass
    grp.add(new translate(new rotatey(new translate(new rotatey(new sphere(vec3d(0, 1, 0), 1, leaf_mat(new dielectric(1.5))), 30), vec3d(3, 0, 0)), 40), vec3d(2, 0, -6)));
    // 5 rotatey(translate(sphere)), metal
    grp.add(new rotatey(new translate(new sphere(vec3d(0, 1.2, 0), 1.2, leaf_mat(new metal(vec3d(0.8, 0.8, 0.6), 0.1))), vec3d(7, 0, 0)), 50));
    // 6 volume(box)
    grp.add(leaf_volume(new box(vec3d(-9, 0, 2), vec3d(-6, 3, 5), new dielectric(1.5)), 0.4, vec3d(0.9, 0.9, 0.9)));
    // 7 volume(translate(rotatey(box)))
    grp.add(leaf_volume(new translate(new rotatey(new box(vec3d(0, 0, 0), vec3d(2.5, 3, 2.5), new dielectric(1.5)), -18), vec3d(5, 0, 3)), 0.5, vec3d(0.2, 0.4, 0.9)));
    // 8 translate(volume(sphere))
    grp.add(new translate(leaf_volume(new sphere(vec3d(0, 1.5, 0), 1.5, new dielectric(1.5)), 0.6, vec3d(0.9, 0.5, 0.2)), vec3d(-3, 0, -8)));
    // 9-11 a group as an object
    {
        group* sub = new group();
        sub->add(new sphere(vec3d(-10, 1, -6), 1, leaf_lam(0.7, 0.7, 0.2)));
        sub->add(new triangle(vec3d(-12, 0, -4), vec3d(-9, 0, -4), vec3d(-10.5, 3, -4.5), leaf_lam(0.2, 0.7, 0.7)));
        sub->add(new box(vec3d(-13, 0, -9), vec3d(-11, 2, -7), leaf_lam(0.7, 0.2, 0.7)));
        grp.add(sub);
    }
    // 12-13 translate(group)
    {
        group* sub = new group();
        sub->add(new sphere(vec3d(0, 1, 0), 1, leaf_lam(0.9, 0.6, 0.3)));
        sub->add(new sphere(vec3d(2.2, 1, 0), 1, leaf_mat(new metal(vec3d(0.9, 0.9, 0.9), 0.0))));
        grp.add(new translate(sub, vec3d(8, 0, -9)));
    }
    // 14-18 an instance inside an instance: translate(rotatey(bvh{ sphere, translate(bvh{3 spheres}), sphere }))
    {
        group in;
        in.add(new sphere(vec3d(0, 0.6, 0), 0.6, leaf_lam(0.9, 0.2, 0.2)));
        in.add(new sphere(vec3d(1.5, 0.6, 0), 0.6, leaf_lam(0.2, 0.9, 0.2)));
        in.add(new sphere(vec3d(0, 0.6, 1.5), 0.6, leaf_lam(0.2, 0.2, 0.9)));
        object* inner = new bvh(in);
        group mid;
        mid.add(new sphere(vec3d(-2, 1, 0), 1, leaf_lam(0.6, 0.6, 0.6)));
        mid.add(new translate(inner, vec3d(0, 2.2, 0)));
        mid.add(new sphere(vec3d(2.5, 1, 1), 1, leaf_mat(new dielectric(1.5))));
        grp.add(new translate(new rotatey(new bvh(mid), 35), vec3d(-9, 0, 9)));
    }
    // 19 flip(translate(sphere)): the flip counts.  20 translate(flip(sphere)): translate's set_face_normal overwrites it
    grp.add(new flip(new translate(new sphere(vec3d(0, 1, 0), 1, leaf_mat(new dielectric(1.5))), vec3d(4, 0, 8))));
    grp.add(new translate(new flip(new sphere(vec3d(0, 1, 0), 1, leaf_mat(new dielectric(1.5)))), vec3d(7, 0, 7)));
    // 21-22 flip(bvh{sphere, rect})
    {
        group m;
        m.add(new sphere(vec3d(11, 1, 2), 1, leaf_mat(new dielectric(1.5))));
        m.add(new planexy(9, 13, 0, 3, 0, leaf_lam(0.4, 0.4, 0.8)));
        grp.add(new flip(new bvh(m)));
    }
    // 23 translate(msphere)
    grp.add(new translate(new msphere(vec3d(0, 1, 0), vec3d(0, 1.5, 0), 0, 1, 0.8, leaf_lam(0.7, 0.3, 0.1)), vec3d(-5, 0, 6)));
    b->scn->add_object(grp);
    b->scn->add_light(new flip(new planexz(-4, 4, -4, 4, 14, new emissive(vec3d(8, 8, 8)))));
    b->scn->set_camera(camera(vec3d(2, 9, 26), vec3d(0, 1, 0), vec3d(0, 1, 0), 40, double(w) / h, 0.1, 25, 0, 1));
    b->scn->build_bvh();
    b->root = b->scn->objects.content()[0];
    b->td.walk(b->root);
    return b;
}

// OBJ scene through the reference loader (loader.hpp:20-38)
static built_scene* build_obj(const char* path, int w, int h)
{
    built_scene* b = new built_scene();
    b->scn = new scene(w, h);
    loader* ld = new loader(path);
    b->scn->add_object(ld->meshes());
    b->scn->add_light(*ld->lights());
    b->scn->set_camera(camera(vec3d(0, 1.6, 4.2), vec3d(0, 0.6, 0), vec3d(0, 1, 0), 40, double(w) / h, 0, 1, 0, 1));
    b->scn->build_bvh();
    b->root = b->scn->objects.content()[0];
    b->td.walk(b->root);
    return b;
}

static void dump_camera(const char* name, const camera& c)
{
    std::vector<double> v;
    auto put = [&](const vec3d& a) { v.push_back(a.x()); v.push_back(a.y()); v.push_back(a.z()); };
    put(c.position); put(c.left_bottom); put(c.horizontal); put(c.vertical); put(c.u); put(c.v); put(c.w);
    v.push_back(c.lens_radius); v.push_back(c.shutter_start); v.push_back(c.shutter_end);
    dump_f64(name, v);
}

static void dump_lights(const char* name, const scene& scn)
{
    tree_dump td;
    for (int i = 0; i < scn.lights->size(); i++) {
        const object* o = scn.lights->content()[i];
        pentry e; e.o = o; e.base = o; e.flipped = 0; e.left = e.right = -1;
        unwrap(e);
        td.first[o] = (int) td.pre.size();
        td.pre.push_back(e);
    }
    dump_f64(name, td.rows());
}

// ---------------------------------------------------------------- first-hit fixture
static void do_firsthit(const std::string& tag, built_scene* b, int n_primary_side, int n_random, double lo[3], double hi[3], bool cross_check = true)
{
    std::vector<ray> rays;
    // (a) primary rays through the reference camera (lens + shutter randoms come from rand())
    for (int i = 0; i < n_primary_side; i++)
        for (int j = 0; j < n_primary_side; j++)
            rays.push_back(b->scn->render_camera.emit((j + 0.5) / n_primary_side, (i + 0.5) / n_primary_side));
    // (b) incoherent secondary-like rays
    for (int i = 0; i < n_random; i++) {
        vec3d o(sm_range(lo[0], hi[0]), sm_range(lo[1], hi[1]), sm_range(lo[2], hi[2]));
        vec3d d(sm_range(-1, 1), sm_range(-1, 1), sm_range(-1, 1));
        rays.push_back(ray(o, d, sm_unit()));
    }
    // (c) edge cases: axis-parallel directions (zero components => inv = +-inf, NaN slabs), zero direction
    for (int i = 0; i < 96; i++) {
        vec3d o(sm_range(lo[0], hi[0]), sm_range(lo[1], hi[1]), sm_range(lo[2], hi[2]));
        double s = (i & 1) ? -1.0 : 1.0;
        vec3d d;
        switch ((i >> 1) % 8) {
            case 0: d = vec3d(s, 0, 0); break;
            case 1: d = vec3d(0, s, 0); break;
            case 2: d = vec3d(0, 0, s); break;
            case 3: d = vec3d(s, s * 0.5, 0); break;
            case 4: d = vec3d(0, -s * 2, s); break;
            case 5: d = vec3d(-0.0, s, -0.0); break;
            case 6: d = vec3d(s * 1e-300, -1, s * 1e-300); break;
            default: d = vec3d(0, 0, 0); break;
        }
        if ((i >> 4) & 1) { o = vec3d(std::floor(o.x()), 0.2, std::floor(o.z())); }   // lattice-aligned origins
        rays.push_back(ray(o, d, 0.25 * (i % 5)));
    }

    size_t n = rays.size();
    std::vector<double> rin(n * 7), tout(n), rec(n * 9);
    std::vector<int32_t> hit(n), nbox(n), nleaf(n), front(n), leaf(n);
    std::vector<uint64_t> hash(n);
    size_t mism = 0;
    for (size_t i = 0; i < n; i++) {
        const ray& r = rays[i];
        for (int k = 0; k < 3; k++) { rin[i * 7 + k] = r.origin[k]; rin[i * 7 + 3 + k] = r.direction[k]; }
        rin[i * 7 + 6] = r.time;
        walk_stats st; st.nbox = st.nleaf = 0; st.hash = 0xcbf29ce484222325ULL; st.hit_entry = -1;
        intersection x; x.t = 0; x.frontward = false;
        bool h = rewalk(b->td, 0, r, x, st);
        // cross-check against the reference's own entry point (group.hpp:68-83 -> bvh.hpp:58-68)
        // (not for scenes with volumes: volume::intersect draws from rand(), so a second walk sees other free-flight distances)
        if (cross_check) {
            intersection y; y.t = 0;
            bool h2 = b->scn->objects.intersect(r, y);
            if (h != h2 || (h && (memcmp(&x.t, &y.t, 8) != 0 || x.mat != y.mat))) mism++;
        }
        hit[i] = h ? b->td.first.at(b->td.pre[st.hit_entry].o) : -1;
        tout[i] = h ? x.t : 0.0;
        nbox[i] = st.nbox; nleaf[i] = st.nleaf; hash[i] = st.hash;
        front[i] = h ? (int) x.frontward : -1;
        leaf[i] = h && g_leaf_of_mat.count(x.mat) ? g_leaf_of_mat.at(x.mat) : -1;
        if (h) {
            for (int k = 0; k < 3; k++) { rec[i * 9 + k] = x.position[k]; rec[i * 9 + 3 + k] = x.normal[k]; }
            rec[i * 9 + 6] = x.uv_coord.content[0]; rec[i * 9 + 7] = x.uv_coord.content[1]; rec[i * 9 + 8] = 0;
        }
    }
    printf("INFO %s firsthit rays=%zu rewalk-vs-reference mismatches=%zu\n", tag.c_str(), n, mism);
    if (mism) { fprintf(stderr, "harness re-walk disagrees with reference\n"); exit(3); }
    dump_f64((tag + "_fh_rays").c_str(), rin);
    dump_i32((tag + "_fh_entry").c_str(), hit);
    dump_f64((tag + "_fh_t").c_str(), tout);
    dump_i32((tag + "_fh_nbox").c_str(), nbox);
    dump_i32((tag + "_fh_nleaf").c_str(), nleaf);
    dump_i32((tag + "_fh_front").c_str(), front);
    dump_u64((tag + "_fh_hash").c_str(), hash);
    dump_f64((tag + "_fh_rec").c_str(), rec);
    if (!g_leaf_of_mat.empty()) dump_i32((tag + "_fh_prim").c_str(), leaf);     // the leaf behind the hit record's material (build_nested)
}

// ---------------------------------------------------------------- image fixture
class ray_counter : public object {   // forwards to the scene root, counts trace() invocations that traverse
public:
    const object* inner; mutable uint64_t count = 0;
    ray_counter(const object* o) : inner(o) {}
    bool intersect(const ray& r, intersection& c) const override { count++; return inner->intersect(r, c); }
    bool bounding_box(double s, double e, aabb& b) const override { return inner->bounding_box(s, e, b); }
};

static void do_image(const std::string& tag, built_scene* b, int w, int h, int spp, int max_depth)
{
    scene& scn = *b->scn;
    ray_counter* rc = new ray_counter(b->root);
    scn.objects.content()[0] = rc;
    std::vector<double> radiance((size_t) w * h * 3);
    std::vector<uint8_t> data((size_t) w * h * 3);
    uint64_t nan_samples = 0;
    // same statements as raytracing.cpp:96-123, single thread (deterministic rand() stream)
    for (int i = 0; i < h; i++) {
        for (int j = 0; j < w; j++) {
            vec3d color = vec3d(0, 0, 0);
            for (int k = 0; k < spp; k++) {
                double x = double(j + random_double()) / (w - 1);
                double y = double(i + random_double()) / (h - 1);
                ray r = scn.render_camera.emit(x, y);
                vec3d sample = trace(scn, r, max_depth);
                if (is_nan(sample) || is_infinity(sample)) { sample = vec3d(0, 0, 0); nan_samples++; }
                color += sample;
            }
            color /= spp;
            int index = (i * w + j) * 3;
            radiance[index] = color.x(); radiance[index + 1] = color.y(); radiance[index + 2] = color.z();
            color = gamma(color, 0.45);
            color = clamp(color, 0, 1);
            data[index] = int(color.x() * 255);
            data[index + 1] = int(color.y() * 255);
            data[index + 2] = int(color.z() * 255);
        }
    }
    scn.objects.content()[0] = const_cast<object*>(b->root);
    std::vector<uint64_t> counters = { rc->count, nan_samples, (uint64_t) rand() };   // last: next rand() = stream position pin
    printf("INFO %s image %dx%d spp=%d depth=%d rays=%llu nan=%llu\n", tag.c_str(), w, h, spp, max_depth,
           (unsigned long long) rc->count, (unsigned long long) nan_samples);
    dump_f64((tag + "_img_radiance").c_str(), radiance);
    dump_u8((tag + "_img_u8").c_str(), data);
    dump_u64((tag + "_img_counters").c_str(), counters);
}

// ---------------------------------------------------------------- unit vectors
static void do_units()
{
    // aabb::intersect (aabb.hpp:23-41) incl. inf / NaN slabs
    {
        const int n = 512;
        std::vector<double> in(n * 14); std::vector<int32_t> out(n);
        for (int i = 0; i < n; i++) {
            double mn[3], mx[3], o[3], d[3];
            for (int k = 0; k < 3; k++) { double a = sm_range(-2, 2), c = sm_range(-2, 2); mn[k] = std::min(a, c); mx[k] = std::max(a, c); o[k] = sm_range(-3, 3); d[k] = sm_range(-1, 1); }
            if (i % 7 == 1) d[i % 3] = 0.0;
            if (i % 7 == 2) d[i % 3] = -0.0;
            if (i % 7 == 3) { d[i % 3] = 0.0; o[i % 3] = mn[i % 3]; }          // 0 * inf = NaN slab
            if (i % 7 == 4) { d[(i + 1) % 3] = 0.0; o[(i + 1) % 3] = mx[(i + 1) % 3]; }
            if (i % 11 == 5) { for (int k = 0; k < 3; k++) o[k] = 0.5 * (mn[k] + mx[k]); }  // origin inside
            double tmin = (i % 5 == 0) ? 1e-3 : sm_range(0, 0.5), tmax = (i % 3 == 0) ? infinity : sm_range(0.5, 6);
            aabb box(vec3d(mn[0], mn[1], mn[2]), vec3d(mx[0], mx[1], mx[2]));
            ray r(vec3d(o[0], o[1], o[2]), vec3d(d[0], d[1], d[2]), 0, tmin, tmax);
            double* p = &in[i * 14];
            for (int k = 0; k < 3; k++) { p[k] = mn[k]; p[3 + k] = mx[k]; p[6 + k] = o[k]; p[9 + k] = d[k]; }
            p[12] = tmin; p[13] = tmax;
            out[i] = box.intersect(r);
        }
        dump_f64("unit_aabb_in", in); dump_i32("unit_aabb_out", out);
    }
    // primitive intersect records: sphere, msphere, planexz, triangle (primitive.hpp:29-56,103-130,227-250; polygon.hpp:90-124)
    {
        const int n = 256;
        material* m = new lambertian(vec3d(0.5, 0.5, 0.5));
        sphere sp(vec3d(0.3, -0.2, 0.1), 1.25, m);
        msphere ms(vec3d(0.3, -0.2, 0.1), vec3d(0.3, 0.4, 0.1), 0, 1, 0.8, m);
        planexz px(-1.5, 1.0, -0.5, 2.0, 0.75, m);
        vertex v[3];
        v[0].position = vec3d(-1, -0.5, 0.2); v[1].position = vec3d(1.2, -0.4, -0.3); v[2].position = vec3d(0.1, 1.3, 0.4);
        v[0].uv_coord = vec2d(0.1, 0.2); v[1].uv_coord = vec2d(0.9, 0.15); v[2].uv_coord = vec2d(0.45, 0.95);
        v[0].normal = vec3d(0.1, 0.2, 0.97); v[1].normal = vec3d(-0.2, 0.1, 0.95); v[2].normal = vec3d(0.0, -0.3, 0.9);
        triangle tr(v, m);
        flip fpx(new planexz(-1.5, 1.0, -0.5, 2.0, 0.75, m));
        const object* objs[5] = { &sp, &ms, &px, &tr, &fpx };
        std::vector<double> in(n * 9), out(5 * n * 12), pdfs(5 * n);
        for (int i = 0; i < n; i++) {
            vec3d o(sm_range(-2.5, 2.5), sm_range(-2.5, 2.5), sm_range(-2.5, 2.5));
            vec3d tgt(sm_range(-1, 1), sm_range(-0.6, 1.0), sm_range(-0.5, 0.8));
            vec3d d = (i % 4 == 3) ? vec3d(sm_range(-1, 1), sm_range(-1, 1), sm_range(-1, 1)) : (tgt - o);
            if (i % 16 == 5) o = vec3d(0.3, -0.1, 0.2);   // inside both spheres
            double time = sm_unit(), tmax = (i % 6 == 0) ? sm_range(0.2, 3) : infinity;
            double* p = &in[i * 9];
            for (int k = 0; k < 3; k++) { p[k] = o[k]; p[3 + k] = d[k]; }
            p[6] = time; p[7] = 1e-3; p[8] = tmax;
            ray r(o, d, time, 1e-3, tmax);
            for (int q = 0; q < 5; q++) {
                intersection x; x.t = 0; x.frontward = false; x.mat = nullptr;
                bool h = objs[q]->intersect(r, x);
                double* w = &out[(q * n + i) * 12];
                w[0] = h;
                if (h) { w[1] = x.t; for (int k = 0; k < 3; k++) { w[2 + k] = x.position[k]; w[5 + k] = x.normal[k]; } w[8] = x.uv_coord.content[0]; w[9] = x.uv_coord.content[1]; w[10] = x.frontward; }
                pdfs[q * n + i] = objs[q]->pdf(o, d);   // object.hpp:16 default 0 for spheres; planexz primitive.hpp:258-269; triangle polygon.hpp:140-151
            }
        }
        dump_f64("unit_prim_in", in); dump_f64("unit_prim_out", out); dump_f64("unit_prim_pdf", pdfs);
    }
    // materials with the rand() stream re-seeded per case (material.hpp:29-52,74-88,105-133,211-266), and the
    // light-sampling step of trace (raytracing.cpp:56-60) via a 2-light group
    {
        const int n = 192;
        texture* chk = new checker(vec3d(0.2, 0.3, 0.1), vec3d(0.9, 0.9, 0.9));
        material* mats[6];
        mats[0] = new lambertian(chk);
        mats[1] = new metal(vec3d(0.8, 0.6, 0.2), 0.4);
        mats[2] = new dielectric(1.5);
        mats[3] = new emissive(vec3d(3, 2, 1));
        phong* p1 = new phong(vec3d(0.6, 0.5, 0.2)); p1->Ks = vec3d(0.5, 0.5, 0.5); p1->Ns = 20; mats[4] = p1;
        phong* p2 = new phong(vec3d(0.6, 0.5, 0.2)); p2->Tr = vec3d(0.9, 0.9, 0.9); p2->Ni = 1.45f; mats[5] = p2;
        std::vector<double> in(n * 13), out(6 * n * 12);
        for (int i = 0; i < n; i++) {
            vec3d pos(sm_range(-1, 1), sm_range(-1, 1), sm_range(-1, 1));
            vec3d nrm = vec3d(sm_range(-1, 1), sm_range(-1, 1), sm_range(-1, 1)).normalize();
            vec3d wi_d(sm_range(-1, 1), sm_range(-1, 1), sm_range(-1, 1));
            double time = sm_unit();
            double* p = &in[i * 13];
            for (int k = 0; k < 3; k++) { p[k] = pos[k]; p[3 + k] = nrm[k]; p[6 + k] = wi_d[k]; }
            p[9] = time; p[10] = sm_unit(); p[11] = sm_unit(); p[12] = 1000 + i;
            ray wi(pos - wi_d, wi_d, time);
            for (int q = 0; q < 6; q++) {
                intersection x; x.position = pos; x.uv_coord = vec2d(p[10], p[11]); x.t = 1; x.mat = mats[q];
                x.set_face_normal(wi, nrm);
                srand(1000 + i);
                ray wo; vec3d att; bool spec = false;
                vec3d em = mats[q]->emit(wi, x, x.position, x.uv_coord);
                bool sc = mats[q]->shading(wi, x, wo, att, spec);
                double pdf = sc ? mats[q]->pdf(wi, x, wo) : 0.0;
                double* w = &out[(q * n + i) * 12];
                w[0] = sc; w[1] = spec; w[2] = pdf;
                for (int k = 0; k < 3; k++) { w[3 + k] = em[k]; w[6 + k] = sc ? att[k] : 0.0; w[9 + k] = sc ? wo.direction[k] : 0.0; }
            }
        }
        dump_f64("unit_mat_in", in); dump_f64("unit_mat_out", out);
    }
    // camera::emit (camera.hpp:62-68) under a re-seeded stream
    {
        const int n = 64;
        camera cam = c0_camera(1200, 800);
        std::vector<double> out(n * 9);
        for (int i = 0; i < n; i++) {
            srand(500 + i);
            ray r = cam.emit((i % 8 + 0.5) / 8.0, (i / 8 + 0.5) / 8.0);
            double* w = &out[i * 9];
            for (int k = 0; k < 3; k++) { w[k] = r.origin[k]; w[3 + k] = r.direction[k]; }
            w[6] = r.time; w[7] = r.t_min; w[8] = r.t_max;
        }
        dump_f64("unit_cam_out", out);
        dump_camera("unit_cam_fields", cam);
    }
}

// ---------------------------------------------------------------- main
int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_harness <outdir> <mode> [args]\n"); return 1; }
    g_outdir = argv[1];
    std::string mode = argv[2];
    omp_set_num_threads(1);

    if (mode == "rand") {
        std::vector<int32_t> v;
        unsigned seeds[3] = { 1u, 12345u, 0u };
        for (unsigned s : seeds) { srand(s); for (int i = 0; i < 400; i++) v.push_back(rand()); }
        dump_i32("rand_stream", v);
    }
    else if (mode == "units") {
        do_units();
    }
    else if (mode == "c0") {            // c0 <seed> <w> <h> <spp> <depth>
        unsigned seed = atoi(argv[3]); int w = atoi(argv[4]), h = atoi(argv[5]), spp = atoi(argv[6]), depth = atoi(argv[7]);
        srand(seed);
        built_scene* b = build_c0(w, h);
        dump_f64("c0_tree", b->td.rows());
        dump_lights("c0_lights", *b->scn);
        dump_camera("c0_camera", b->scn->render_camera);
        double lo[3] = { -12, 0.02, -12 }, hi[3] = { 12, 3.0, 12 };
        do_firsthit("c0", b, 48, 1600, lo, hi);
        do_image("c0", b, w, h, spp, depth);
    }
    else if (mode == "mixed") {         // mixed <seed> <w> <h> <spp> <depth>
        unsigned seed = atoi(argv[3]); int w = atoi(argv[4]), h = atoi(argv[5]), spp = atoi(argv[6]), depth = atoi(argv[7]);
        srand(seed);
        built_scene* b = build_mixed(w, h);
        dump_f64("mixed_tree", b->td.rows());
        dump_lights("mixed_lights", *b->scn);
        dump_camera("mixed_camera", b->scn->render_camera);
        double lo[3] = { -3, -0.5, -3 }, hi[3] = { 3, 3, 3.5 };
        do_firsthit("mixed", b, 32, 1024, lo, hi);
        do_image("mixed", b, w, h, spp, depth);
    }
    else if (mode == "obj") {           // obj <path> <seed> <w> <h> <spp> <depth>
        const char* path = argv[3];
        unsigned seed = atoi(argv[4]); int w = atoi(argv[5]), h = atoi(argv[6]), spp = atoi(argv[7]), depth = atoi(argv[8]);
        srand(seed);
        built_scene* b = build_obj(path, w, h);
        dump_f64("obj_tree", b->td.rows());
        dump_lights("obj_lights", *b->scn);
        dump_camera("obj_camera", b->scn->render_camera);
        double lo[3] = { -2.5, -0.1, -2.5 }, hi[3] = { 2.5, 3, 2.5 };
        do_firsthit("obj", b, 32, 1024, lo, hi);
        do_image("obj", b, w, h, spp, depth);
    }
    else if (mode == "cornell") {       // cornell <seed> <w> <h> <spp> <depth>
        unsigned seed = atoi(argv[3]); int w = atoi(argv[4]), h = atoi(argv[5]), spp = atoi(argv[6]), depth = atoi(argv[7]);
        srand(seed);
        built_scene* b = build_cornell(w, h);
        dump_f64("cornell_tree", b->td.rows());
        dump_lights("cornell_lights", *b->scn);
        dump_camera("cornell_camera", b->scn->render_camera);
        double lo[3] = { 1, 1, 1 }, hi[3] = { 554, 554, 554 };
        do_firsthit("cornell", b, 48, 1600, lo, hi);
        do_image("cornell", b, w, h, spp, depth);
    }
    else if (mode == "smoke") {         // smoke <seed> <w> <h> <spp> <depth>
        unsigned seed = atoi(argv[3]); int w = atoi(argv[4]), h = atoi(argv[5]), spp = atoi(argv[6]), depth = atoi(argv[7]);
        srand(seed);
        built_scene* b = build_smoke(w, h);
        dump_f64("smoke_tree", b->td.rows());
        dump_lights("smoke_lights", *b->scn);
        dump_camera("smoke_camera", b->scn->render_camera);
        double lo[3] = { -100, 1, -100 }, hi[3] = { 554, 554, 554 };
        do_firsthit("smoke", b, 32, 1024, lo, hi, false);      // volume::intersect draws from rand() during these walks too
        do_image("smoke", b, w, h, spp, depth);
    }
    else if (mode == "features") {      // features <seed> <w> <h> <spp> <depth>
        unsigned seed = atoi(argv[3]); int w = atoi(argv[4]), h = atoi(argv[5]), spp = atoi(argv[6]), depth = atoi(argv[7]);
        srand(seed);
        built_scene* b = build_features(w, h);
        dump_f64("features_tree", b->td.rows());
        dump_lights("features_lights", *b->scn);
        dump_camera("features_camera", b->scn->render_camera);
        double lo[3] = { -200, 1, -200 }, hi[3] = { 554, 554, 600 };
        do_firsthit("features", b, 32, 1024, lo, hi, false);
        do_image("features", b, w, h, spp, depth);
    }
    else if (mode == "nested") {        // nested <seed> <w> <h> <spp> <depth>
        unsigned seed = atoi(argv[3]); int w = atoi(argv[4]), h = atoi(argv[5]), spp = atoi(argv[6]), depth = atoi(argv[7]);
        srand(seed);
        g_wrapper_rows = true;
        built_scene* b = build_nested(w, h);
        dump_f64("nested_tree", b->td.rows());
        dump_lights("nested_lights", *b->scn);
        dump_camera("nested_camera", b->scn->render_camera);
        double lo[3] = { -14, 0.1, -12 }, hi[3] = { 14, 8, 12 };
        do_firsthit("nested", b, 32, 1536, lo, hi, false);      // volumes draw from rand() during the walks
        do_image("nested", b, w, h, spp, depth);
    }
    else if (mode == "time_c0") {       // time_c0 <seed> <w> <h> <spp> <depth> : throughput of the real reference, 1 thread
        unsigned seed = atoi(argv[3]); int w = atoi(argv[4]), h = atoi(argv[5]), spp = atoi(argv[6]), depth = atoi(argv[7]);
        srand(seed);
        built_scene* b = build_c0(w, h);
        double t0 = omp_get_wtime();
        do_image("time", b, w, h, spp, depth);
        double t1 = omp_get_wtime();
        printf("INFO time_c0 seconds=%f\n", t1 - t0);
    }
    else if (mode == "time_obj") {      // time_obj <path> <w> <h> <spp> <depth> <cam x y z> <lookat x y z> <fov> : the real reference on an OBJ scene, 1 thread
        const char* path = argv[3];
        int w = atoi(argv[4]), h = atoi(argv[5]), spp = atoi(argv[6]), depth = atoi(argv[7]);
        vec3d pos(atof(argv[8]), atof(argv[9]), atof(argv[10])), at(atof(argv[11]), atof(argv[12]), atof(argv[13]));
        double fov = atof(argv[14]);
        srand(1);
        built_scene* b = new built_scene();
        b->scn = new scene(w, h);
        loader* ld = new loader(path);
        b->scn->add_object(ld->meshes());
        b->scn->add_light(*ld->lights());
        b->scn->set_camera(camera(pos, at, vec3d(0, 1, 0), fov, double(w) / h, 0, 1, 0, 1));
        double tb = omp_get_wtime();
        b->scn->build_bvh();
        b->root = b->scn->objects.content()[0];
        double t0 = omp_get_wtime();
        do_image("time", b, w, h, spp, depth);
        double t1 = omp_get_wtime();
        printf("INFO time_obj seconds=%f bvh_seconds=%f\n", t1 - t0, t0 - tb);
    }
    else if (mode == "images") {        // images <file>... : what the reference's image texture holds after loading each file
        for (int k = 3; k < argc; k++) {
            int w = 0, h = 0, ch = 0;
            unsigned char* px = stbi_load(argv[k], &w, &h, &ch, 0);      // texture.hpp:87
            std::vector<int32_t> dims = { w, h, ch, px ? 1 : 0 };
            std::vector<uint8_t> bytes;
            if (px) bytes.assign(px, px + (size_t) w * h * ch);
            char name[64];
            snprintf(name, sizeof name, "img%d_dims", k - 3); dump_i32(name, dims);
            snprintf(name, sizeof name, "img%d_pixels", k - 3); dump_u8(name, bytes);
            stbi_image_free(px);
        }
    }
    else { fprintf(stderr, "unknown mode %s\n", mode.c_str()); return 1; }
    return 0;
}
