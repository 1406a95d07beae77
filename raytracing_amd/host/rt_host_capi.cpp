// raytracing_amd/host/rt_host_capi.cpp -> librthost.so
// C entry points over the C++ host mirror, for callers that are not C++ (bench.py, tests): build one of the
// BASELINE workload scenes with the host classes, flatten it, hand back an rt_scene_desc whose arrays stay owned
// by the returned handle.  No rendering here: the desc goes to librtcore.so (include/rtcore.h).
#include <cstdlib>
#include <cstring>
#include <string>

#include "rt_host.hpp"

struct rth_scene {
    flat_scene flat;
    rt_scene_desc desc;
    int width, height;
};

static thread_local std::string g_err;

static rth_scene* finish(scene& scn, int w, int h)
{
    rth_scene* out = new rth_scene();
    scn.flatten(out->flat);
    out->desc = out->flat.desc();
    out->width = w; out->height = h;
    return out;
}

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

// BASELINE configs 0/1: random_scene() + one overhead area light (the reference has no sky and needs >= 1 light,
// SURVEY 9-2): flip(planexz(-20,20,-20,20,30, emissive(4,4,4))) as an object and, as a second instance, as the light
// (the way raytracing.cpp:81-85 does for the Cornell box).  srand(seed) first: scene and tree come from rand().
rth_scene* rth_build_random_spheres(int width, int height, unsigned seed)
{
    try {
        srand(seed);
        scene scn(width, height);
        scn.add_object(random_scene());
        scn.add_object(new flip(new planexz(-20, 20, -20, 20, 30, new emissive(vec3d(4, 4, 4)))));
        scn.add_light(new flip(new planexz(-20, 20, -20, 20, 30, new emissive(vec3d(4, 4, 4)))));
        scn.set_camera(camera(vec3d(13, 2, 3), vec3d(0, 0, 0), vec3d(0, 1, 0), 20, double(width) / height, 0.1, 10, 0, 1));
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// OBJ scene through the loader: meshes -> objects, emissive faces -> lights, camera given by the caller
rth_scene* rth_build_obj(const char* path, int width, int height, unsigned seed, const double cam_pos[3], const double cam_lookat[3], double fov)
{
    try {
        srand(seed);
        scene scn(width, height);
        loader ld(path);
        if (ld.meshes().size() == 0) { g_err = std::string("no triangles loaded from ") + path; return nullptr; }
        scn.add_object(ld.meshes());
        scn.add_light(*ld.lights());
        scn.set_camera(camera(vec3d(cam_pos[0], cam_pos[1], cam_pos[2]), vec3d(cam_lookat[0], cam_lookat[1], cam_lookat[2]), vec3d(0, 1, 0), fov,
                              double(width) / height, 0, 1, 0, 1));
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// the hand-made scene of the `mixed` fixture (every primitive and material kind in one tree)
rth_scene* rth_build_mixed(int width, int height, unsigned seed)
{
    try {
        srand(seed);
        scene scn(width, height);
        group grp;
        grp.add(new sphere(vec3d(0, -100.5, -1), 100, new lambertian(new checker(vec3d(0.2, 0.3, 0.1), vec3d(0.9, 0.9, 0.9)))));
        grp.add(new sphere(vec3d(0, 0, -1), 0.5, new lambertian(vec3d(0.1, 0.2, 0.5))));
        grp.add(new sphere(vec3d(-1.05, 0, -1), 0.5, new dielectric(1.5)));
        grp.add(new sphere(vec3d(1.05, 0, -1), 0.5, new metal(vec3d(0.8, 0.6, 0.2), 0.3)));
        grp.add(new msphere(vec3d(0.3, 0.7, -0.6), vec3d(0.3, 0.9, -0.6), 0, 1, 0.15, new lambertian(vec3d(0.7, 0.2, 0.2))));
        grp.add(new sphere(vec3d(-0.4, -0.3, -0.2), 0.2, new metal(vec3d(0.9, 0.9, 0.9), 0.0)));
        grp.add(new triangle(vec3d(-2, -0.5, -2.5), vec3d(2, -0.5, -2.5), vec3d(0, 2.0, -2.8), new lambertian(vec3d(0.3, 0.7, 0.3))));
        grp.add(new triangle(vec3d(1.2, -0.5, 0.2), vec3d(1.9, -0.5, -0.4), vec3d(1.5, 0.6, -0.1), new metal(vec3d(0.7, 0.7, 0.9), 0.1)));
        grp.add(new triangle(vec3d(-1.9, -0.5, 0.3), vec3d(-1.2, -0.5, 0.1), vec3d(-1.6, 0.5, 0.0), new phong(vec3d(0.6, 0.5, 0.2))));
        grp.add(new planexy(-3, 3, -0.5, 3, -3.2, new lambertian(vec3d(0.6, 0.6, 0.7))));
        grp.add(new planeyz(-0.5, 3, -3.2, 1, -3, new lambertian(vec3d(0.7, 0.3, 0.3))));
        grp.add(new flip(new planexz(-1, 1, -2, 0, 3, new emissive(vec3d(6, 6, 6)))));
        phong* lamp = new phong(vec3d(0, 0, 0));
        lamp->Ke = vec3d(5, 4, 3);
        triangle* lamp_tri = new triangle(vec3d(2.5, 2.5, -1.5), vec3d(2.5, 0.5, -0.5), vec3d(2.5, 0.5, -2.5), lamp);
        grp.add(lamp_tri);
        scn.add_object(grp);
        scn.add_light(new flip(new planexz(-1, 1, -2, 0, 3, new emissive(vec3d(6, 6, 6)))));
        scn.add_light(lamp_tri);
        scn.set_camera(camera(vec3d(-0.5, 1.2, 3.5), vec3d(0, 0.2, -1), vec3d(0, 1, 0), 45, double(width) / height, 0.05, 4.5, 0, 1));
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// the reference's own main() scene: raytracing.cpp:77-88 (Cornell box, camera at (278,278,-800), one flipped rect light)
rth_scene* rth_build_cornell(int width, int height, unsigned seed)
{
    try {
        srand(seed);
        camera cam = camera(vec3d(278, 278, -800), vec3d(278, 278, 0), vec3d(0, 1, 0), 40, 1, 0, 1, 0, 1);
        object* light = new flip(new planexz(213, 343, 227, 332, 554, new emissive(vec3d(30, 30, 30))));
        scene scn(width, height);
        scn.add_object(cornell_box());
        scn.add_light(light);
        scn.set_camera(cam);
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// The Cornell room with instances in place of its blocks and no medium anywhere: a turned and moved cloud of spheres under a tree of its
// own (the wrapping all_feature_test() gives its cloud), a metal sphere that is only moved and a glass sphere that is only turned
// (transforms around single leaves).  The cloud is sparse and diffuse on purpose: 200 balls of radius 12, a third of them fuzzy metal, gave 4 % of
// the pixels another ray count in fp32 than in fp64 -- as many as the same balls placed in world space without any instance: grazes among
// small spheres, not a defect; 40 diffuse balls of radius 25 give 1.3 %.
// Not in the reference: a test scene for the instanced kernel variants, whose other scene --
// all_feature_test() -- carries media, and a medium's draws follow the visit order, so its per-pixel ray counts cannot be compared
// between trees.
rth_scene* rth_build_instanced(int width, int height, unsigned seed)
{
    try {
        const unsigned omit = seed >> 8;        // test hook: bit 0 leaves out the cloud, bit 1 the moved sphere, bit 2 the turned sphere
        srand(seed & 0xffu);
        camera cam = camera(vec3d(278, 278, -800), vec3d(278, 278, 0), vec3d(0, 1, 0), 40, 1, 0, 1, 0, 1);
        scene scn(width, height);
        material* white = new lambertian(vec3d(0.75, 0.75, 0.75));
        scn.add_object(new planeyz(0, 555, 0, 555, 555, new lambertian(vec3d(0.12, 0.45, 0.15))));
        scn.add_object(new planeyz(0, 555, 0, 555, 0, new lambertian(vec3d(0.65, 0.05, 0.05))));
        scn.add_object(new planexz(0, 555, 0, 555, 0, white));
        scn.add_object(new planexz(0, 555, 0, 555, 555, white));
        scn.add_object(new planexy(0, 555, 0, 555, 555, white));
        scn.add_object(new flip(new planexz(213, 343, 227, 332, 554, new emissive(vec3d(30, 30, 30)))));
        group cloud;
        for (int k = 0; k < 40; k++) cloud.add(new sphere(random_vector(0, 165), 25, white));
        if (!(omit & 1u)) scn.add_object(new translate(new rotatey(new bvh(cloud), 15), vec3d(265, 0, 295)));
        if (!(omit & 2u)) scn.add_object(new translate(new sphere(vec3d(0, 0, 0), 60, new metal(vec3d(0.9, 0.7, 0.5), 0)), vec3d(150, 60, 120)));
        if (!(omit & 4u)) scn.add_object(new rotatey(new sphere(vec3d(120, 250, 150), 50, new dielectric(1.5)), 30));
        scn.add_light(new flip(new planexz(213, 343, 227, 332, 554, new emissive(vec3d(30, 30, 30)))));
        scn.set_camera(cam);
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// all_feature_test() (demo.hpp) under the light and camera of oracle/ref_harness.cpp build_features
rth_scene* rth_build_features(int width, int height, unsigned seed)
{
    try {
        srand(seed);
        scene scn(width, height);
        scn.add_object(all_feature_test());
        scn.add_light(new planexz(123, 423, 147, 412, 554, new emissive(vec3d(7, 7, 7))));
        scn.set_camera(camera(vec3d(478, 278, -600), vec3d(278, 278, 0), vec3d(0, 1, 0), 40, double(width) / height, 0, 10, 0, 1));
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// all_feature_test() LIT: the demo's own rect light is not flipped and shines upwards, so the scene above renders black; here the light
// is a flipped rect just under it (as an object and as the light), the way render_image's `features` scene is built
rth_scene* rth_build_features_lit(int width, int height, unsigned seed)
{
    try {
        srand(seed);
        scene scn(width, height);
        scn.add_object(all_feature_test());
        scn.add_object(new flip(new planexz(123, 423, 147, 412, 553, new emissive(vec3d(7, 7, 7)))));
        scn.add_light(new flip(new planexz(123, 423, 147, 412, 553, new emissive(vec3d(7, 7, 7)))));
        scn.set_camera(camera(vec3d(478, 278, -600), vec3d(278, 278, 0), vec3d(0, 1, 0), 40, double(width) / height, 0, 10, 0, 1));
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// nested_test() (demo.hpp) under the light and camera of oracle/ref_harness.cpp build_nested
rth_scene* rth_build_nested(int width, int height, unsigned seed)
{
    try {
        srand(seed);
        scene scn(width, height);
        scn.add_object(nested_test());
        scn.add_light(new flip(new planexz(-4, 4, -4, 4, 14, new emissive(vec3d(8, 8, 8)))));
        scn.set_camera(camera(vec3d(2, 9, 26), vec3d(0, 1, 0), vec3d(0, 1, 0), 40, double(width) / height, 0.1, 25, 0, 1));
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// volume / fog: smoke_scene() (demo.hpp) under its flipped rect light, camera of the classic final-scene view
rth_scene* rth_build_smoke(int width, int height, unsigned seed)
{
    try {
        srand(seed);
        scene scn(width, height);
        scn.add_object(smoke_scene());
        scn.add_light(new flip(new planexz(123, 423, 147, 412, 554, new emissive(vec3d(7, 7, 7)))));
        scn.set_camera(camera(vec3d(478, 278, -600), vec3d(278, 278, 0), vec3d(0, 1, 0), 40, double(width) / height, 0, 10, 0, 1));
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// BASELINE config 4: deep_bounce_scene() (demo.hpp) under the same light and camera as the random-spheres configs
rth_scene* rth_build_deep_bounce(int width, int height, unsigned seed)
{
    try {
        srand(seed);
        scene scn(width, height);
        scn.add_object(deep_bounce_scene());
        scn.add_object(new flip(new planexz(-20, 20, -20, 20, 30, new emissive(vec3d(4, 4, 4)))));
        scn.add_light(new flip(new planexz(-20, 20, -20, 20, 30, new emissive(vec3d(4, 4, 4)))));
        scn.set_camera(camera(vec3d(13, 2, 3), vec3d(0, 0, 0), vec3d(0, 1, 0), 20, double(width) / height, 0.1, 10, 0, 1));
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// image textures (texture.hpp:78-133) on a sphere (uv from the normal), on a lambertian triangle and as a phong map_Kd
// (color_based = true: pow(., 2.2) per fetch), over a checker floor; texels are procedural
rth_scene* rth_build_textured(int width, int height, unsigned seed)
{
    try {
        srand(seed);
        std::vector<unsigned char> px(64 * 32 * 3), px4(16 * 16 * 4);
        for (int y = 0; y < 32; y++) for (int x = 0; x < 64; x++) {
            unsigned char* p = &px[(y * 64 + x) * 3];
            p[0] = (unsigned char) (x * 4); p[1] = (unsigned char) (y * 8); p[2] = (unsigned char) (((x / 8 + y / 8) & 1) ? 220 : 40);
        }
        for (int i = 0; i < 16 * 16; i++) { px4[i * 4] = (unsigned char) (i); px4[i * 4 + 1] = (unsigned char) (255 - i); px4[i * 4 + 2] = (unsigned char) ((i * 7) & 255); px4[i * 4 + 3] = 255; }
        scene scn(width, height);
        group grp;
        grp.add(new sphere(vec3d(0, -100.5, -1), 100, new lambertian(new checker(vec3d(0.2, 0.3, 0.1), vec3d(0.9, 0.9, 0.9)))));
        grp.add(new sphere(vec3d(-0.6, 0.1, -1), 0.6, new lambertian(new image(px.data(), 64, 32, 3, false))));
        grp.add(new msphere(vec3d(0.9, 0.0, -0.6), vec3d(0.9, 0.2, -0.6), 0, 1, 0.35, new metal(new image(px4.data(), 16, 16, 4, true), 0.2)));
        vertex v[3];
        v[0].position = vec3d(-1.8, -0.5, -2.2); v[1].position = vec3d(1.8, -0.5, -2.2); v[2].position = vec3d(0, 1.9, -2.4);
        v[0].uv_coord = vec2d(-0.2, 0.0); v[1].uv_coord = vec2d(1.2, 0.0); v[2].uv_coord = vec2d(0.5, 1.1);     // outside [0,1]: exercises the clamp
        for (int k = 0; k < 3; k++) v[k].normal = vec3d(0, 0.1, 1);
        phong* mapped = new phong(new image(px.data(), 64, 32, 3, true));
        grp.add(new triangle(v, mapped));
        grp.add(new flip(new planexz(-1.5, 1.5, -2.5, 0.5, 3, new emissive(vec3d(5, 5, 5)))));
        scn.add_object(grp);
        scn.add_light(new flip(new planexz(-1.5, 1.5, -2.5, 0.5, 3, new emissive(vec3d(5, 5, 5)))));
        scn.set_camera(camera(vec3d(0, 0.8, 3.2), vec3d(0, 0.3, -1), vec3d(0, 1, 0), 42, double(width) / height, 0.02, 4.0, 0, 1));
        scn.build_bvh();
        return finish(scn, width, height);
    }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// What an image texture holds after loading `path` (texture.hpp:87: stbi_load with 0 requested channels): size and channel
// count as in the file, texels top row first.  Returns the byte count w*h*ch and copies the texels when `capacity` holds
// them (call once with capacity 0 for the size); -1 with rth_last_error set when the file cannot be decoded.
long long rth_load_image(const char* path, int* w, int* h, int* channels, unsigned char* texels, unsigned long long capacity)
{
    try {
        std::vector<unsigned char> px;
        int iw = 0, ih = 0, ich = 0;
        if (!path || !rt_load_image(path, px, iw, ih, ich)) { g_err = std::string("can not load image: ") + (path ? path : "(null)"); return -1; }
        if (w) *w = iw;
        if (h) *h = ih;
        if (channels) *channels = ich;
        if (texels && capacity >= px.size()) std::memcpy(texels, px.data(), px.size());
        return (long long) px.size();
    }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}

const rt_scene_desc* rth_desc(const rth_scene* s) { return s ? &s->desc : nullptr; }
void rth_free(rth_scene* s) { delete s; }

}  // extern "C"
