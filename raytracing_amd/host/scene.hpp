// raytracing_amd/host/scene.hpp -- scene container (reference: src/cpp/scene.hpp:12-63), plus the boundary this
// project inserts under it: flatten() -> rt_scene_desc -> rtcore C ABI -> HIP.
//
//   scene scn(w, h); scn.add_object(...); scn.add_light(...); scn.set_camera(cam); scn.build_bvh();   // as before
//   std::vector<unsigned char> rgb = scn.render(spp, max_depth);                                       // new: GPU
#pragma once

#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "bvh.hpp"
#include "camera.hpp"
#include "group.hpp"
#include "rtcore.h"

class scene {
public:
    group objects;
    group* lights = new group();
    camera render_camera;

    scene() {}
    scene(int w, int h) { set_size(w, h); }

    void set_size(int w, int h) { width = w; height = h; fix_aspect(); }
    vec2i get_size() { return vec2i(width, height); }
    void set_camera(const camera& cam) { render_camera = cam; fix_aspect(); }
    void add_object(const group& grp) { objects.add(grp); }
    void add_object(object* obj) { objects.add(obj); }
    void add_light(object* lgt) { lights->add(lgt); }
    void add_light(const group& lgt) { lights->add(lgt); }

    void build_bvh()
    {
        std::printf("[INFO] building bvh acceleration architecture...\n");
        object* root = new bvh(objects);
        objects = group(root);
        std::printf("[INFO] bvh done.\n");
    }

    // object graph -> flat arrays.  `out` must outlive any rt_scene_desc taken from it.
    void flatten(flat_scene& out) const
    {
        flattener f;
        for (int i = 0; i < objects.size(); i++) f.collect(objects.content()[i]);
        f.assign_prim_ids();
        for (int i = 0; i < objects.size(); i++) f.out.roots.push_back(f.emit(objects.content()[i]));
        for (int i = 0; i < lights->size(); i++) {
            flat_leaf leaf;
            if (!lights->content()[i]->leaf_record(leaf)) throw std::runtime_error("flatten: a light must be a primitive (optionally flipped)");
            f.store_light(leaf);
        }
        render_camera.fields(f.out.camera);
        out = std::move(f.out);
    }

    // Renders the frame: the replacement for the pixel/sample loop of render_image (reference raytracing.cpp:96-123).  Returns
    // width*height*3 bytes, row 0 = bottom row, ready for rt_write_png(..., flip_vertically = true).
    // mode: RT_MODE_FAST_FP64 (default: the reference's fp64 arithmetic behind the library's tree), RT_MODE_FAST_FP32,
    // RT_MODE_REFERENCE_FP64 (reference tree and visit order), RT_MODE_FP32_REFERENCE_TREE.
    // devices: HIP device indices to spread the frame over (empty: the current device).  The scene is replicated, every device renders
    // its share of the global sample indices (streams are keyed by the global index: the frame does not depend on the split), the
    // accumulation buffers are summed onto the first device by rt_frame_reduce (peer copies over xGMI), which resolves.
    std::vector<unsigned char> render(int spp, int max_depth, int mode = RT_MODE_FAST_FP64, unsigned long long seed = 0, rt_render_stats* stats_out = nullptr,
                                      std::vector<int> devices = {}) const
    {
        const bool fp32_sums = mode == RT_MODE_FAST_FP32 || mode == RT_MODE_FP32_REFERENCE_TREE;
        flat_scene flat;
        flatten(flat);
        rt_scene_desc d = flat.desc();
        const bool current_only = devices.empty();
        if (current_only) devices.push_back(-1);
        const int n_dev = (int) devices.size();
        const size_t n = (size_t) width * height * 3;
        std::vector<rt_scene*> dev(n_dev, nullptr);
        std::vector<void*> accum(n_dev, nullptr);
        void* rgb = nullptr;
        std::vector<unsigned char> image(n);
        auto release = [&]() {
            for (int i = 0; i < n_dev; i++) {
                if (!current_only) rt_device_set(devices[i]);
                rt_buffer_free(accum[i]); rt_scene_destroy(dev[i]);
            }
            if (!current_only) rt_device_set(devices[0]);
            rt_buffer_free(rgb);
        };
        try {
            for (int i = 0; i < n_dev; i++) {
                if (!current_only) check(rt_device_set(devices[i]));
                check(rt_scene_create(&d, &dev[i]));
                check(rt_buffer_alloc(n * (fp32_sums ? sizeof(float) : sizeof(double)), &accum[i]));
            }
            rt_render_params p;
            p.width = width; p.height = height; p.y_begin = 0; p.y_end = height; p.max_depth = max_depth;
            p.mode = mode; p.flags = 0; p.seed = seed; p.stream = nullptr;
            for (int i = 0; i < n_dev; i++) {       // asynchronous launches: the devices render side by side
                if (!current_only) check(rt_device_set(devices[i]));
                p.spp_begin = (int) ((long long) spp * i / n_dev); p.spp_end = (int) ((long long) spp * (i + 1) / n_dev);
                check(rt_render(dev[i], &p, accum[i], nullptr));
            }
            rt_render_stats total{};
            for (int i = 0; i < n_dev; i++) {
                rt_render_stats st{};
                check(rt_render_wait(dev[i], &st));
                total.rays += st.rays; total.rejected_samples += st.rejected_samples; total.untraced_rays += st.untraced_rays;
                if (st.kernel_ms > total.kernel_ms) total.kernel_ms = st.kernel_ms;
            }
            if (stats_out) *stats_out = total;
            if (n_dev > 1) check(rt_frame_reduce(accum[0], devices[0], &accum[1], &devices[1], n_dev - 1, fp32_sums ? 0 : 1, n));
            if (!current_only) check(rt_device_set(devices[0]));
            check(rt_buffer_alloc(n, &rgb));
            check(rt_resolve(accum[0], fp32_sums ? 0 : 1, (size_t) width * height, spp, static_cast<uint8_t*>(rgb), nullptr));
            check(rt_buffer_download(image.data(), rgb, n));
        }
        catch (...) {
            release();
            throw;
        }
        release();
        return image;
    }

private:
    int width = 1280;
    int height = 720;
    static void check(int rc) { if (rc != RT_OK) throw std::runtime_error(std::string("rtcore: ") + rt_last_error()); }
    void fix_aspect()   // the scene's size overrides the camera's aspect ratio (reference scene.hpp:57-62)
    {
        render_camera.aspect_ratio = double(width) / height;
        render_camera.setup_camera();
    }
};
