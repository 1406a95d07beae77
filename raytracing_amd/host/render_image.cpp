// raytracing_amd/host/render_image.cpp -- the reference's driver (src/cpp/raytracing.cpp:73-137) on the GPU core:
// same scene-building statements, the pixel/sample loop replaced by scene::render().
//   render_image [out.png] [width height spp max_depth] [scene.obj | cornell | features | smoke | spheres] [fp64 | fp32 | fp64ref] [devices, e.g. 0,1,2,3]
// `render_image output.png 1024 1024 100 10 cornell` is the reference's own main() (raytracing.cpp:77-90,132-137).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rt_host.hpp"

int main(int argc, char* argv[])
{
    const char* path = argc > 1 ? argv[1] : "./output.png";
    int width = argc > 3 ? std::atoi(argv[2]) : 1200, height = argc > 3 ? std::atoi(argv[3]) : 800;
    int spp = argc > 4 ? std::atoi(argv[4]) : 10, max_depth = argc > 5 ? std::atoi(argv[5]) : 50;

    scene scn(width, height);
    if (argc > 6 && !std::strcmp(argv[6], "cornell")) {
        scn.add_object(cornell_box());
        scn.add_light(new flip(new planexz(213, 343, 227, 332, 554, new emissive(vec3d(30, 30, 30)))));
        scn.set_camera(camera(vec3d(278, 278, -800), vec3d(278, 278, 0), vec3d(0, 1, 0), 40, 1, 0, 1, 0, 1));
    }
    else if (argc > 6 && (!std::strcmp(argv[6], "features") || !std::strcmp(argv[6], "smoke"))) {
        // all_feature_test() (the reference's demo.hpp:94-143, which its main() never renders) / its sphere-bounded media alone.
        // The demo's rect light is not flipped and shines upwards; here it is flipped so that the picture is lit.
        const bool all = !std::strcmp(argv[6], "features");
        scn.add_object(all ? all_feature_test() : smoke_scene());
        if (all) scn.add_object(new flip(new planexz(123, 423, 147, 412, 553, new emissive(vec3d(7, 7, 7)))));
        scn.add_light(new flip(new planexz(123, 423, 147, 412, all ? 553 : 554, new emissive(vec3d(7, 7, 7)))));
        scn.set_camera(camera(vec3d(478, 278, -600), vec3d(278, 278, 0), vec3d(0, 1, 0), 40, double(width) / height, 0, 10, 0, 1));
    }
    else if (argc > 6 && std::strcmp(argv[6], "spheres")) {
        loader obj_loader(argv[6]);
        scn.add_object(obj_loader.meshes());
        scn.add_light(*obj_loader.lights());
        scn.set_camera(camera(vec3d(0, 1.6, 4.2), vec3d(0, 0.6, 0), vec3d(0, 1, 0), 40, 1.78, 0, 1, 0, 1));
    }
    else {
        object* light = new flip(new planexz(-20, 20, -20, 20, 30, new emissive(vec3d(4, 4, 4))));
        scn.add_object(random_scene());
        scn.add_object(new flip(new planexz(-20, 20, -20, 20, 30, new emissive(vec3d(4, 4, 4)))));
        scn.add_light(light);
        scn.set_camera(camera(vec3d(13, 2, 3), vec3d(0, 0, 0), vec3d(0, 1, 0), 20, 1.5, 0.1, 10, 0, 1));
    }
    scn.build_bvh();

    std::printf("[INFO] start render...\n");
    auto t0 = std::chrono::steady_clock::now();
    rt_render_stats stats;
    std::vector<unsigned char> data;
    const int mode = argc > 7 && !std::strcmp(argv[7], "fp32") ? RT_MODE_FAST_FP32 : (argc > 7 && !std::strcmp(argv[7], "fp64ref") ? RT_MODE_REFERENCE_FP64 : RT_MODE_FAST_FP64);
    std::vector<int> devices;                 // argv[8]: comma-separated HIP device indices to split the frame's samples over
    if (argc > 8) for (const char* c = argv[8]; *c;) { devices.push_back(std::atoi(c)); while (*c && *c != ',') c++; if (*c == ',') c++; }
    try { data = scn.render(spp, max_depth, mode, 0, &stats, devices); }
    catch (const std::exception& e) { std::fprintf(stderr, "[ERROR] %s\n", e.what()); return 1; }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!rt_write_png(path, width, height, 3, data.data(), true)) { std::fprintf(stderr, "[ERROR] can not write %s\n", path); return 1; }
    std::printf("[INFO] render done, using %f sec (kernel %.3f ms, %.1f Mrays/s, %llu rays).\n", sec, stats.kernel_ms,
                stats.rays / (stats.kernel_ms * 1e3), (unsigned long long) stats.rays);
    return 0;
}
