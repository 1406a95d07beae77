// raytracing_amd/host/camera.hpp -- look-at thin-lens camera (reference: src/cpp/camera.hpp:9-78).
// setup_camera() stays on the host (so tan() is glibc's, as in the reference); emit() runs on the GPU
// (csrc/rt_device.h camera_emit) from the derived fields exported by fields().
#pragma once

#include <cmath>

#include "algebra.hpp"
#include "util.hpp"

class camera {
public:
    vec3d position, lookat, up;
    double fov = 0, aspect_ratio = 0, aperture = 0, focus_distance = 0;

    camera() {}
    camera(vec3d position_, vec3d lookat_, vec3d up_, double fov_, double aspect_ratio_, double aperture_ = 0, double focus_distance_ = 1,
           double shutter_start_ = 0, double shutter_end_ = 0)
        : position(position_), lookat(lookat_), up(up_), fov(fov_), aspect_ratio(aspect_ratio_), aperture(aperture_), focus_distance(focus_distance_),
          shutter_start(shutter_start_), shutter_end(shutter_end_)
    {
        setup_camera();
    }

    void setup_camera()
    {
        double sensor_h = std::tan(degree_to_arc(fov) / 2) * 2;
        double sensor_w = sensor_h * aspect_ratio;
        w = (position - lookat).normalize();
        u = up.cross(w).normalize();
        v = w.cross(u);
        horizontal = focus_distance * sensor_w * u;
        vertical = focus_distance * sensor_h * v;
        left_bottom = position - horizontal / 2.0 - vertical / 2.0 - focus_distance * w;
        lens_radius = aperture / 2;
    }

    // layout of rt_scene_desc.camera (include/rtcore.h)
    void fields(double out[24]) const
    {
        const vec3d* vs[7] = { &position, &left_bottom, &horizontal, &vertical, &u, &v, &w };
        for (int i = 0; i < 7; i++) for (int k = 0; k < 3; k++) out[3 * i + k] = (*vs[i])[k];
        out[21] = lens_radius; out[22] = shutter_start; out[23] = shutter_end;
    }

private:
    vec3d left_bottom, horizontal, vertical, u, v, w;
    double lens_radius = 0, shutter_start = 0, shutter_end = 0;
};
