// raytracing_amd/host/ray.hpp -- ray value type (reference: src/cpp/ray.hpp:8-27)
#pragma once
#include "algebra.hpp"
#include "util.hpp"

struct ray {
    vec3d origin, direction;
    double time = 0, t_min = 1e-3, t_max = infinity;
    ray() {}
    ray(vec3d o, vec3d d, double tm = 0, double lo = 1e-3, double hi = infinity) : origin(o), direction(d), time(tm), t_min(lo), t_max(hi) {}
    vec3d at(double t) const { return origin + t * direction; }
};
