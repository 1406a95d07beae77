// raytracing_amd/host/aabb.hpp -- axis-aligned box used by the host-side BVH builder
// (reference: src/cpp/aabb.hpp:8-61).  The slab test itself runs on the GPU (csrc/rt_device.h box_hit).
#pragma once
#include <cmath>
#include "algebra.hpp"

class aabb {
public:
    aabb() {}
    aabb(const vec3d& lo, const vec3d& hi) : minimum(lo), maximum(hi) {}
    vec3d min() const { return minimum; }
    vec3d max() const { return maximum; }
    friend aabb surrounding_box(const aabb& a, const aabb& b)
    {
        vec3d lo(std::fmin(a.minimum.x(), b.minimum.x()), std::fmin(a.minimum.y(), b.minimum.y()), std::fmin(a.minimum.z(), b.minimum.z()));
        vec3d hi(std::fmax(a.maximum.x(), b.maximum.x()), std::fmax(a.maximum.y(), b.maximum.y()), std::fmax(a.maximum.z(), b.maximum.z()));
        return aabb(lo, hi);
    }
private:
    vec3d minimum, maximum;
};
