// raytracing_amd/host/object.hpp -- abstract scene object (reference: src/cpp/object.hpp:10-18).
// Host objects describe geometry; `intersect` is not a host virtual any more -- the hot path runs in HIP.
#pragma once

#include "aabb.hpp"
#include "flatten.hpp"

class object {
public:
    uint64_t serial = rt_next_serial();
    virtual ~object() {}
    virtual bool bounding_box(double start, double end, aabb& bbox) const = 0;
    // a leaf writes its record and returns true; containers (bvh, group) return false
    virtual bool leaf_record(flat_leaf& out) const { (void) out; return false; }
    // flip / translate / rotatey: describe this wrapper in w and return the wrapped object; everything else returns null
    virtual const object* wrapped(flat_wrapper& w) const { (void) w; return nullptr; }
};
