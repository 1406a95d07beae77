// raytracing_amd/host/transform.hpp -- flip (reference: src/cpp/transform.hpp:124-149).
// translate / rotatey are the next tier (Cornell box), see DESIGN.md.
#pragma once

#include "object.hpp"

class flip : public object {
public:
    flip(object* inner) : obj(inner) {}
    bool bounding_box(double start, double end, aabb& bbox) const override { return obj->bounding_box(start, end, bbox); }
    bool leaf_record(flat_leaf& out) const override
    {
        if (!obj->leaf_record(out)) return false;
        out.flags ^= RT_PRIM_FLAG_FLIP;
        return true;
    }
private:
    object* obj;
};
