// raytracing_amd/host/transform.hpp -- translate, rotatey, flip (reference: src/cpp/transform.hpp:9-149).
// The wrappers describe geometry only; the ray transforms and the wrappers' set_face_normal calls run in the HIP core
// (csrc/rt_device.h chain_ray / fill_record / box_t).  They nest freely, around leaves, groups and bvhs alike, as in the reference:
// the flattener (bvh.hpp) turns them into chains of instance-transform records.
#pragma once

#include <cmath>

#include "object.hpp"
#include "util.hpp"

class translate : public object {
public:
    translate(object* inner, vec3d t) : obj(inner), target(t) {}
    bool bounding_box(double start, double end, aabb& bbox) const override      // transform.hpp:30-37
    {
        if (!obj->bounding_box(start, end, bbox)) return false;
        bbox = aabb(bbox.min() + target, bbox.max() + target);
        return true;
    }
    const object* wrapped(flat_wrapper& w) const override
    {
        w.kind = flat_wrapper::translate;
        for (int k = 0; k < 3; k++) w.target[k] = target[k];
        return obj;
    }
private:
    object* obj;
    vec3d target;
};

class rotatey : public object {
public:
    rotatey(object* inner, double angle) : obj(inner)                           // transform.hpp:47-79
    {
        double arc = degree_to_arc(angle);
        sin_theta = sin(arc);
        cos_theta = cos(arc);
        has_bbox = obj->bounding_box(0, 1, rotated_bbox);
        vec3d lo(infinity, infinity, infinity), hi(-infinity, -infinity, -infinity);
        for (int corner = 0; corner < 8; corner++) {                             // i, j, k of the reference's loops: bits 2, 1, 0
            const aabb& b = rotated_bbox;
            double x = (corner & 4) ? b.max().x() : b.min().x(), y = (corner & 2) ? b.max().y() : b.min().y(), z = (corner & 1) ? b.max().z() : b.min().z();
            vec3d turned(cos_theta * x + sin_theta * z, y, -sin_theta * x + cos_theta * z);
            for (int l = 0; l < 3; l++) { lo[l] = fmin(lo[l], turned[l]); hi[l] = fmax(hi[l], turned[l]); }
        }
        rotated_bbox = aabb(lo, hi);
    }
    bool bounding_box(double, double, aabb& bbox) const override { bbox = rotated_bbox; return has_bbox; }
    const object* wrapped(flat_wrapper& w) const override
    {
        w.kind = flat_wrapper::rotatey; w.sin_theta = sin_theta; w.cos_theta = cos_theta;
        return obj;
    }
private:
    object* obj;
    double sin_theta, cos_theta;
    bool has_bbox;
    aabb rotated_bbox;
};

class flip : public object {
public:
    flip(object* inner) : obj(inner) {}
    bool bounding_box(double start, double end, aabb& bbox) const override { return obj->bounding_box(start, end, bbox); }
    bool leaf_record(flat_leaf& out) const override                             // a flipped primitive in the light list
    {
        if (!obj->leaf_record(out)) return false;
        out.flags ^= RT_PRIM_FLAG_FLIP;
        return true;
    }
    const object* wrapped(flat_wrapper& w) const override { w.kind = flat_wrapper::flip; return obj; }
private:
    object* obj;
};
