// raytracing_amd/host/transform.hpp -- translate, rotatey, flip (reference: src/cpp/transform.hpp:9-149).
// The wrappers describe geometry only; the ray transform and the wrappers' set_face_normal calls run in the HIP core
// (csrc/rt_device.h box_t / xform_rays).  Around a `box` -- what cornell_box() builds (demo.hpp:80-88) -- the wrappers
// travel in the box's own record; around any other leaf or around a bvh -- all_feature_test()'s sphere cloud
// (demo.hpp:136-140) -- they become an instance transform of the flattened scene.  Supported nesting, once:
// translate(rotatey(x)), rotatey(x), translate(x).
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
    bool leaf_record(flat_leaf& out) const override
    {
        if (!obj->leaf_record(out)) return false;
        if (out.kind != RT_PRIM_BOX) { peel_transform(out.xform); return true; }
        if (out.flags & RT_PRIM_FLAG_TRANSLATE) throw std::runtime_error("flatten: translate(translate(.)) is not supported");
        out.flags |= RT_PRIM_FLAG_TRANSLATE;
        for (int k = 0; k < 3; k++) out.data[8 + k] = target[k];
        return true;
    }
    const object* peel_transform(flat_xform& x) const override
    {
        if (x.flags) throw std::runtime_error("flatten: translate is supported as the outermost wrapper only");
        x.flags |= RT_PRIM_FLAG_TRANSLATE;
        for (int k = 0; k < 3; k++) x.target[k] = target[k];
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
        vec3d minimum(infinity, infinity, infinity), maximum(-infinity, -infinity, -infinity);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    double x = i * rotated_bbox.max().x() + (1 - i) * rotated_bbox.min().x();
                    double y = j * rotated_bbox.max().y() + (1 - j) * rotated_bbox.min().y();
                    double z = k * rotated_bbox.max().z() + (1 - k) * rotated_bbox.min().z();
                    double new_x = cos_theta * x + sin_theta * z;
                    double new_z = -sin_theta * x + cos_theta * z;
                    vec3d temp(new_x, y, new_z);
                    for (int l = 0; l < 3; l++) {
                        minimum[l] = fmin(minimum[l], temp[l]);
                        maximum[l] = fmax(maximum[l], temp[l]);
                    }
                }
        rotated_bbox = aabb(minimum, maximum);
    }
    bool bounding_box(double, double, aabb& bbox) const override { bbox = rotated_bbox; return has_bbox; }
    bool leaf_record(flat_leaf& out) const override
    {
        if (!obj->leaf_record(out)) return false;
        if (out.kind != RT_PRIM_BOX) {
            if (out.xform.flags) throw std::runtime_error("flatten: rotatey is supported directly around the wrapped object only");
            out.xform.flags |= RT_PRIM_FLAG_ROTATEY; out.xform.sin_theta = sin_theta; out.xform.cos_theta = cos_theta;
            return true;
        }
        if (out.flags & (RT_PRIM_FLAG_ROTATEY | RT_PRIM_FLAG_TRANSLATE)) throw std::runtime_error("flatten: rotatey is supported directly around a box only");
        out.flags |= RT_PRIM_FLAG_ROTATEY;
        out.data[6] = sin_theta; out.data[7] = cos_theta;
        return true;
    }
    const object* peel_transform(flat_xform& x) const override
    {
        if (x.flags & RT_PRIM_FLAG_ROTATEY) throw std::runtime_error("flatten: rotatey(rotatey(.)) is not supported");
        x.flags |= RT_PRIM_FLAG_ROTATEY; x.sin_theta = sin_theta; x.cos_theta = cos_theta;
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
    bool leaf_record(flat_leaf& out) const override
    {
        if (!obj->leaf_record(out)) return false;
        out.flags ^= RT_PRIM_FLAG_FLIP;
        return true;
    }
private:
    object* obj;
};
