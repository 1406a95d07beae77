// raytracing_amd/host/primitive.hpp -- sphere, msphere, planexy/xz/yz
// (reference: src/cpp/primitive.hpp:13-458), `box`, and `volume` around a sphere boundary.
#pragma once

#include <stdexcept>

#include "material.hpp"
#include "object.hpp"

class sphere : public object {
public:
    vec3d center;
    double radius;
    material* mat;
    sphere() : radius(0), mat(nullptr) {}
    sphere(vec3d c, double r, material* m) : center(c), radius(r), mat(m) {}
    bool bounding_box(double, double, aabb& bbox) const override
    {
        vec3d r(radius, radius, radius);
        bbox = aabb(center - r, center + r);
        return true;
    }
    bool leaf_record(flat_leaf& out) const override
    {
        out.kind = RT_PRIM_SPHERE; out.mat = mat;
        for (int k = 0; k < 3; k++) out.data[k] = center[k];
        out.data[3] = radius;
        return true;
    }
};

class msphere : public object {
public:
    vec3d center0, center1;
    double start, end;
    double radius;
    material* mat;
    msphere() : start(0), end(0), radius(0), mat(nullptr) {}
    msphere(vec3d c0, vec3d c1, double t0, double t1, double r, material* m) : center0(c0), center1(c1), start(t0), end(t1), radius(r), mat(m) {}
    vec3d center(double time) const { return center0 + ((time - start) / (end - start)) * (center1 - center0); }
    bool bounding_box(double t0, double t1, aabb& bbox) const override
    {
        vec3d r(radius, radius, radius);
        aabb a(center(t0) - r, center(t0) + r), b(center(t1) - r, center(t1) + r);
        bbox = surrounding_box(a, b);
        return true;
    }
    bool leaf_record(flat_leaf& out) const override
    {
        out.kind = RT_PRIM_MSPHERE; out.mat = mat;
        for (int k = 0; k < 3; k++) { out.data[k] = center0[k]; out.data[3 + k] = center1[k]; }
        out.data[6] = start; out.data[7] = end; out.data[8] = radius;
        return true;
    }
};

class box;
// constant-density medium inside `surface` (reference: src/cpp/primitive.hpp:329-396).  The core takes a sphere or a box as the boundary,
// the box optionally inside rotatey and then translate (volume(translate(rotatey(box))), the smoke blocks of "the next week"); the
// boundary's own material is never used by the reference either.  leaf_record is defined below `box`.
class volume : public object {
public:
    volume(object* surface_, double density, texture* tex) : surface(surface_), mat(new fog(tex)), neg_inv_d(-1.0 / density) {}
    volume(object* surface_, double density, const vec3d& color) : surface(surface_), mat(new fog(color)), neg_inv_d(-1.0 / density) {}
    bool bounding_box(double start, double end, aabb& bbox) const override { return surface->bounding_box(start, end, bbox); }
    bool leaf_record(flat_leaf& out) const override;
private:
    object* surface;
    material* mat;
    double neg_inv_d;
};

// shared body of the three axis-aligned rectangles; `axis` is the constant coordinate
template <int kKind>
class plane_base : public object {
public:
    material* mat;
protected:
    plane_base(material* m) : mat(m) {}
    bool record(flat_leaf& out, double a0, double a1, double b0, double b1, double k) const
    {
        out.kind = kKind; out.mat = mat;
        out.data[0] = a0; out.data[1] = a1; out.data[2] = b0; out.data[3] = b1; out.data[4] = k;
        return true;
    }
};

class planexy : public plane_base<RT_PRIM_PLANEXY> {
public:
    double x0, x1, y0, y1, k;
    planexy(double x0_, double x1_, double y0_, double y1_, double k_, material* m) : plane_base(m), x0(x0_), x1(x1_), y0(y0_), y1(y1_), k(k_) {}
    bool bounding_box(double, double, aabb& bbox) const override { bbox = aabb(vec3d(x0, y0, k - 0.001), vec3d(x1, y1, k + 0.001)); return true; }
    bool leaf_record(flat_leaf& out) const override { return record(out, x0, x1, y0, y1, k); }
};

class planexz : public plane_base<RT_PRIM_PLANEXZ> {
public:
    double x0, x1, z0, z1, k;
    planexz(double x0_, double x1_, double z0_, double z1_, double k_, material* m) : plane_base(m), x0(x0_), x1(x1_), z0(z0_), z1(z1_), k(k_) {}
    bool bounding_box(double, double, aabb& bbox) const override { bbox = aabb(vec3d(x0, k - 0.001, z0), vec3d(x1, k + 0.001, z1)); return true; }
    bool leaf_record(flat_leaf& out) const override { return record(out, x0, x1, z0, z1, k); }
};

class planeyz : public plane_base<RT_PRIM_PLANEYZ> {
public:
    double y0, y1, z0, z1, k;
    planeyz(double y0_, double y1_, double z0_, double z1_, double k_, material* m) : plane_base(m), y0(y0_), y1(y1_), z0(z0_), z1(z1_), k(k_) {}
    bool bounding_box(double, double, aabb& bbox) const override { bbox = aabb(vec3d(k - 0.001, y0, z0), vec3d(k + 0.001, y1, z1)); return true; }
    bool leaf_record(flat_leaf& out) const override { return record(out, y0, y1, z0, z1, k); }
};

// box: six rects tested as one object (reference: src/cpp/primitive.hpp:399-458).  One leaf of the bvh; the HIP core walks the
// sides in the constructor's order (xy at max z, min z; xz at max y, min y; yz at max x, min x).
class box : public object {
public:
    vec3d minimum, maximum;
    material* mat;
    box(const vec3d& mn, const vec3d& mx, material* m) : minimum(mn), maximum(mx), mat(m) {}
    bool bounding_box(double, double, aabb& bbox) const override { bbox = aabb(minimum, maximum); return true; }
    bool leaf_record(flat_leaf& out) const override
    {
        out.kind = RT_PRIM_BOX; out.mat = mat;
        for (int k = 0; k < 3; k++) { out.data[k] = minimum[k]; out.data[3 + k] = maximum[k]; }
        out.data[6] = 0.0; out.data[7] = 1.0;
        return true;
    }
};


inline bool volume::leaf_record(flat_leaf& out) const
{
    out.kind = RT_PRIM_VOLUME; out.mat = mat;
    // the boundary's wrappers, outermost first: flips do not matter to a medium, translate(rotatey(.)) goes into the record
    const object* s = surface;
    int fl = 0;
    flat_wrapper w, rot, tr;
    for (const object* in; (in = s->wrapped(w)) != nullptr; s = in) {
        if (w.kind == flat_wrapper::translate && fl == 0) { tr = w; fl |= RT_PRIM_FLAG_TRANSLATE; }
        else if (w.kind == flat_wrapper::rotatey && !(fl & RT_PRIM_FLAG_ROTATEY)) { rot = w; fl |= RT_PRIM_FLAG_ROTATEY; }
        else if (w.kind != flat_wrapper::flip) throw std::runtime_error("volume: the boundary's wrappers must be translate(rotatey(.)), rotatey(.) or translate(.)");
    }
    if (const sphere* sp = dynamic_cast<const sphere*>(s)) {
        if (fl) throw std::runtime_error("volume: a wrapped sphere as the boundary is not supported (wrap the volume instead)");
        for (int k = 0; k < 3; k++) out.data[k] = sp->center[k];
        out.data[3] = sp->radius; out.data[4] = neg_inv_d;
        return true;
    }
    if (const box* b = dynamic_cast<const box*>(s)) {
        out.flags |= RT_PRIM_FLAG_BOUNDARY_BOX | fl;
        for (int k = 0; k < 3; k++) { out.data[k] = b->minimum[k]; out.data[3 + k] = b->maximum[k]; out.data[8 + k] = tr.target[k]; }
        out.data[6] = rot.sin_theta; out.data[7] = rot.cos_theta; out.data[11] = neg_inv_d;
        return true;
    }
    throw std::runtime_error("volume: the boundary must be a sphere or a box");
}
