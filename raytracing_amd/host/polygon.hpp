// raytracing_amd/host/polygon.hpp -- vertex + triangle (reference: src/cpp/polygon.hpp:15-183).
#pragma once

#include <algorithm>

#include "material.hpp"
#include "object.hpp"

struct vertex {
    vec3d position;
    vec2d uv_coord;
    vec3d normal;
};

class triangle : public object {
public:
    triangle(const vertex* v, material* m) : mat(m) { set(v[0], v[1], v[2]); }
    triangle(vertex v0, vertex v1, vertex v2, material* m) : mat(m) { set(v0, v1, v2); }
    // positions only: uv = (x, y) of each corner, one geometric normal per corner (reference polygon.hpp:54-80)
    triangle(vec3d p0, vec3d p1, vec3d p2, material* m) : mat(m)
    {
        vertex a, b, c;
        a.position = p0; b.position = p1; c.position = p2;
        a.uv_coord = vec2d(p0.x(), p0.y()); b.uv_coord = vec2d(p1.x(), p1.y()); c.uv_coord = vec2d(p2.x(), p2.y());
        a.normal = (p1 - p0).cross(p2 - p0).normalize();
        b.normal = (p2 - p1).cross(p0 - p1).normalize();
        c.normal = (p0 - p2).cross(p1 - p2).normalize();
        set(a, b, c);
    }
    bool bounding_box(double, double, aabb& bbox) const override
    {
        vec3d lo, hi;
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min({ verts[0].position[k], verts[1].position[k], verts[2].position[k] }) - epsilon;
            hi[k] = std::max({ verts[0].position[k], verts[1].position[k], verts[2].position[k] }) + epsilon;
        }
        bbox = aabb(lo, hi);
        return true;
    }
    bool leaf_record(flat_leaf& out) const override
    {
        out.kind = RT_PRIM_TRIANGLE; out.mat = mat;
        for (int k = 0; k < 3; k++) {
            out.data[k] = verts[0].position[k]; out.data[3 + k] = edge1[k]; out.data[6 + k] = edge2[k];
            out.data[9 + k] = verts[0].normal[k]; out.data[12 + k] = verts[1].normal[k]; out.data[15 + k] = verts[2].normal[k];
        }
        for (int v = 0; v < 3; v++) { out.data[18 + 2 * v] = verts[v].uv_coord[0]; out.data[19 + 2 * v] = verts[v].uv_coord[1]; }
        return true;
    }
    const vertex& corner(int i) const { return verts[i]; }
    material* get_material() const { return mat; }
private:
    void set(const vertex& a, const vertex& b, const vertex& c)
    {
        verts[0] = a; verts[1] = b; verts[2] = c;
        edge1 = b.position - a.position;
        edge2 = c.position - a.position;
    }
    vertex verts[3];
    vec3d edge1, edge2;
    material* mat;
};
