// raytracing_amd/host/flatten.hpp -- object graph -> plain-C rt_scene_desc (include/rtcore.h).
//
// This is the step inserted under the reference's `scene`: after scene::build_bvh() the pointer tree of
// bvh/flip/primitive objects is walked once and written into flat, index-linked arrays that the C ABI copies to HBM.
// Primitive ids are the construction order of the leaf objects (for OBJ scenes: the face order of the file).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <vector>

#include "rtcore.h"

class object;
class material;
class texture;

// every scene entity records when it was constructed; flatten orders primitives by it
inline uint64_t rt_next_serial() { static uint64_t counter = 0; return counter++; }

struct flat_scene {
    std::vector<double> node_bbox, prim_data, mat_data, tex_data, light_data, xform_data;
    std::vector<int32_t> node_child, roots, prim_kind, prim_flags, prim_mat, mat_kind, mat_tex, tex_kind, light_kind, light_flags, light_mat;
    std::vector<int32_t> node_xform, prim_xform;     // instances (rt_scene_desc): transform index per node / primitive, -1 = world space
    std::vector<uint8_t> tex_pixels;
    double camera[24] = { 0 };

    rt_scene_desc desc() const
    {
        rt_scene_desc d;
        std::memset(&d, 0, sizeof d);
        d.n_nodes = (int32_t) (node_child.size() / 2); d.node_bbox = node_bbox.data(); d.node_child = node_child.data();
        d.n_roots = (int32_t) roots.size(); d.roots = roots.data();
        d.n_prims = (int32_t) prim_kind.size(); d.prim_kind = prim_kind.data(); d.prim_flags = prim_flags.data(); d.prim_mat = prim_mat.data();
        d.prim_data = prim_data.data();
        d.n_mats = (int32_t) mat_kind.size(); d.mat_kind = mat_kind.data(); d.mat_tex = mat_tex.data(); d.mat_data = mat_data.data();
        d.n_tex = (int32_t) tex_kind.size(); d.tex_kind = tex_kind.data(); d.tex_data = tex_data.data();
        d.n_pixel_bytes = tex_pixels.size(); d.tex_pixels = tex_pixels.data();
        d.n_lights = (int32_t) light_kind.size(); d.light_kind = light_kind.data(); d.light_flags = light_flags.data();
        d.light_mat = light_mat.data(); d.light_data = light_data.data();
        std::memcpy(d.camera, camera, sizeof camera);
        d.n_xforms = (int32_t) (xform_data.size() / RT_XFORM_STRIDE);
        if (d.n_xforms) { d.xform_data = xform_data.data(); d.node_xform = node_xform.data(); d.prim_xform = prim_xform.data(); }
        return d;
    }
};

// one record of an instance transform (rtcore.h xform_data): translate(rotatey(.)), rotatey(.) or translate(.); translate is the outer one
struct flat_xform {
    int flags = 0;                          // RT_PRIM_FLAG_ROTATEY | RT_PRIM_FLAG_TRANSLATE
    double sin_theta = 0, cos_theta = 1, target[3] = { 0, 0, 0 };
};

// one wrapper object as object::wrapped() describes it (reference: src/cpp/transform.hpp:9-149)
struct flat_wrapper {
    enum { none = 0, flip = 1, rotatey = 2, translate = 3 };
    int kind = none;
    double sin_theta = 0, cos_theta = 1, target[3] = { 0, 0, 0 };
};

// one leaf as the flattener sees it
struct flat_leaf {
    int kind = 0, flags = 0;
    const material* mat = nullptr;
    double data[RT_PRIM_STRIDE] = { 0 };
};

class flattener {
public:
    flat_scene out;

    int texture_id(const texture* t);     // defined in texture.hpp
    int material_id(const material* m);   // defined in material.hpp

    // pass 1: collect leaves (so ids can follow construction order); pass 2: emit nodes.  Both are defined in bvh.hpp.
    //   translate / rotatey around anything become instance transforms: the members keep their object-space geometry and name a CHAIN of
    //     records (rtcore.h RT_XFORM_FLAG_CHAIN), outermost first.  One record holds translate(rotatey(.)), rotatey(.) or translate(.): a
    //     rotatey joins the record before it when that one holds a lone translate; every other wrapper opens a new record;
    //   a box takes the innermost record into its own fields (RT_PRIM_FLAG_ROTATEY / _TRANSLATE), as cornell_box()'s boxes do;
    //   a flip counts only where no translate / rotatey lies outside it -- their set_face_normal calls overwrite `frontward`
    //     (transform.hpp:25, 105) -- and reaches the leaves below it as RT_PRIM_FLAG_FLIP (no chain) or RT_PRIM_FLAG_FLIP_OUTER;
    //   a group used as an object becomes group links (nodes with an inverted box): link(m0, link(m1, ...)).
    void collect(const object* o);
    int32_t emit(const object* o);        // returns the child code of o
    int32_t emit_group(object* const* items, int n);

    void assign_prim_ids()
    {
        std::sort(leaves.begin(), leaves.end(), [](const std::pair<uint64_t, const object*>& a, const std::pair<uint64_t, const object*>& b) { return a.first < b.first; });
        leaves.erase(std::unique(leaves.begin(), leaves.end()), leaves.end());
        for (size_t i = 0; i < leaves.size(); i++) prim_ids[leaves[i].second] = (int) i;
        out.prim_kind.resize(leaves.size()); out.prim_flags.resize(leaves.size()); out.prim_mat.resize(leaves.size());
        out.prim_data.assign(leaves.size() * RT_PRIM_STRIDE, 0.0);
        out.prim_xform.assign(leaves.size(), -1);
    }
    // the records of the current chain, written once per distinct chain: index of the first, -1 for no wrapper
    int chain_records()
    {
        if (chain.empty()) return -1;
        if (chain_first >= 0) return chain_first;
        chain_first = (int) (out.xform_data.size() / RT_XFORM_STRIDE);
        for (size_t k = 0; k < chain.size(); k++) {
            const flat_xform& x = chain[k];
            const bool rot = (x.flags & RT_PRIM_FLAG_ROTATEY) != 0;
            const double rec[RT_XFORM_STRIDE] = { rot ? x.sin_theta : 0.0, rot ? x.cos_theta : 1.0, x.target[0], x.target[1], x.target[2],
                                                  (double) (x.flags | (k + 1 < chain.size() ? RT_XFORM_FLAG_CHAIN : 0)) };
            out.xform_data.insert(out.xform_data.end(), rec, rec + RT_XFORM_STRIDE);
        }
        return chain_first;
    }
    std::vector<flat_xform> chain;          // the wrappers around the object being emitted, outermost first
    int chain_first = -1;                   // first record of `chain` once written
    int flip_parity = 0;                    // the flips that count
    std::set<int> emitted;                  // primitives whose record has been written
    void store_prim(int id, const flat_leaf& l)
    {
        out.prim_kind[id] = l.kind; out.prim_flags[id] = l.flags; out.prim_mat[id] = material_id(l.mat);
        std::memcpy(&out.prim_data[(size_t) id * RT_PRIM_STRIDE], l.data, sizeof l.data);
    }
    void store_light(const flat_leaf& l)
    {
        out.light_kind.push_back(l.kind); out.light_flags.push_back(l.flags); out.light_mat.push_back(l.mat ? material_id(l.mat) : 0);
        out.light_data.insert(out.light_data.end(), l.data, l.data + RT_PRIM_STRIDE);
    }

    std::vector<std::pair<uint64_t, const object*>> leaves;
    std::map<const object*, int> prim_ids;
    std::map<const material*, int> mat_ids;
    std::map<const texture*, int> tex_ids;
};
