// raytracing_amd/host/bvh.hpp -- the reference's BVH builder, reproduced so the tree (and therefore the visit
// order of every ray) is the reference's (src/cpp/bvh.hpp:22-56):
//   * one random_int(0,2) per node, drawn before the span test;
//   * span 1: both children are the object; span 2: ascending by bbox.min[axis] at time (0,0);
//   * span > 2: the reference calls qsort with a comparator that returns +1 when a.min < b.min and -1 otherwise.
//     On glibc 2.35 qsort is a merge sort that takes the left run on cmp <= 0, so the result is a STABLE
//     DESCENDING order.  qsort's behaviour on such an inconsistent comparator is libc-specific, so this builder
//     carries its own stable merge sort instead of calling qsort (SURVEY 7 "Bit-exact BVH");
//   * split at first + span/2, left subtree built first; node box = union of the children's boxes at (start, end).
// Traversal is not here: csrc/rt_device.h closest_hit walks the flattened tree on the GPU.
#pragma once

#include <cstdio>
#include <vector>

#include "group.hpp"
#include "util.hpp"

class bvh : public object {
public:
    object* left = nullptr;
    object* right = nullptr;
    aabb bbox;

    bvh() {}
    bvh(group& grp, double start = 0, double end = 0) : bvh(grp.content(), 0, grp.size(), start, end) {}
    bvh(object** objects, int first, int last, double start, double end)
    {
        int axis = random_int(0, 2);
        int span = last - first;
        if (span == 1) {
            left = right = objects[first];
        }
        else if (span == 2) {
            bool ascending = key(objects[first], axis) < key(objects[first + 1], axis);
            left = objects[ascending ? first : first + 1];
            right = objects[ascending ? first + 1 : first];
        }
        else {
            sort_descending(objects + first, span, axis);
            int middle = first + span / 2;
            left = new bvh(objects, first, middle, start, end);
            right = new bvh(objects, middle, last, start, end);
        }
        aabb a, b;
        if (!left->bounding_box(start, end, a) || !right->bounding_box(start, end, b)) std::fprintf(stderr, "[ERROR] no bounding box in bvh constructor.");
        bbox = surrounding_box(a, b);
    }
    bool bounding_box(double, double, aabb& out) const override { out = bbox; return true; }

private:
    static double key(const object* o, int axis)
    {
        aabb b;
        o->bounding_box(0, 0, b);   // sort keys use the boxes at time (0,0), reference bvh.hpp:76-83
        return b.min()[axis];
    }
    static void sort_descending(object** a, int n, int axis)
    {
        std::vector<double> keys(n), tk(n);
        std::vector<object*> to(n);
        for (int i = 0; i < n; i++) keys[i] = key(a[i], axis);
        merge_sort(a, keys.data(), n, to.data(), tk.data());
    }
    static void merge_sort(object** a, double* k, int n, object** ta, double* tk)
    {
        if (n <= 1) return;
        int n1 = n / 2;
        merge_sort(a, k, n1, ta, tk);
        merge_sort(a + n1, k + n1, n - n1, ta, tk);
        int i = 0, j = n1, o = 0;
        while (i < n1 && j < n) {
            if (k[i] < k[j]) { ta[o] = a[j]; tk[o++] = k[j++]; }   // right run first only when strictly larger: stable, descending
            else { ta[o] = a[i]; tk[o++] = k[i++]; }
        }
        while (i < n1) { ta[o] = a[i]; tk[o++] = k[i++]; }
        while (j < n) { ta[o] = a[j]; tk[o++] = k[j++]; }
        for (int q = 0; q < n; q++) { a[q] = ta[q]; k[q] = tk[q]; }
    }
};

// ---- flattener traversal (needs the complete bvh / group types; the wrappers speak through object::wrapped)
inline void flattener::collect(const object* o)
{
    flat_wrapper w;
    while (const object* inner = o->wrapped(w)) o = inner;
    if (const bvh* b = dynamic_cast<const bvh*>(o)) { collect(b->left); if (b->right != b->left) collect(b->right); return; }
    if (const group* g = dynamic_cast<const group*>(o)) { for (int i = 0; i < g->size(); i++) collect(g->content()[i]); return; }
    leaves.push_back(std::make_pair(o->serial, o));
}

// a group used as an object (group.hpp:68-83): link(m0, link(m1, ... link(mk-1, mk))), a group of one is its member
inline int32_t flattener::emit_group(object* const* items, int n)
{
    if (n == 1) return emit(items[0]);
    int32_t idx = (int32_t) (out.node_child.size() / 2);
    out.node_child.push_back(0); out.node_child.push_back(0);
    out.node_xform.push_back(-1);
    for (int k = 0; k < 3; k++) out.node_bbox.push_back(infinity);             // inverted box: no box test (rtcore.h)
    for (int k = 0; k < 3; k++) out.node_bbox.push_back(-infinity);
    int32_t l = emit(items[0]);
    int32_t r = emit_group(items + 1, n - 1);
    out.node_child[(size_t) idx * 2] = l;
    out.node_child[(size_t) idx * 2 + 1] = r;
    return idx;
}

inline int32_t flattener::emit(const object* o)
{
    flat_wrapper w;
    if (const object* inner = o->wrapped(w)) {
        if (w.kind == flat_wrapper::flip) {                  // counts only where no translate / rotatey lies outside it
            const bool counts = chain.empty();
            if (counts) flip_parity ^= 1;
            int32_t code = emit(inner);
            if (counts) flip_parity ^= 1;
            return code;
        }
        const std::vector<flat_xform> saved = chain;
        const int saved_first = chain_first;
        if (w.kind == flat_wrapper::rotatey && !chain.empty() && chain.back().flags == RT_PRIM_FLAG_TRANSLATE) {
            chain.back().flags |= RT_PRIM_FLAG_ROTATEY; chain.back().sin_theta = w.sin_theta; chain.back().cos_theta = w.cos_theta;
        }
        else {
            flat_xform x;
            if (w.kind == flat_wrapper::translate) { x.flags = RT_PRIM_FLAG_TRANSLATE; for (int k = 0; k < 3; k++) x.target[k] = w.target[k]; }
            else { x.flags = RT_PRIM_FLAG_ROTATEY; x.sin_theta = w.sin_theta; x.cos_theta = w.cos_theta; }
            chain.push_back(x);
        }
        chain_first = -1;
        int32_t code = emit(inner);
        chain = saved; chain_first = saved_first;
        return code;
    }
    if (const group* g = dynamic_cast<const group*>(o)) {
        if (g->size() == 0) throw std::runtime_error("flatten: an empty group inside the scene");
        return emit_group(g->content(), g->size());
    }
    if (const bvh* b = dynamic_cast<const bvh*>(o)) {
        int32_t idx = (int32_t) (out.node_child.size() / 2);
        out.node_child.push_back(0); out.node_child.push_back(0);
        out.node_xform.push_back(chain_records());           // an instanced sub-tree stays in its own frame
        for (int k = 0; k < 3; k++) out.node_bbox.push_back(b->bbox.min()[k]);
        for (int k = 0; k < 3; k++) out.node_bbox.push_back(b->bbox.max()[k]);
        int32_t l = emit(b->left);
        int32_t r = b->right == b->left ? l : emit(b->right);      // span 1 (reference bvh.hpp:31-34): the same child twice
        out.node_child[(size_t) idx * 2] = l;
        out.node_child[(size_t) idx * 2 + 1] = r;
        return idx;
    }
    flat_leaf leaf;
    if (!o->leaf_record(leaf)) throw std::runtime_error("flatten: object type is not supported inside a scene (unknown class)");
    int id = prim_ids.at(o);
    const std::vector<flat_xform> saved = chain;
    const int saved_first = chain_first;
    if (leaf.kind == RT_PRIM_BOX && !chain.empty()) {        // the innermost record moves into the box's own fields
        const flat_xform& x = chain.back();
        leaf.flags |= x.flags;
        const bool rot = (x.flags & RT_PRIM_FLAG_ROTATEY) != 0;
        leaf.data[6] = rot ? x.sin_theta : 0.0; leaf.data[7] = rot ? x.cos_theta : 1.0;
        for (int k = 0; k < 3; k++) leaf.data[8 + k] = x.target[k];
        chain.pop_back(); chain_first = -1;
    }
    const int xi = chain_records();
    if (chain.size() != saved.size()) { chain = saved; chain_first = saved_first; }
    if (flip_parity) leaf.flags |= xi >= 0 ? RT_PRIM_FLAG_FLIP_OUTER : RT_PRIM_FLAG_FLIP;
    // one object reached along two paths (both children of a span-1 node, or the same pointer added twice) is one primitive: fine as long
    // as both paths wrap it alike -- a record has one set of flags and one transform chain
    if (emitted.count(id)) {
        const double* a = &out.xform_data[0];
        bool same = out.prim_flags[id] == leaf.flags && (out.prim_xform[id] >= 0) == (xi >= 0);
        for (int x0 = out.prim_xform[id], x1 = xi; same && x0 >= 0; x0++, x1++) {
            for (int k = 0; k < RT_XFORM_STRIDE; k++) same &= a[(size_t) x0 * RT_XFORM_STRIDE + k] == a[(size_t) x1 * RT_XFORM_STRIDE + k];
            if (!((int) a[(size_t) x0 * RT_XFORM_STRIDE + 5] & RT_XFORM_FLAG_CHAIN)) break;
        }
        if (!same) throw std::runtime_error("flatten: one object is used under two different sets of wrappers (give each use its own object)");
        return ~id;
    }
    emitted.insert(id);
    store_prim(id, leaf);
    out.prim_xform[id] = xi;
    return ~id;
}
