// raytracing_amd/csrc/rt_bvh_build.h -- host-side SAH builder for the fast mode's own tree.
//
// The reference's builder (bvh.hpp:22-56: random axis, median split, unordered traversal) costs ~38 box tests per
// ray on the random-spheres scene.  The closest hit does not depend on the tree, so the fast mode builds its own:
// binned surface-area heuristic over all three axes, leaves of up to 8 primitives, children's boxes stored in the
// parent.  The reference tree stays resident for the parity entry points (rt_first_hit, RT_MODE_REFERENCE_FP64).
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rtb {

struct Box {
    double lo[3] = { DBL_MAX, DBL_MAX, DBL_MAX }, hi[3] = { -DBL_MAX, -DBL_MAX, -DBL_MAX };
    void grow(const Box& b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const double p[3]) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    double area() const
    {
        double d[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0;
        return 2 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
};

struct Item { Box box; double c[3]; int32_t id; };

struct Node { Box lbox, rbox; int32_t left, right; };   // child code: >= 0 node, < 0 leaf ~(first * 8 + count - 1)

struct Tree {
    std::vector<Node> nodes;
    std::vector<int32_t> order;   // primitive ids in leaf order
    int depth = 0;
};

class Builder {
public:
    // depth_limit: no leaf deeper than this (the traversal stack lives in LDS and its size decides how many workgroups share a
    // CU).  A split whose larger side could not be finished within the limit even by halving is replaced by a median split, so the
    // limit holds whenever 1 + balanced_height(n) <= depth_limit; otherwise the builder falls back to RT_BUILD_SAH_DEPTH.
    Builder(std::vector<Item>& items, int max_leaf, double cost_prim, int depth_limit = RT_BUILD_SAH_DEPTH, int bins = 16)
        : it(items), leaf_max(std::max(1, std::min(max_leaf, 8))), c_prim(cost_prim), limit(depth_limit), kBins(std::max(4, std::min((int) kMaxBins, bins)))
    {
        if (1 + balanced_height((int) items.size()) > limit) limit = std::max(limit, (int) RT_BUILD_SAH_DEPTH);
    }
    int balanced_height(int n) const { int h = 0; while (n > leaf_max) { n = (n + 1) / 2; h++; } return h; }

    Tree build()
    {
        Tree t;
        tree = &t;
        int n = (int) it.size();
        t.order.reserve(n);
        if (n == 0) return t;
        t.nodes.reserve(std::max(1, 2 * n / std::max(1, leaf_max / 2)));
        if (n == 1) {   // a node always has two children: test the single primitive from both
            Node root;
            root.lbox = root.rbox = it[0].box;
            t.order.push_back(it[0].id);
            root.left = root.right = ~(0 * 8 + 0);
            t.nodes.push_back(root);
            t.depth = 1;
            return t;
        }
        Box all;
        for (const Item& i : it) all.grow(i.box);
        int32_t root = split(0, n, all, 1);
        if (root < 0) {   // everything fitted one leaf: wrap it
            Node r;
            r.lbox = r.rbox = all;
            r.left = r.right = root;
            t.nodes.push_back(r);
            t.depth = std::max(t.depth, 1);
        }
        return t;
    }

private:
    std::vector<Item>& it;
    int leaf_max;
    double c_prim;
    int limit;
    Tree* tree = nullptr;
    static const int kMaxBins = 128;
    int kBins;

    int32_t make_leaf(int b, int e)
    {
        int first = (int) tree->order.size();
        for (int i = b; i < e; i++) tree->order.push_back(it[i].id);
        return ~(first * 8 + (e - b - 1));
    }

    // returns the child code of the subtree over items [b, e)
    int32_t split(int b, int e, const Box& bounds, int depth)
    {
        tree->depth = std::max(tree->depth, depth);
        int n = e - b;
        if (n == 1) return make_leaf(b, e);
        Box cb;
        for (int i = b; i < e; i++) cb.grow(it[i].c);
        double best_cost = DBL_MAX; int best_axis = -1, best_bin = -1;
        const double parent_area = std::max(bounds.area(), 1e-300);
        const bool deep = depth >= RT_BUILD_SAH_DEPTH;
        if (!deep) {
            for (int axis = 0; axis < 3; axis++) {
                double ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0)) continue;
                Box bb[kMaxBins]; int cnt[kMaxBins] = { 0 };
                double scale = kBins / ext;
                for (int i = b; i < e; i++) {
                    int k = std::min(kBins - 1, std::max(0, (int) ((it[i].c[axis] - cb.lo[axis]) * scale)));
                    cnt[k]++; bb[k].grow(it[i].box);
                }
                double right_area[kMaxBins]; int right_cnt[kMaxBins];
                Box acc; int c = 0;
                for (int k = kBins - 1; k > 0; k--) { acc.grow(bb[k]); c += cnt[k]; right_area[k] = acc.area(); right_cnt[k] = c; }
                Box lacc; int lc = 0;
                for (int k = 0; k < kBins - 1; k++) {
                    lacc.grow(bb[k]); lc += cnt[k];
                    if (lc == 0 || right_cnt[k + 1] == 0) continue;
                    double cost = 1.0 + c_prim * (lacc.area() * lc + right_area[k + 1] * right_cnt[k + 1]) / parent_area;
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = k; }
                }
            }
        }
        double leaf_cost = c_prim * n;
        if (n <= leaf_max && (best_axis < 0 || leaf_cost <= best_cost || depth >= limit)) return make_leaf(b, e);
        int mid;
        if (best_axis >= 0) {
            double ext = cb.hi[best_axis] - cb.lo[best_axis], scale = kBins / ext, lo = cb.lo[best_axis];
            int axis = best_axis, bin = best_bin;
            auto pivot = std::partition(it.begin() + b, it.begin() + e, [&](const Item& x) {
                int k = std::min(kBins - 1, std::max(0, (int) ((x.c[axis] - lo) * scale)));
                return k <= bin;
            });
            mid = (int) (pivot - it.begin());
        }
        else {   // coincident centroids or too deep: median split along the widest axis
            int axis = 0;
            for (int k = 1; k < 3; k++) if (bounds.hi[k] - bounds.lo[k] > bounds.hi[axis] - bounds.lo[axis]) axis = k;
            mid = b + n / 2;
            std::nth_element(it.begin() + b, it.begin() + mid, it.begin() + e, [axis](const Item& x, const Item& y) { return x.c[axis] < y.c[axis]; });
        }
        if (mid == b || mid == e) mid = b + n / 2;
        if (depth + 1 + balanced_height(std::max(mid - b, e - mid)) > limit && depth + balanced_height(n) <= limit) {
            // the larger side could not be finished within the depth limit: halve instead (along the widest axis of the centroids)
            int axis = 0;
            for (int k = 1; k < 3; k++) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
            mid = b + (n + 1) / 2;
            std::nth_element(it.begin() + b, it.begin() + mid, it.begin() + e, [axis](const Item& x, const Item& y) { return x.c[axis] < y.c[axis]; });
        }
        Box lb, rb;
        for (int i = b; i < mid; i++) lb.grow(it[i].box);
        for (int i = mid; i < e; i++) rb.grow(it[i].box);
        int32_t idx = (int32_t) tree->nodes.size();
        tree->nodes.push_back(Node());
        int32_t l = split(b, mid, lb, depth + 1);
        int32_t r = split(mid, e, rb, depth + 1);
        Node& nd = tree->nodes[idx];
        nd.lbox = lb; nd.rbox = rb; nd.left = l; nd.right = r;
        return idx;
    }

public:
    static const int RT_BUILD_SAH_DEPTH = 40;   // below this depth fall back to median splits: bounds the traversal stack
};

// One axis of a 32-byte quantised node (rt_device.h QNode, decoded in rt_wave.h slab_quant): lo[c] / hi[c] are the two children's
// planes, already padded by the kernel's fp32 decode error.  word = a float origin <= both lo whose low byte is the signed exponent e
// of the step 2^e (the kernel uses the whole dword as the origin, so the origin must stay below the boxes whatever that byte
// holds); q = bytes left lo, right lo, left hi, right hi in steps from the origin, lo rounded down and hi up.
inline void quantise_axis(const double lo[2], const double hi[2], uint32_t& word, uint32_t& q)
{
    const double mn = std::min(lo[0], lo[1]), mx = std::max(hi[0], hi[1]);
    // origin: a float <= mn whatever its low byte holds (the kernel reads the exponent byte as part of the origin).  Setting
    // low bits moves a negative float down and a positive one up, so a positive origin starts one 256-ulp step lower.
    float f = (float) mn; if ((double) f > mn) f = std::nextafterf(f, -INFINITY);
    uint32_t bits; std::memcpy(&bits, &f, 4);
    if (bits & 0x80000000u) { if (bits & 0xffu) bits = (bits | 0xffu) + 1u; }       // negative: magnitude up to the next multiple of 256 ulp
    else if (bits >= 0x200u) bits = (bits & ~0xffu) - 0x100u;
    else bits = 0x80000100u;                                                       // at zero: a tiny negative origin
    auto as_float = [](uint32_t u) { float v; std::memcpy(&v, &u, 4); return (double) v; };
    const double lowest = (bits & 0x80000000u) ? as_float(bits | 0xffu) : as_float(bits);
    const double ext = std::max(mx - lowest, 0.0);
    int e = (int) std::ceil(std::log2(std::max(ext / 254.0, 1e-37)));
    while (std::ldexp(254.0, e) < ext) e++;
    e = std::min(127, std::max(-126, e));
    word = bits | (uint32_t) (e & 0xff);
    const double origin = as_float(word), step = std::ldexp(1.0, e);
    q = 0;
    for (int c = 0; c < 2; c++) {   // bytes: left lo, right lo, left hi, right hi
        const double ql = std::floor((lo[c] - origin) / step), qh = std::ceil((hi[c] - origin) / step);
        const uint32_t l8 = (uint32_t) std::min(255.0, std::max(0.0, ql)), h8 = (uint32_t) std::min(255.0, std::max(0.0, qh));
        q |= (l8 << (8 * c)) | (h8 << (16 + 8 * c));
    }
}

}  // namespace rtb
