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

// ---- quantised four-wide nodes (rt_device.h WNode): the binary tree collapsed.  A wide node starts from a binary node's two children and, while it
// has fewer than four, replaces its inner child of the largest surface area by that child's two children.
struct WideNode { Box box[4]; int32_t child[4]; int n = 0; int axis = 0; };   // axis: the children are stored in ascending order of their centres along it
struct WideTree {
    std::vector<WideNode> nodes;
    int stack_need = 0;           // most entries a traversal can hold: a visit pushes up to n - 1 children
};

// bin_need[i]: stack entries the binary subtree under node i needs when walked as it is (one push per level).  `budget`: entries the
// wide subtree may use; a child is only expanded while every child's binary subtree still fits what the extra push leaves, so
// the tallest paths of the tree stay narrower and need <= budget holds by construction (given budget >= bin_need of the root).
inline int32_t widen_node(const Tree& t, const std::vector<int>& bin_need, int32_t bin, int budget, WideTree& out, int& need)
{
    WideNode w;
    const Node& b = t.nodes[bin];
    auto need_of = [&](int32_t code) { return code >= 0 ? bin_need[code] : 0; };
    w.box[0] = b.lbox; w.child[0] = b.left; w.n = 1;
    if (!(b.left == b.right && b.left < 0)) { w.box[1] = b.rbox; w.child[1] = b.right; w.n = 2; }     // a single-leaf tree names its leaf twice
    while (w.n >= 2 && w.n < 4) {
        int pick = -1; double area = -1;
        for (int k = 0; k < w.n; k++) {
            if (w.child[k] < 0 || w.box[k].area() <= area) continue;
            const Node& c = t.nodes[w.child[k]];
            int others = std::max(need_of(c.left), need_of(c.right));     // the tallest binary subtree among the children there would be
            for (int j = 0; j < w.n; j++) if (j != k) others = std::max(others, need_of(w.child[j]));
            if (others + w.n > budget) continue;           // w.n + 1 children push up to w.n entries
            area = w.box[k].area(); pick = k;
        }
        if (pick < 0) break;
        const Node& c = t.nodes[w.child[pick]];
        w.box[pick] = c.lbox; w.child[pick] = c.left;
        w.box[w.n] = c.rbox; w.child[w.n] = c.right; w.n++;
    }
    // Slot order = visiting order.  The children are sorted along the axis over which their centres spread most; a ray going up that
    // axis visits the slots forwards, a ray going down it backwards (rt_wave.h node_step): nearest-first to a good approximation
    // without the kernel comparing entry distances (tools/trav_policy_sim.cpp: +1..3 % visits and tests against an exact sort).
    {
        double best_spread = -1;
        for (int a = 0; a < 3; a++) {
            double lo = DBL_MAX, hi = -DBL_MAX;
            for (int k = 0; k < w.n; k++) { const double c = w.box[k].lo[a] + w.box[k].hi[a]; lo = std::min(lo, c); hi = std::max(hi, c); }
            if (hi - lo > best_spread) { best_spread = hi - lo; w.axis = a; }
        }
        const int a = w.axis;
        for (int i = 1; i < w.n; i++)            // insertion sort, stable: at most four entries
            for (int j = i; j > 0 && w.box[j].lo[a] + w.box[j].hi[a] < w.box[j - 1].lo[a] + w.box[j - 1].hi[a]; j--) {
                std::swap(w.box[j], w.box[j - 1]); std::swap(w.child[j], w.child[j - 1]);
            }
    }
    const int32_t idx = (int32_t) out.nodes.size();
    out.nodes.push_back(w);
    int deepest = 0;
    for (int k = 0; k < w.n; k++)
        if (w.child[k] >= 0) {
            int sub = 0;
            const int32_t c = widen_node(t, bin_need, w.child[k], budget - (w.n - 1), out, sub);
            out.nodes[idx].child[k] = c;
            deepest = std::max(deepest, sub);
        }
    need = deepest + w.n - 1;
    return idx;
}

inline WideTree widen(const Tree& t, int budget)
{
    WideTree w;
    if (t.nodes.empty()) return w;
    // children follow their parent in Tree::nodes (split() reserves the parent's slot first): one backward pass
    std::vector<int> bin_need(t.nodes.size(), 0);
    for (size_t i = t.nodes.size(); i-- > 0;) {
        const Node& n = t.nodes[i];
        bin_need[i] = 1 + std::max(n.left >= 0 ? bin_need[n.left] : 0, n.right >= 0 ? bin_need[n.right] : 0);
    }
    w.nodes.reserve(t.nodes.size() / 2 + 1);
    widen_node(t, bin_need, 0, std::max(budget, bin_need[0]), w, w.stack_need);
    return w;
}

// Which visiting order pays on this tree (rt_wave.h node_step)?  The slots of a wide node are sorted along one axis and walked forwards or
// backwards by the ray's sign on it -- no comparison of entry distances, 29 vector instructions fewer per visit.  On a surface mesh that
// order is nearly the distance order (+1.8 % box tests on the 1.3 M-triangle mesh, -0.6 % time); in a sparse soup of small triangles it
// is not (+27 % box tests, +9.6 % time on the 10 M-triangle soup).  This census walks `n_rays` bounced rays (origins on primitives chosen
// at random, directions uniform on the sphere: a frame's rays start on surfaces, not in the empty parts of the scene's box) through
// the tree under both orders, with real triangle tests so that the closest hit culls as it does in the kernel, and returns box tests
// per ray: [0] slot order, [1] distance order.
// `start(u in [0, 1), o)`: a point on the primitive that u picks; `hit(leaf_order_index, o, d, t_max, &t)`: the caller's primitive test.
template <typename Start, typename Hit>
inline void visit_order_census(const WideTree& wt, int n_rays, Start&& start, Hit&& hit, double out[2])
{
    out[0] = out[1] = 0;
    if (wt.nodes.empty() || n_rays <= 0) return;
    uint64_t s = 0x2545F4914F6CDD1Dull;
    auto unit = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double) (s >> 11) * (1.0 / 9007199254740992.0); };
    std::vector<int32_t> stack;
    for (int r = 0; r < n_rays; r++) {
        double o[3], d[3], inv[3];
        start(unit(), o);
        const double z = 2 * unit() - 1, phi = 6.283185307179586 * unit(), q = std::sqrt(std::max(0.0, 1 - z * z));
        d[0] = q * std::cos(phi); d[1] = q * std::sin(phi); d[2] = z;
        for (int k = 0; k < 3; k++) inv[k] = 1.0 / (std::fabs(d[k]) < 1e-30 ? 1e-30 : d[k]);
        for (int policy = 0; policy < 2; policy++) {
            double best = DBL_MAX;
            stack.assign(1, 0);
            while (!stack.empty()) {
                const int32_t cur = stack.back(); stack.pop_back();
                if (cur < 0) {
                    const int code = ~cur, first = code >> 3, cnt = (code & 7) + 1;
                    for (int k = 0; k < cnt; k++) { double t; if (hit(first + k, o, d, best, &t)) best = t; }
                    continue;
                }
                const WideNode& n = wt.nodes[cur];
                out[policy] += n.n;
                double tn[4]; int ord[4], m = 0;
                const bool rev = d[n.axis] < 0;
                for (int i = 0; i < n.n; i++) {
                    const int k = rev ? n.n - 1 - i : i;
                    double t0 = 1e-3, t1 = best;
                    for (int a = 0; a < 3; a++) {
                        double lo = (n.box[k].lo[a] - o[a]) * inv[a], hi = (n.box[k].hi[a] - o[a]) * inv[a];
                        if (lo > hi) std::swap(lo, hi);
                        t0 = std::max(t0, lo); t1 = std::min(t1, hi);
                    }
                    if (t0 <= t1) { tn[k] = t0; ord[m++] = k; }
                }
                if (policy == 1) std::sort(ord, ord + m, [&](int a, int b) { return tn[a] < tn[b]; });
                for (int i = m - 1; i >= 0; i--) stack.push_back(n.child[ord[i]]);      // the first to visit on top
            }
        }
    }
    out[0] /= n_rays; out[1] /= n_rays;
}

// One axis of a wide node (decoded in rt_wave.h slab_wide): lo[k] / hi[k] are the children's planes, already padded by the kernel's
// fp32 decode error.  word = a float origin <= all lo whose low byte is the signed exponent e of the step 2^e (the kernel uses the
// whole dword as the origin, so the origin must stay below the boxes whatever that byte holds); qlo / qhi: byte k = child k's low /
// high plane in steps from the origin, lo rounded down and hi up; slots k >= n hold lo 255, hi 0 (the kernel masks them by their
// child code).
inline void quantise_axis_wide(const double* lo, const double* hi, int n, uint32_t& word, uint32_t& qlo, uint32_t& qhi)
{
    double mn = lo[0], mx = hi[0];
    for (int k = 1; k < n; k++) { mn = std::min(mn, lo[k]); mx = std::max(mx, hi[k]); }
    // origin: a float <= mn whatever its low byte holds.  Setting low bits moves a negative float down and a positive one up, so a
    // positive origin starts one 256-ulp step lower.
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
    qlo = 0; qhi = 0;
    for (int k = 0; k < 4; k++) {
        uint32_t l8 = 255, h8 = 0;
        if (k < n) {
            l8 = (uint32_t) std::min(255.0, std::max(0.0, std::floor((lo[k] - origin) / step)));
            h8 = (uint32_t) std::min(255.0, std::max(0.0, std::ceil((hi[k] - origin) / step)));
        }
        qlo |= l8 << (8 * k); qhi |= h8 << (8 * k);
    }
}

}  // namespace rtb
