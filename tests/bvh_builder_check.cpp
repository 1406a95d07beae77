// tests/bvh_builder_check.cpp -- host-side check of the fast mode's SAH builder (raytracing_amd/csrc/rt_bvh_build.h), built and run by
// tests/test_bvh_builder.py.  usage: bvh_builder_check <n_items> <depth_limit> <bins> <leaf_max> [wide_stack_budget] ; prints one line of key=value pairs.
// The w_* keys describe the tree collapsed to four-wide nodes (rtb::widen).
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>

#include "rt_bvh_build.h"

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const int n = std::atoi(argv[1]), limit = std::atoi(argv[2]), bins = std::atoi(argv[3]), leaf_max = std::atoi(argv[4]);
    std::mt19937 g(7);
    std::uniform_real_distribution<double> u(0, 1);
    std::vector<rtb::Item> items(n);
    for (int i = 0; i < n; i++) {   // clustered, anisotropic boxes; a few duplicates of one centroid (the builder's median fallback)
        double c[3] = { u(g) * u(g) * u(g) * 40 - 5, u(g) * 3, u(g) * u(g) * 10 };
        if (i % 97 == 0) c[0] = c[1] = c[2] = 1.25;
        double lo[3], hi[3];
        for (int k = 0; k < 3; k++) { lo[k] = c[k] - 0.02 * u(g); hi[k] = c[k] + 0.02 * u(g) * (k == 1 ? 5 : 1); }
        items[i].box.grow(lo); items[i].box.grow(hi);
        for (int k = 0; k < 3; k++) items[i].c[k] = c[k];
        items[i].id = i;
    }
    const std::vector<rtb::Item> original = items;
    rtb::Tree t = rtb::Builder(items, leaf_max, 1.0, limit, bins).build();

    std::vector<int> seen_prim(n, 0), seen_node(t.nodes.size(), 0);
    long bad_box = 0, bad_leaf = 0, shared = 0;
    int max_depth = 0;
    auto inside = [](const rtb::Box& outer, const rtb::Box& inner) {
        for (int k = 0; k < 3; k++) if (inner.lo[k] < outer.lo[k] || inner.hi[k] > outer.hi[k]) return false;
        return true;
    };
    // returns the bounds of the subtree under child code c, checking the box the parent stores for it
    std::function<rtb::Box(int32_t, int)> walk = [&](int32_t c, int depth) {
        rtb::Box b;
        if (depth > max_depth) max_depth = depth;
        if (c < 0) {
            const int code = ~c, first = code >> 3, count = (code & 7) + 1;
            if (count > leaf_max) bad_leaf++;
            for (int k = 0; k < count; k++) { const int id = t.order[first + k]; seen_prim[id]++; b.grow(original[id].box); }
            return b;
        }
        if (seen_node[c]++) { shared++; return b; }
        const rtb::Node& nd = t.nodes[c];
        const rtb::Box l = walk(nd.left, depth + 1), r = walk(nd.right, depth + 1);
        if (!inside(nd.lbox, l) || !inside(nd.rbox, r)) bad_box++;
        b.grow(nd.lbox); b.grow(nd.rbox);
        return b;
    };
    if (!t.nodes.empty()) walk(0, 1);
    long missing = 0, repeated = 0, unreachable = 0;
    for (int v : seen_prim) { missing += v == 0; repeated += v > 1; }
    for (int v : seen_node) unreachable += v == 0;
    std::printf("n=%d nodes=%zu depth=%d walked_depth=%d missing=%ld repeated=%ld unreachable=%ld shared=%ld bad_box=%ld bad_leaf=%ld", n, t.nodes.size(),
                t.depth, max_depth, missing, repeated, unreachable, shared, bad_box, bad_leaf);

    // the same tree in four-wide nodes: still a partition, every stored box contains what hangs below it, and no traversal order can
    // hold more stack entries than the tree reports (a visit pushes at most n - 1 children)
    const int budget = argc > 5 ? std::atoi(argv[5]) : 35;
    const rtb::WideTree w = rtb::widen(t, budget);
    std::vector<int> w_prim(n, 0), w_node(w.nodes.size(), 0);
    long w_bad_box = 0, w_shared = 0, w_bad_n = 0, w_full = 0, w_unsorted = 0;
    int w_worst_stack = 0;
    std::function<rtb::Box(int32_t, int)> wwalk = [&](int32_t c, int on_stack) {
        rtb::Box b;
        if (c < 0) {
            const int code = ~c, first = code >> 3, count = (code & 7) + 1;
            for (int k = 0; k < count; k++) { const int id = t.order[first + k]; w_prim[id]++; b.grow(original[id].box); }
            return b;
        }
        if (w_node[c]++) { w_shared++; return b; }
        const rtb::WideNode& nd = w.nodes[c];
        if (nd.n < 1 || nd.n > 4 || (nd.n == 1 && n != 1)) w_bad_n++;
        w_full += nd.n == 4;
        // slot order is visiting order: the children's centres ascend along the node's sort axis, which is the axis of their widest spread
        if (nd.axis < 0 || nd.axis > 2) w_unsorted++;
        else {
            for (int k = 1; k < nd.n; k++) if (nd.box[k].lo[nd.axis] + nd.box[k].hi[nd.axis] < nd.box[k - 1].lo[nd.axis] + nd.box[k - 1].hi[nd.axis]) w_unsorted++;
            double spread[3];
            for (int a = 0; a < 3; a++) {
                double lo = 1e300, hi = -1e300;
                for (int k = 0; k < nd.n; k++) { const double c = nd.box[k].lo[a] + nd.box[k].hi[a]; lo = std::min(lo, c); hi = std::max(hi, c); }
                spread[a] = hi - lo;
            }
            for (int a = 0; a < 3; a++) if (spread[a] > spread[nd.axis]) w_unsorted++;
        }
        if (on_stack + nd.n - 1 > w_worst_stack) w_worst_stack = on_stack + nd.n - 1;
        for (int k = 0; k < nd.n; k++) {
            const rtb::Box sub = wwalk(nd.child[k], on_stack + nd.n - 1);      // worst case: every sibling still waits on the stack
            if (!inside(nd.box[k], sub)) w_bad_box++;
            b.grow(nd.box[k]);
        }
        return b;
    };
    if (!w.nodes.empty()) wwalk(0, 0);
    long w_missing = 0, w_repeated = 0, w_unreachable = 0;
    for (int v : w_prim) { w_missing += v == 0; w_repeated += v > 1; }
    for (int v : w_node) w_unreachable += v == 0;
    std::printf(" w_nodes=%zu w_full=%ld w_stack=%d w_worst_stack=%d w_missing=%ld w_repeated=%ld w_unreachable=%ld w_shared=%ld w_bad_box=%ld w_bad_n=%ld w_unsorted=%ld\n",
                w.nodes.size(), w_full, w.stack_need, w_worst_stack, w_missing, w_repeated, w_unreachable, w_shared, w_bad_box, w_bad_n, w_unsorted);
    return 0;
}
