// tools/trav_policy_sim.cpp -- host-only census of traversal policies over the library's own SAH tree (csrc/rt_bvh_build.h).
//
// Question: the four-wide visit sorts its hit children by entry distance (a five-exchange network) and pushes up
// to three of them.  What do cheaper orders cost in visits and tests -- children stored sorted along one axis and walked forwards or
// backwards by the sign of the ray's direction on that axis (no sort, one stack entry per visit: base + mask), hit leaves tested before
// any inner child is entered, eight children per node?  This program answers in node visits, box tests, leaf tests, stack entries and
// stack depth per ray, for primary rays of the bench camera and two generations of cosine-distributed secondary rays, so that only
// the policies worth their instructions are built as kernels.
//
//   g++ -O2 -std=c++17 -I raytracing_amd/csrc tools/trav_policy_sim.cpp -o /tmp/trav_sim && /tmp/trav_sim mesh.obj [rays]
#include <array>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>

#include "rt_bvh_build.h"

struct V3 { double x, y, z; };
static V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static V3 operator*(double s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
static double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static V3 norm(V3 a) { double l = std::sqrt(dot(a, a)); return (1.0 / l) * a; }
static double comp(V3 a, int k) { return k == 0 ? a.x : k == 1 ? a.y : a.z; }

struct Tri { V3 p0, e1, e2; };
struct Ray { V3 o, d; };

static bool tri_hit(const Tri& t, const Ray& r, double tmax, double& tout)
{
    V3 q = cross(r.d, t.e2);
    double a = dot(t.e1, q);
    if (std::fabs(a) < 1e-7) return false;
    double f = 1.0 / a;
    V3 s = r.o - t.p0;
    double u = f * dot(s, q);
    if (u < 0) return false;
    V3 k = cross(s, t.e1);
    double v = f * dot(r.d, k);
    if (v < 0 || u + v > 1) return false;
    double tt = f * dot(t.e2, k);
    if (tt < 1e-3 || tt > tmax) return false;
    tout = tt;
    return true;
}

static bool box_hit(const rtb::Box& b, const Ray& r, const V3& inv, double tmax, double& tn)
{
    double t0 = 1e-3, t1 = tmax;
    for (int k = 0; k < 3; k++) {
        double i = comp(inv, k), o = comp(r.o, k);
        double a = (b.lo[k] - o) * i, c = (b.hi[k] - o) * i;
        if (a > c) std::swap(a, c);
        t0 = std::max(t0, a); t1 = std::min(t1, c);
    }
    tn = t0;
    return t0 <= t1;
}

// a wide node of up to W children, collapsed from the binary tree like rtb::widen (largest inner child opened first)
struct WN { std::vector<rtb::Box> box; std::vector<int32_t> child; int axis = 0; int split_axis[3] = { 0, 0, 0 }; };
struct WT { std::vector<WN> nodes; };

static int widen_rec(const rtb::Tree& t, int32_t bin, int W, WT& out)
{
    WN w;
    const rtb::Node& b = t.nodes[bin];
    w.box = { b.lbox, b.rbox }; w.child = { b.left, b.right };
    while ((int) w.child.size() < W) {
        int pick = -1; double area = -1;
        for (size_t k = 0; k < w.child.size(); k++) if (w.child[k] >= 0 && w.box[k].area() > area) { area = w.box[k].area(); pick = (int) k; }
        if (pick < 0) break;
        const rtb::Node& c = t.nodes[w.child[pick]];
        w.box[pick] = c.lbox; w.child[pick] = c.left;
        w.box.push_back(c.rbox); w.child.push_back(c.right);
    }
    // slot order: children sorted along the axis over which their centres spread most
    double spread[3];
    for (int a = 0; a < 3; a++) {
        double lo = 1e300, hi = -1e300;
        for (auto& bx : w.box) { double c = 0.5 * (bx.lo[a] + bx.hi[a]); lo = std::min(lo, c); hi = std::max(hi, c); }
        spread[a] = hi - lo;
    }
    w.axis = spread[1] > spread[0] ? (spread[2] > spread[1] ? 2 : 1) : (spread[2] > spread[0] ? 2 : 0);
    std::vector<int> idx(w.child.size());
    for (size_t k = 0; k < idx.size(); k++) idx[k] = (int) k;
    std::sort(idx.begin(), idx.end(), [&](int i, int j) { return w.box[i].lo[w.axis] + w.box[i].hi[w.axis] < w.box[j].lo[w.axis] + w.box[j].hi[w.axis]; });
    WN s = w;
    for (size_t k = 0; k < idx.size(); k++) { s.box[k] = w.box[idx[k]]; s.child[k] = w.child[idx[k]]; }
    const int me = (int) out.nodes.size();
    out.nodes.push_back(s);
    for (size_t k = 0; k < s.child.size(); k++)
        if (s.child[k] >= 0) { int c = widen_rec(t, s.child[k], W, out); out.nodes[me].child[k] = c; }
    return me;
}

struct Counts { double visits = 0, box = 0, leaf = 0, pushes = 0, entries = 0; int max_stack = 0; long rays = 0; long hits = 0; };
static std::vector<int> g_level;            // level of each wide node (root 0); set: traverse() adds its visits to g_level_visits
static std::vector<double> g_level_visits;

enum Policy { SORT_ALL, AXIS_ORDER, AXIS_LEAVES_FIRST, NEAREST_THEN_AXIS };

// returns closest t (or -1) and primitive; counts into c.  Stack entries hold one child each (SORT_ALL / NEAREST_THEN_AXIS: what the
// kernel does today) -- `entries` also counts the one-entry-per-visit alternative (a visit that leaves anything behind = 1 entry)
static double traverse(const WT& wt, const std::vector<Tri>& tris, const std::vector<int32_t>& order, const Ray& r, Policy pol, Counts& c, int& prim_out)
{
    V3 inv = { 1.0 / (std::fabs(r.d.x) < 1e-30 ? 1e-30 : r.d.x), 1.0 / (std::fabs(r.d.y) < 1e-30 ? 1e-30 : r.d.y), 1.0 / (std::fabs(r.d.z) < 1e-30 ? 1e-30 : r.d.z) };
    double best = 1e300; int prim = -1;
    std::vector<int32_t> stack;
    std::vector<int> group_depth;          // one-entry-per-visit stack depth model
    int groups = 0;
    auto leaf = [&](int32_t code) {
        int cc = ~code, first = cc >> 3, cnt = (cc & 7) + 1;
        for (int k = 0; k < cnt; k++) { double t; c.leaf++; if (tri_hit(tris[order[first + k]], r, best, t)) { best = t; prim = order[first + k]; } }
    };
    stack.push_back(0);
    int max_stack = 1;
    while (!stack.empty()) {
        int32_t cur = stack.back(); stack.pop_back();
        if (cur < 0) { leaf(cur); continue; }
        const WN& n = wt.nodes[cur];
        c.visits++;
        if (!g_level.empty()) g_level_visits[g_level[cur]]++;
        const int nc = (int) n.child.size();
        c.box += nc;
        double tn[8]; bool h[8]; int nh = 0;
        for (int k = 0; k < nc; k++) { h[k] = box_hit(n.box[k], r, inv, best, tn[k]); nh += h[k]; }
        if (nh == 0) continue;
        int ord[8], m = 0;
        const bool rev = comp(r.d, n.axis) < 0;
        for (int i = 0; i < nc; i++) { int k = rev ? nc - 1 - i : i; if (h[k]) ord[m++] = k; }
        if (pol == SORT_ALL) std::sort(ord, ord + m, [&](int a, int b) { return tn[a] < tn[b]; });
        else if (pol == NEAREST_THEN_AXIS) { int bi = 0; for (int i = 1; i < m; i++) if (tn[ord[i]] < tn[ord[bi]]) bi = i; int keep = ord[bi]; for (int i = bi; i > 0; i--) ord[i] = ord[i - 1]; ord[0] = keep; }
        else if (pol == AXIS_LEAVES_FIRST) {
            int o2[8], m2 = 0;
            for (int i = 0; i < m; i++) if (n.child[ord[i]] < 0) leaf(n.child[ord[i]]); else o2[m2++] = ord[i];
            // the leaves just tested may have shrunk `best`: inner children are NOT re-tested (the kernel would not either)
            m = m2; std::memcpy(ord, o2, sizeof(int) * m2);
            if (m == 0) continue;
        }
        if (m > 1) { c.entries++; groups++; }
        for (int i = m - 1; i >= 0; i--) stack.push_back(n.child[ord[i]]);
        c.pushes += m - 1;
        max_stack = std::max(max_stack, (int) stack.size());
    }
    c.max_stack = std::max(c.max_stack, max_stack);
    c.rays++;
    if (prim >= 0) c.hits++;
    prim_out = prim;
    return prim >= 0 ? best : -1;
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: trav_sim mesh.obj [n_primary]\n"); return 2; }
    const int n_primary = argc > 2 ? std::atoi(argv[2]) : 60000;
    std::vector<V3> v; std::vector<Tri> tris;
    std::ifstream in(argv[1]);
    std::string line;
    while (std::getline(in, line)) {
        if (line.size() > 2 && line[0] == 'v' && line[1] == ' ') { V3 p; std::sscanf(line.c_str() + 2, "%lf %lf %lf", &p.x, &p.y, &p.z); v.push_back(p); }
        else if (line.size() > 2 && line[0] == 'f' && line[1] == ' ') {
            std::istringstream ss(line.substr(2)); std::string tok; std::vector<int> idx;
            while (ss >> tok) idx.push_back(std::atoi(tok.c_str()) - 1);
            for (size_t k = 1; k + 1 < idx.size(); k++) tris.push_back({ v[idx[0]], v[idx[k]] - v[idx[0]], v[idx[k + 1]] - v[idx[0]] });
        }
    }
    std::printf("%zu triangles\n", tris.size());
    std::vector<rtb::Item> items(tris.size());
    for (size_t i = 0; i < tris.size(); i++) {
        V3 p[3] = { tris[i].p0, tris[i].p0 + tris[i].e1, tris[i].p0 + tris[i].e2 };
        for (auto& q : p) { double a[3] = { q.x, q.y, q.z }; items[i].box.grow(a); }
        for (int k = 0; k < 3; k++) { double pad = 1e-4 + 1e-6 * std::max(std::fabs(items[i].box.lo[k]), std::fabs(items[i].box.hi[k])); items[i].box.lo[k] -= pad; items[i].box.hi[k] += pad; }
        for (int k = 0; k < 3; k++) items[i].c[k] = 0.5 * (items[i].box.lo[k] + items[i].box.hi[k]);
        items[i].id = (int32_t) i;
    }
    rtb::Tree tree = rtb::Builder(items, 4, 1.0, 26, 64).build();
    std::printf("binary tree: %zu nodes, depth %d\n", tree.nodes.size(), tree.depth);

    // rays: the bench camera (bench.py OBJ_CAMERA["obj"]), then two generations of cosine-distributed secondary rays
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> U(0, 1);
    V3 pos = { 0, 1.8, 5.0 }, at = { 0, 1.0, 0 };
    V3 w = norm(pos - at), u = norm(cross({ 0, 1, 0 }, w)), vv = cross(w, u);
    const double hh = std::tan(40.0 * 3.14159265358979 / 360.0), hw = hh * 16.0 / 9.0;
    std::vector<Ray> rays[3];
    for (int i = 0; i < n_primary; i++) {
        double x = (2 * U(g) - 1) * hw, y = (2 * U(g) - 1) * hh;
        rays[0].push_back({ pos, x * u + y * vv - w });
    }
    for (int W : { 4, 8 }) {
        WT wt; widen_rec(tree, 0, W, wt);
        size_t slots = 0; for (auto& n : wt.nodes) slots += n.child.size();
        std::printf("== %d-wide: %zu nodes, %.2f children per node\n", W, wt.nodes.size(), (double) slots / wt.nodes.size());
        const char* names[] = { "sort all by distance (today)", "axis order, forwards/backwards by ray sign", "axis order, hit leaves first", "nearest by distance, rest in axis order" };
        // visits by tree level (4-wide, slot order): how much of the walk the top of the tree takes -- what a copy of the top levels in
        // LDS could serve
        g_level.assign(wt.nodes.size(), 0);
        std::vector<size_t> per_level(64, 0);
        for (size_t i = 0; i < wt.nodes.size(); i++) { per_level[g_level[i]]++; for (int32_t ch : wt.nodes[i].child) if (ch >= 0) g_level[ch] = g_level[i] + 1; }   // parents precede children
        g_level_visits.assign(64, 0.0);
        for (int gen = 0; gen < 3; gen++) {
            if (W == 4 && gen < 2) rays[gen + 1].clear();
            for (int pol = 0; pol < 4; pol++) {
                Counts c;
                std::mt19937_64 g2(777 + gen);
                for (const Ray& r : rays[gen]) {
                    int prim; double t = traverse(wt, tris, tree.order, r, (Policy) pol, c, prim);
                    if (W == 4 && pol == 0 && gen < 2 && t > 0) {       // next generation from this hit
                        V3 p = r.o + t * r.d; const Tri& T = tris[prim];
                        V3 n = norm(cross(T.e1, T.e2)); if (dot(n, r.d) > 0) n = -1.0 * n;
                        double r1 = U(g2), r2 = U(g2), phi = 6.283185307 * r1, s = std::sqrt(r2);
                        V3 a = std::fabs(n.x) > 0.9 ? V3{ 0, 1, 0 } : V3{ 1, 0, 0 }, b1 = norm(cross(n, a)), b0 = cross(n, b1);
                        V3 d = (std::cos(phi) * s) * b0 + (std::sin(phi) * s) * b1 + std::sqrt(1 - r2) * n;
                        rays[gen + 1].push_back({ p, d });
                    }
                }
                std::printf("  gen %d %-46s rays %7ld hit %4.1f%%  visits %6.2f  box tests %6.2f  leaf tests %5.2f  child pushes %5.2f  group entries %5.2f  max stack (children) %d\n",
                            gen, names[pol], c.rays, 100.0 * c.hits / std::max(1l, c.rays), c.visits / c.rays, c.box / c.rays, c.leaf / c.rays, c.pushes / c.rays, c.entries / c.rays, c.max_stack);
            }
        }
        double all = 0, cum = 0; size_t nodes = 0;
        for (double x : g_level_visits) all += x;
        for (int l = 0; l < 64 && per_level[l]; l++) { cum += g_level_visits[l]; nodes += per_level[l]; std::printf("  level %2d: %8zu nodes (%9zu so far, %8.1f KB)  visits %5.1f %%  cumulative %5.1f %%\n", l, per_level[l], nodes, nodes * 64 / 1024.0, 100 * g_level_visits[l] / all, 100 * cum / all); }
        g_level.clear();
    }
    return 0;
}
