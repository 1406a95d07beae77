// tests/visit_order_check.cpp -- host-side check of rtb::visit_order_census (raytracing_amd/csrc/rt_bvh_build.h), built and run by
// tests/test_bvh_builder.py: a tessellated surface (a bumpy sphere) and a sparse soup of small triangles through the builder, the
// collapse to wide nodes and the census.  Prints key=value pairs: box tests per incoherent ray in slot order and by distance.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "rt_bvh_build.h"

struct Tri { double p0[3], e1[3], e2[3]; };

static bool tri_hit(const Tri& g, const double* o, const double* d, double t_max, double* t_out)
{
    const double qx = d[1] * g.e2[2] - d[2] * g.e2[1], qy = d[2] * g.e2[0] - d[0] * g.e2[2], qz = d[0] * g.e2[1] - d[1] * g.e2[0];
    const double a = g.e1[0] * qx + g.e1[1] * qy + g.e1[2] * qz;
    if (std::fabs(a) < 1e-12) return false;
    const double f = 1.0 / a, sx = o[0] - g.p0[0], sy = o[1] - g.p0[1], sz = o[2] - g.p0[2];
    const double u = f * (sx * qx + sy * qy + sz * qz);
    if (u < 0) return false;
    const double kx = sy * g.e1[2] - sz * g.e1[1], ky = sz * g.e1[0] - sx * g.e1[2], kz = sx * g.e1[1] - sy * g.e1[0];
    const double v = f * (d[0] * kx + d[1] * ky + d[2] * kz);
    if (v < 0 || u + v > 1) return false;
    const double tt = f * (g.e2[0] * kx + g.e2[1] * ky + g.e2[2] * kz);
    if (tt < 1e-3 || tt > t_max) return false;
    *t_out = tt;
    return true;
}

static void run(const char* name, const std::vector<Tri>& tris)
{
    std::vector<rtb::Item> items(tris.size());
    for (size_t i = 0; i < tris.size(); i++) {
        for (int v = 0; v < 3; v++) {
            double p[3];
            for (int k = 0; k < 3; k++) p[k] = tris[i].p0[k] + (v == 1 ? tris[i].e1[k] : v == 2 ? tris[i].e2[k] : 0.0);
            items[i].box.grow(p);
        }
        for (int k = 0; k < 3; k++) { items[i].box.lo[k] -= 1e-6; items[i].box.hi[k] += 1e-6; items[i].c[k] = 0.5 * (items[i].box.lo[k] + items[i].box.hi[k]); }
        items[i].id = (int32_t) i;
    }
    rtb::Tree t = rtb::Builder(items, 2, 1.0, 26, 64).build();
    rtb::WideTree w = rtb::widen(t, 35);
    double tests[2];
    rtb::visit_order_census(w, 2048, [&](double u, double* o) { const Tri& g = tris[(size_t) (u * tris.size()) % tris.size()]; for (int k = 0; k < 3; k++) o[k] = g.p0[k] + (g.e1[k] + g.e2[k]) / 3; },
                            [&](int at, const double* o, const double* d, double t_max, double* t_out) { return tri_hit(tris[t.order[at]], o, d, t_max, t_out); }, tests);
    std::printf("%s_slot=%.3f %s_dist=%.3f ", name, tests[0], name, tests[1]);
}

int main()
{
    std::mt19937 g(11);
    std::uniform_real_distribution<double> u(0, 1);
    std::normal_distribution<double> nrm(0, 1);
    // surface: a sphere of radius 1 with bumps, 200 x 100 quads of two triangles
    std::vector<Tri> mesh;
    auto P = [](int i, int j, double* p) {
        const double th = 3.141592653589793 * (j + 0.5) / 101.0, ph = 6.283185307179586 * i / 200.0, r = 1.0 + 0.08 * std::sin(9 * th) * std::cos(7 * ph);
        p[0] = r * std::sin(th) * std::cos(ph); p[1] = r * std::cos(th); p[2] = r * std::sin(th) * std::sin(ph);
    };
    for (int i = 0; i < 200; i++) for (int j = 0; j < 100; j++) {
        double a[3], b[3], c[3], d[3];
        P(i, j, a); P(i + 1, j, b); P(i + 1, j + 1, c); P(i, j + 1, d);
        Tri t1, t2;
        for (int k = 0; k < 3; k++) { t1.p0[k] = a[k]; t1.e1[k] = b[k] - a[k]; t1.e2[k] = c[k] - a[k]; t2.p0[k] = a[k]; t2.e1[k] = c[k] - a[k]; t2.e2[k] = d[k] - a[k]; }
        mesh.push_back(t1); mesh.push_back(t2);
    }
    run("mesh", mesh);
    // soup: 60 000 triangles of edge ~0.2 scattered in a cube of side 18 (the density of BASELINE configs[3])
    std::vector<Tri> soup(60000);
    for (Tri& t : soup) {
        double c[3] = { 18 * u(g) - 9, 18 * u(g) - 9, 18 * u(g) - 9 }, p[3][3];
        for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) p[v][k] = c[k] + 0.1 * nrm(g);
        for (int k = 0; k < 3; k++) { t.p0[k] = p[0][k]; t.e1[k] = p[1][k] - p[0][k]; t.e2[k] = p[2][k] - p[0][k]; }
    }
    run("soup", soup);
    std::printf("\n");
    return 0;
}
