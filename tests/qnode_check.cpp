// tests/qnode_check.cpp -- host-side check of rtb::quantise_axis_wide (raytracing_amd/csrc/rt_bvh_build.h), the encoder of the quantised
// four-wide nodes, decoded the way rt_wave.h slab_wide reads them (whole dword as the origin, signed exponent byte, byte offsets), in
// exact arithmetic.  Built and run by tests/test_bvh_builder.py.  Prints key=value pairs.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "rt_bvh_build.h"

int main()
{
    std::mt19937_64 g(11);
    std::uniform_real_distribution<double> u(0, 1);
    long cases = 0, not_contained = 0, disordered = 0, loose = 0, clipped = 0, bad_empty = 0;
    double worst_slack_steps = 0;
    const double magnitudes[] = { 0.0, 1e-30, 1e-6, 1e-3, 0.37, 1.0, 2.0, 50.0, 1000.0, 1e6, 1e12 };
    const double sizes[] = { 0.0, 1e-9, 1e-6, 1e-3, 0.1, 1.0, 77.0, 1e4 };
    for (double mag : magnitudes) for (double size : sizes) for (int sign = -1; sign <= 1; sign += 2) for (int rep = 0; rep < 200; rep++) {
        const double base = sign * mag * (0.5 + u(g));
        const int n = 1 + rep % 4;
        double lo[4], hi[4];
        for (int c = 0; c < n; c++) { lo[c] = base + size * (u(g) - 0.5); hi[c] = lo[c] + size * u(g) * (rep % 7 == 0 ? 0.0 : 1.0); }
        if (rep % 13 == 0) { lo[0] = -size * u(g); hi[0] = size * u(g); }            // straddles zero
        uint32_t word, qlo, qhi;
        rtb::quantise_axis_wide(lo, hi, n, word, qlo, qhi);
        float of; std::memcpy(&of, &word, 4);
        const double origin = of, step = std::ldexp(1.0, (int) (int8_t) (word & 0xffu));
        cases++;
        for (int c = 0; c < 4; c++) {
            const unsigned ql = (qlo >> (8 * c)) & 0xffu, qh = (qhi >> (8 * c)) & 0xffu;
            if (c >= n) { if (ql != 255 || qh != 0) bad_empty++; continue; }
            const double dlo = origin + ql * step, dhi = origin + qh * step;
            if (!(dlo <= lo[c]) || !(dhi >= hi[c])) not_contained++;
            if (ql > qh) disordered++;
            if (qh == 255 && dhi < hi[c]) clipped++;
            // tightness: a plane is off by less than one step, plus what the origin gives away (<= 512 ulp of its magnitude)
            const double give = 600 * std::ldexp(std::fabs(origin) + 1e-300, -23);
            const double slack = std::max(lo[c] - dlo, dhi - hi[c]);
            if (slack > step + give) loose++;
            worst_slack_steps = std::max(worst_slack_steps, (slack - give) / step);
        }
    }
    std::printf("cases=%ld not_contained=%ld disordered=%ld clipped=%ld loose=%ld bad_empty=%ld worst_slack_milli_steps=%ld\n", cases, not_contained, disordered,
                clipped, loose, bad_empty, (long) std::ceil(worst_slack_steps * 1000));
    return 0;
}
