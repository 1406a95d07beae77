// tests/qnode_check.cpp -- host-side check of rtb::quantise_axis (raytracing_amd/csrc/rt_bvh_build.h), the encoder of the 32-byte
// nodes, decoded the way rt_wave.h slab_quant reads them (whole dword as the origin, signed exponent byte, byte offsets), in exact
// arithmetic.  Built and run by tests/test_bvh_builder.py.  Prints key=value pairs.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "rt_bvh_build.h"

int main()
{
    std::mt19937_64 g(11);
    std::uniform_real_distribution<double> u(0, 1);
    long cases = 0, not_contained = 0, disordered = 0, loose = 0, clipped = 0;
    double worst_slack_steps = 0;
    const double magnitudes[] = { 0.0, 1e-30, 1e-6, 1e-3, 0.37, 1.0, 2.0, 50.0, 1000.0, 1e6, 1e12 };
    const double sizes[] = { 0.0, 1e-9, 1e-6, 1e-3, 0.1, 1.0, 77.0, 1e4 };
    for (double mag : magnitudes) for (double size : sizes) for (int sign = -1; sign <= 1; sign += 2) for (int rep = 0; rep < 200; rep++) {
        const double base = sign * mag * (0.5 + u(g));
        double lo[2], hi[2];
        for (int c = 0; c < 2; c++) { lo[c] = base + size * (u(g) - 0.5); hi[c] = lo[c] + size * u(g) * (rep % 7 == 0 ? 0.0 : 1.0); }
        if (rep % 13 == 0) { lo[0] = -size * u(g); hi[0] = size * u(g); }            // straddles zero
        uint32_t word, q;
        rtb::quantise_axis(lo, hi, word, q);
        float of; std::memcpy(&of, &word, 4);
        const double origin = of, step = std::ldexp(1.0, (int) (int8_t) (word & 0xffu));
        cases++;
        for (int c = 0; c < 2; c++) {
            const unsigned ql = (q >> (8 * c)) & 0xffu, qh = (q >> (16 + 8 * c)) & 0xffu;
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
    std::printf("cases=%ld not_contained=%ld disordered=%ld clipped=%ld loose=%ld worst_slack_milli_steps=%ld\n", cases, not_contained, disordered, clipped, loose,
                (long) std::ceil(worst_slack_steps * 1000));
    return 0;
}
