// raytracing_amd/host/util.hpp -- constants and rand()-based helpers used while BUILDING scenes
// (reference: src/cpp/util.hpp:12-57).  Sampling during rendering does not use these: the HIP core has its
// own counter-based generator.  Draw order inside random_vector is fixed here to what g++ does for the
// reference on x86-64 (last constructor argument first), so a scene comes out the same with any compiler.
#pragma once

#include <cstdlib>
#include <limits>

#include "algebra.hpp"

const double infinity = std::numeric_limits<double>::infinity();
const double pi = 3.141592653589;   // the reference's truncated value (util.hpp:13): camera setup depends on it
const double epsilon = 1e-7;

inline double degree_to_arc(double degree) { return degree * pi / 180; }
inline double random_double() { return rand() / (RAND_MAX + 1.0); }
inline double random_double(double lo, double hi) { return lo + (hi - lo) * random_double(); }
inline int random_int(int lo, int hi) { return int(random_double(lo, hi + 1)); }
inline vec3d random_vector()
{
    double c = random_double(), b = random_double(), a = random_double();
    return vec3d(a, b, c);
}
inline vec3d random_vector(double lo, double hi)
{
    double c = random_double(lo, hi), b = random_double(lo, hi), a = random_double(lo, hi);
    return vec3d(a, b, c);
}
