// fp32 fast kernels (FMA contraction allowed).
#define RT_REAL float
#define RT_WITH_FAST_BVH 1
#define RT_WITH_WAVE 1
#include "rt_kernels.inl"
