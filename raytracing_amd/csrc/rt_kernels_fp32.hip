// fp32 fast kernels (FMA contraction allowed).
#define RT_REAL float
#include "rt_kernels.inl"
