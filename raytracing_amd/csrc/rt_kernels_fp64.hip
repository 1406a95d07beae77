// fp64 reference-order kernels.  Build with -ffp-contract=off: no FMA, so +,-,*,/ and sqrt round exactly as in
// the reference's g++ -O3 x86-64 build ("fp64 vs fp32").
#define RT_REAL double
#define RT_WITH_FIRST_HIT 1
#define RT_WITH_WAVE 1
#define RT_OBJ_WAVE_KERNELS_ELSEWHERE 1      // the OBJ-loader wave kernels: rt_kernels_fp64_obj.hip
#include "rt_kernels.inl"
