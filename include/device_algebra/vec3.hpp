// include/device_algebra/vec3.hpp -- HIP device value types for the path-tracing core.
//
// Fresh design (not the reference's CUDA copy, include/device_algebra/vec/vec3.hpp:1-126): a plain
// aggregate templated on the scalar so the same kernels instantiate for fp64 (reference-order
// parity mode) and fp32 (fast mode).  Operator semantics follow the reference's host vec3
// (include/algebra/vec/vec3.hpp): `v / s` is `(1/s) * v` (:121-123), dot sums x,y,z left to
// right (:67-72), normalize() of a zero vector is the zero vector (:60-65).  With contraction off
// (the fp64 translation unit) results are bit-identical to the g++ -O3 host build.
#pragma once

#include <hip/hip_runtime.h>

namespace rt {

template <typename T>
struct vec3 {
    T x, y, z;

    __host__ __device__ vec3() : x(0), y(0), z(0) {}
    __host__ __device__ vec3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
    template <typename U>
    __host__ __device__ explicit vec3(const vec3<U>& o) : x(T(o.x)), y(T(o.y)), z(T(o.z)) {}

    __host__ __device__ T operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    __host__ __device__ vec3 operator-() const { return vec3(-x, -y, -z); }
    __host__ __device__ vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
};

template <typename T> __host__ __device__ inline vec3<T> operator+(const vec3<T>& a, const vec3<T>& b) { return vec3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> __host__ __device__ inline vec3<T> operator-(const vec3<T>& a, const vec3<T>& b) { return vec3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> __host__ __device__ inline vec3<T> operator*(const vec3<T>& a, const vec3<T>& b) { return vec3<T>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename T> __host__ __device__ inline vec3<T> operator*(T s, const vec3<T>& a) { return vec3<T>(s * a.x, s * a.y, s * a.z); }
template <typename T> __host__ __device__ inline vec3<T> operator*(const vec3<T>& a, T s) { return s * a; }
template <typename T> __host__ __device__ inline vec3<T> operator/(const vec3<T>& a, T s) { return (T(1) / s) * a; }

template <typename T> __host__ __device__ inline T dot(const vec3<T>& a, const vec3<T>& b)
{
    T r1 = a.x * b.x, r2 = a.y * b.y, r3 = a.z * b.z;
    return r1 + r2 + r3;
}
template <typename T> __host__ __device__ inline vec3<T> cross(const vec3<T>& a, const vec3<T>& b)
{
    return vec3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__host__ __device__ inline float rt_sqrt(float v) { return sqrtf(v); }
__host__ __device__ inline double rt_sqrt(double v) { return sqrt(v); }
template <typename T> __host__ __device__ inline T length(const vec3<T>& a) { return rt_sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
template <typename T> __host__ __device__ inline vec3<T> normalize(const vec3<T>& a)
{
    T l = length(a);
    if (l == T(0)) return vec3<T>();
    return a / l;
}

// fp32 on the device: one v_rsq_f32 (1 ulp) instead of sqrt + reciprocal.  Used by the fp32 modes only, which are held to an
// RMSE tolerance; the fp64 reference mode keeps the reference's sqrt-then-divide.
template <> __host__ __device__ inline vec3<float> normalize<float>(const vec3<float>& a)
{
    const float l2 = a.x * a.x + a.y * a.y + a.z * a.z;
    if (l2 == 0.0f) return vec3<float>();
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_rsqf(l2) * a;
#else
    return (1.0f / sqrtf(l2)) * a;
#endif
}

}  // namespace rt
