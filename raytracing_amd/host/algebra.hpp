// raytracing_amd/host/algebra.hpp -- host-side vec2/vec3 with the reference's member API
// (include/algebra/vec/vec3.hpp, vec2.hpp, algebra.hpp:8-21) so scene-building code compiles unchanged.
// Only what scene construction needs; the hot-path arithmetic lives in the HIP core.
#pragma once

#include <cmath>
#include <ostream>

template <typename T>
class vec3 {
public:
    T content[3];

    vec3() : content{ T(0), T(0), T(0) } {}
    vec3(T a, T b, T c) : content{ a, b, c } {}

    T x() const { return content[0]; }
    T y() const { return content[1]; }
    T z() const { return content[2]; }
    T operator[](int i) const { return content[i]; }
    T& operator[](int i) { return content[i]; }

    T length() const { return std::sqrt(content[0] * content[0] + content[1] * content[1] + content[2] * content[2]); }
    vec3 normalize() const
    {
        T l = length();
        if (l == T(0)) return vec3();
        return *this / l;
    }
    T dot(const vec3& o) const
    {
        T a = content[0] * o.content[0], b = content[1] * o.content[1], c = content[2] * o.content[2];
        return a + b + c;
    }
    vec3 cross(const vec3& o) const
    {
        return vec3(content[1] * o.content[2] - content[2] * o.content[1], content[2] * o.content[0] - content[0] * o.content[2],
                    content[0] * o.content[1] - content[1] * o.content[0]);
    }

    vec3 operator-() const { return vec3(-content[0], -content[1], -content[2]); }
    vec3& operator+=(const vec3& o) { for (int i = 0; i < 3; i++) content[i] += o.content[i]; return *this; }
    vec3& operator-=(const vec3& o) { for (int i = 0; i < 3; i++) content[i] -= o.content[i]; return *this; }
    vec3& operator*=(T s) { for (int i = 0; i < 3; i++) content[i] *= s; return *this; }
    vec3& operator/=(T s) { return *this *= T(1) / s; }

    friend vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.content[0] + b.content[0], a.content[1] + b.content[1], a.content[2] + b.content[2]); }
    friend vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.content[0] - b.content[0], a.content[1] - b.content[1], a.content[2] - b.content[2]); }
    friend vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.content[0] * b.content[0], a.content[1] * b.content[1], a.content[2] * b.content[2]); }
    friend vec3 operator*(T s, const vec3& a) { return vec3(s * a.content[0], s * a.content[1], s * a.content[2]); }
    friend vec3 operator*(const vec3& a, T s) { return s * a; }
    friend vec3 operator/(const vec3& a, T s) { return (T(1) / s) * a; }   // reciprocal-multiply, as the reference divides
    friend std::ostream& operator<<(std::ostream& os, const vec3& v) { return os << "vec3(" << v.content[0] << ", " << v.content[1] << ", " << v.content[2] << ")"; }
};

template <typename T>
class vec2 {
public:
    T content[2];

    vec2() : content{ T(0), T(0) } {}
    vec2(T a, T b) : content{ a, b } {}
    T x() const { return content[0]; }
    T y() const { return content[1]; }
    T u() const { return content[0]; }
    T v() const { return content[1]; }
    T operator[](int i) const { return content[i]; }
    T& operator[](int i) { return content[i]; }
    friend vec2 operator+(const vec2& a, const vec2& b) { return vec2(a.content[0] + b.content[0], a.content[1] + b.content[1]); }
    friend vec2 operator-(const vec2& a, const vec2& b) { return vec2(a.content[0] - b.content[0], a.content[1] - b.content[1]); }
    friend vec2 operator*(T s, const vec2& a) { return vec2(s * a.content[0], s * a.content[1]); }
    friend vec2 operator*(const vec2& a, T s) { return s * a; }
};

typedef vec3<double> vec3d;
typedef vec3<float> vec3f;
typedef vec3<int> vec3i;
typedef vec2<double> vec2d;
typedef vec2<float> vec2f;
typedef vec2<int> vec2i;
