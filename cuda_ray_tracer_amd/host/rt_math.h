// rt_math.h -- host-side value types of the scene API: vec3 / point3 / color, interval, ray, aabb.
//
// Mirrors the public surface of the reference's vec3.h, interval.h, ray.h and aabb.h so that host
// code written against the reference (scene construction, camera set-up, framebuffer access)
// compiles unchanged.  Host only: everything that ran in __device__ code in the reference lives in
// libcrt_hip.so behind include/crt_hip.h.
#ifndef CRT_HOST_RT_MATH_H
#define CRT_HOST_RT_MATH_H

#include <cmath>
#include <iostream>
#include <limits>

const float infinity = std::numeric_limits<float>::infinity();  // rtweekend.h:11
const float pi = 3.1415926535897932385f;                         // rtweekend.h:12

inline float degrees_to_radians(float degrees) { return float(double(degrees * pi) / 180.0); }  // rtweekend.h:16-18

// vec3.h:5-82 (fields and host-callable members)
class vec3 {
  public:
    float e[3];

    vec3() : e{0, 0, 0} {}
    vec3(float e0, float e1, float e2) : e{e0, e1, e2} {}

    float x() const { return e[0]; }
    float y() const { return e[1]; }
    float z() const { return e[2]; }
    float r() const { return e[0]; }
    float g() const { return e[1]; }
    float b() const { return e[2]; }

    vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    float operator[](int i) const { return e[i]; }
    float& operator[](int i) { return e[i]; }

    vec3& operator+=(const vec3& v) { for (int i = 0; i < 3; ++i) e[i] += v.e[i]; return *this; }
    vec3& operator*=(float t) { for (int i = 0; i < 3; ++i) e[i] *= t; return *this; }
    vec3& operator*=(const vec3& v) { for (int i = 0; i < 3; ++i) e[i] *= v.e[i]; return *this; }
    vec3& operator/=(float t) { return *this *= 1 / t; }

    float length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    float length() const { return sqrtf(length_squared()); }
    bool near_zero() const {
        const double s = 1e-8;
        return (fabsf(e[0]) < s) && (fabsf(e[1]) < s) && (fabsf(e[2]) < s);
    }
};

using point3 = vec3;
using color = vec3;  // color.h:7

inline std::ostream& operator<<(std::ostream& out, const vec3& v) { return out << v.e[0] << ' ' << v.e[1] << ' ' << v.e[2]; }
inline vec3 operator+(const vec3& u, const vec3& v) { return vec3(u.e[0] + v.e[0], u.e[1] + v.e[1], u.e[2] + v.e[2]); }
inline vec3 operator-(const vec3& u, const vec3& v) { return vec3(u.e[0] - v.e[0], u.e[1] - v.e[1], u.e[2] - v.e[2]); }
inline vec3 operator*(const vec3& u, const vec3& v) { return vec3(u.e[0] * v.e[0], u.e[1] * v.e[1], u.e[2] * v.e[2]); }
inline vec3 operator*(float t, const vec3& v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(const vec3& v, float t) { return t * v; }
inline vec3 operator/(const vec3& v, float t) { return (1 / t) * v; }
inline float dot(const vec3& u, const vec3& v) { return u.e[0] * v.e[0] + u.e[1] * v.e[1] + u.e[2] * v.e[2]; }
inline vec3 cross(const vec3& u, const vec3& v) {
    return vec3(u.e[1] * v.e[2] - u.e[2] * v.e[1], u.e[2] * v.e[0] - u.e[0] * v.e[2], u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
inline vec3 unit_vector(const vec3& v) { return v / v.length(); }
inline vec3 reflect(const vec3& v, const vec3& n) { return v - 2 * dot(v, n) * n; }

// interval.h:4-49
class interval {
  public:
    float min, max;
    interval() : min(+infinity), max(-infinity) {}
    interval(float lo, float hi) : min(lo), max(hi) {}
    interval(const interval& a, const interval& b) : min(a.min <= b.min ? a.min : b.min), max(a.max >= b.max ? a.max : b.max) {}
    float size() const { return max - min; }
    bool contains(float x) const { return min <= x && x <= max; }
    bool surrounds(float x) const { return min < x && x < max; }
    float clamp(float x) const { return x < min ? min : (x > max ? max : x); }
    interval expand(float delta) const { return interval(min - delta / 2, max + delta / 2); }
};
inline interval empty_interval() { return interval(+infinity, -infinity); }
inline interval universe_interval() { return interval(-infinity, +infinity); }

// ray.h:6-29
class ray {
  public:
    ray() : tm(0) {}
    ray(const point3& origin, const vec3& direction, float time) : orig(origin), dir(direction), tm(time) {}
    ray(const point3& origin, const vec3& direction) : ray(origin, direction, 0) {}
    const point3& origin() const { return orig; }
    const vec3& direction() const { return dir; }
    float time() const { return tm; }
    point3 at(float t) const { return orig + t * dir; }
    point3 orig;
    vec3 dir;
    float tm;
};

// aabb.h:4-32, 69-75 (the ray test runs on the GPU)
class aabb {
  public:
    interval x, y, z;
    aabb() {}
    aabb(const interval& ix, const interval& iy, const interval& iz) : x(ix), y(iy), z(iz) {}
    aabb(const point3& a, const point3& b)
        : x(a[0] <= b[0] ? interval(a[0], b[0]) : interval(b[0], a[0])),
          y(a[1] <= b[1] ? interval(a[1], b[1]) : interval(b[1], a[1])),
          z(a[2] <= b[2] ? interval(a[2], b[2]) : interval(b[2], a[2])) {}
    aabb(const aabb& b0, const aabb& b1) : x(b0.x, b1.x), y(b0.y, b1.y), z(b0.z, b1.z) {}
    const interval& axis_interval(int n) const { return n == 1 ? y : (n == 2 ? z : x); }
};
inline aabb empty_aabb() { return aabb(empty_interval(), empty_interval(), empty_interval()); }
inline aabb universe_aabb() { return aabb(universe_interval(), universe_interval(), universe_interval()); }

#endif
