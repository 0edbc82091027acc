/*
 * glm/glm.hpp -- the subset of g-truc/glm (0.9.9.x, header-only; the `third_party/glm` submodule of the upstream
 * diff-gaussian-rasterization, NOT vendored in the reference tree) that the reference rasterizer uses:
 *   vec3, vec4, mat3 (column-major), operators, dot, length, max, transpose.
 * TEST INFRASTRUCTURE ONLY (oracle/build_ref.py).  The arithmetic follows glm's published implementation operation for
 * operation, so that results are bit-comparable under -ffp-contract=off:
 *   dot(a, b)        = tmp = a * b (component-wise); tmp.x + tmp.y + tmp.z          (detail/func_geometric.inl, compute_dot)
 *   length(v)        = sqrt(dot(v, v))
 *   mat3 * mat3      Result[c][r] = A[0][r] * B[c][0] + A[1][r] * B[c][1] + A[2][r] * B[c][2]   (detail/type_mat3x3.inl)
 *   scalar * mat3, mat3 * scalar, vec op scalar, vec op vec: component-wise
 *   transpose(m)[c][r] = m[r][c]
 *   mat3(a0..a8): columns (a0,a1,a2) (a3,a4,a5) (a6,a7,a8);  mat3(s): s on the diagonal
 */
#pragma once
#include <math.h>

namespace glm {

struct vec3 {
	float x, y, z;
	vec3() : x(0), y(0), z(0) {}
	vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
	explicit vec3(float s) : x(s), y(s), z(s) {}
	float& operator[](int i) { return (&x)[i]; }
	const float& operator[](int i) const { return (&x)[i]; }
	vec3& operator+=(const vec3& b) { x += b.x; y += b.y; z += b.z; return *this; }
	vec3& operator+=(float s) { x += s; y += s; z += s; return *this; }
	vec3& operator-=(const vec3& b) { x -= b.x; y -= b.y; z -= b.z; return *this; }
	vec3& operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
};
static inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, const vec3& a) { return vec3(s * a.x, s * a.y, s * a.z); }
static inline vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
static inline vec3 operator+(const vec3& a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
static inline vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }

struct vec4 {
	float x, y, z, w;
	vec4() : x(0), y(0), z(0), w(0) {}
	vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
	float& operator[](int i) { return (&x)[i]; }
	const float& operator[](int i) const { return (&x)[i]; }
};

static inline float dot(const vec3& a, const vec3& b)
{
	const vec3 tmp(a * b);
	return tmp.x + tmp.y + tmp.z;
}
static inline float dot(const vec4& a, const vec4& b)
{
	/* compute_dot<vec<4>>: (tmp.x + tmp.y) + (tmp.z + tmp.w) */
	const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z, tw = a.w * b.w;
	return (tx + ty) + (tz + tw);
}
static inline float length(const vec3& v) { return sqrtf(dot(v, v)); }
static inline float length(const vec4& v) { return sqrtf(dot(v, v)); }
static inline vec3 max(const vec3& a, float s) { return vec3(fmaxf(a.x, s), fmaxf(a.y, s), fmaxf(a.z, s)); }

struct mat3 {
	vec3 c[3];   // columns
	mat3() { c[0] = vec3(1, 0, 0); c[1] = vec3(0, 1, 0); c[2] = vec3(0, 0, 1); }
	explicit mat3(float s) { c[0] = vec3(s, 0, 0); c[1] = vec3(0, s, 0); c[2] = vec3(0, 0, s); }
	mat3(float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2, float z2)
	{
		c[0] = vec3(x0, y0, z0);
		c[1] = vec3(x1, y1, z1);
		c[2] = vec3(x2, y2, z2);
	}
	mat3(const vec3& a, const vec3& b, const vec3& d) { c[0] = a; c[1] = b; c[2] = d; }
	vec3& operator[](int i) { return c[i]; }
	const vec3& operator[](int i) const { return c[i]; }
};

static inline mat3 operator*(const mat3& m1, const mat3& m2)
{
	const float SrcA00 = m1[0][0], SrcA01 = m1[0][1], SrcA02 = m1[0][2];
	const float SrcA10 = m1[1][0], SrcA11 = m1[1][1], SrcA12 = m1[1][2];
	const float SrcA20 = m1[2][0], SrcA21 = m1[2][1], SrcA22 = m1[2][2];
	const float SrcB00 = m2[0][0], SrcB01 = m2[0][1], SrcB02 = m2[0][2];
	const float SrcB10 = m2[1][0], SrcB11 = m2[1][1], SrcB12 = m2[1][2];
	const float SrcB20 = m2[2][0], SrcB21 = m2[2][1], SrcB22 = m2[2][2];
	mat3 Result;
	Result[0][0] = SrcA00 * SrcB00 + SrcA10 * SrcB01 + SrcA20 * SrcB02;
	Result[0][1] = SrcA01 * SrcB00 + SrcA11 * SrcB01 + SrcA21 * SrcB02;
	Result[0][2] = SrcA02 * SrcB00 + SrcA12 * SrcB01 + SrcA22 * SrcB02;
	Result[1][0] = SrcA00 * SrcB10 + SrcA10 * SrcB11 + SrcA20 * SrcB12;
	Result[1][1] = SrcA01 * SrcB10 + SrcA11 * SrcB11 + SrcA21 * SrcB12;
	Result[1][2] = SrcA02 * SrcB10 + SrcA12 * SrcB11 + SrcA22 * SrcB12;
	Result[2][0] = SrcA00 * SrcB20 + SrcA10 * SrcB21 + SrcA20 * SrcB22;
	Result[2][1] = SrcA01 * SrcB20 + SrcA11 * SrcB21 + SrcA21 * SrcB22;
	Result[2][2] = SrcA02 * SrcB20 + SrcA12 * SrcB21 + SrcA22 * SrcB22;
	return Result;
}
static inline mat3 operator*(float s, const mat3& m) { return mat3(m[0] * s, m[1] * s, m[2] * s); }
static inline mat3 operator*(const mat3& m, float s) { return mat3(m[0] * s, m[1] * s, m[2] * s); }
static inline mat3 transpose(const mat3& m)
{
	mat3 r;
	r[0][0] = m[0][0]; r[0][1] = m[1][0]; r[0][2] = m[2][0];
	r[1][0] = m[0][1]; r[1][1] = m[1][1]; r[1][2] = m[2][1];
	r[2][0] = m[0][2]; r[2][1] = m[1][2]; r[2][2] = m[2][2];
	return r;
}

}  // namespace glm
