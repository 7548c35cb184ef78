// vkv_math.hpp — the handful of glm operations the reference's host code uses on this path
// (glm::mat4 column-major layout, so uniform blocks keep the reference's byte layout).
// Call sites mirrored: src/load_volume.cpp:82-83 (rotate, scale, radians),
// src/volume_render_subpass.cpp:226-239 (inverse, translate, inverseTranspose).
#pragma once

#include <cmath>
#include <cstring>

namespace vkv
{

struct vec3
{
	float x = 0, y = 0, z = 0;
};

struct vec4
{
	float x = 0, y = 0, z = 0, w = 0;
};

// column-major: m[col * 4 + row]
struct mat4
{
	float m[16];

	mat4() { identity(); }
	explicit mat4(const float *p) { std::memcpy(m, p, sizeof(m)); }
	void identity()
	{
		for (int i = 0; i < 16; ++i)
			m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
	}
	float &      at(int row, int col) { return m[col * 4 + row]; }
	const float &at(int row, int col) const { return m[col * 4 + row]; }
	const float *data() const { return m; }
};

inline mat4 operator*(const mat4 &a, const mat4 &b)
{
	mat4 r;
	for (int c = 0; c < 4; ++c)
		for (int row = 0; row < 4; ++row)
		{
			float s = 0.0f;
			for (int k = 0; k < 4; ++k)
				s += a.at(row, k) * b.at(k, c);
			r.at(row, c) = s;
		}
	return r;
}

inline vec4 operator*(const mat4 &a, const vec4 &v)
{
	vec4 r;
	r.x = a.at(0, 0) * v.x + a.at(0, 1) * v.y + a.at(0, 2) * v.z + a.at(0, 3) * v.w;
	r.y = a.at(1, 0) * v.x + a.at(1, 1) * v.y + a.at(1, 2) * v.z + a.at(1, 3) * v.w;
	r.z = a.at(2, 0) * v.x + a.at(2, 1) * v.y + a.at(2, 2) * v.z + a.at(2, 3) * v.w;
	r.w = a.at(3, 0) * v.x + a.at(3, 1) * v.y + a.at(3, 2) * v.z + a.at(3, 3) * v.w;
	return r;
}

inline mat4 transpose(const mat4 &a)
{
	mat4 r;
	for (int c = 0; c < 4; ++c)
		for (int row = 0; row < 4; ++row)
			r.at(row, c) = a.at(c, row);
	return r;
}

// Adjugate / determinant inverse (what glm::inverse computes), templated so the ray generator can run it in double.
template <typename T>
inline bool invert4x4(const T *m, T *out)
{
	T inv[16];
	inv[0]  = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
	inv[4]  = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
	inv[8]  = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
	inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
	inv[1]  = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
	inv[5]  = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
	inv[9]  = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
	inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
	inv[2]  = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
	inv[6]  = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
	inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
	inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
	inv[3]  = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
	inv[7]  = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
	inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
	inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
	const T det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
	if (det == T(0))
		return false;
	const T id = T(1) / det;
	for (int i = 0; i < 16; ++i)
		out[i] = inv[i] * id;
	return true;
}

inline mat4 inverse(const mat4 &a)
{
	mat4 r;
	invert4x4<float>(a.m, r.m);
	return r;
}

inline mat4 inverse_transpose(const mat4 &a) { return transpose(inverse(a)); }

inline mat4 translate(const vec3 &t)
{
	mat4 r;
	r.at(0, 3) = t.x, r.at(1, 3) = t.y, r.at(2, 3) = t.z;
	return r;
}

inline mat4 scale(const vec3 &s)
{
	mat4 r;
	r.at(0, 0) = s.x, r.at(1, 1) = s.y, r.at(2, 2) = s.z;
	return r;
}

inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// glm::rotate(angle, axis)
inline mat4 rotate(float angle, const vec3 &axis_in)
{
	const float c = std::cos(angle), s = std::sin(angle);
	vec3        n   = axis_in;
	const float len = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
	if (len > 0.0f)
		n.x /= len, n.y /= len, n.z /= len;
	const vec3 t{(1 - c) * n.x, (1 - c) * n.y, (1 - c) * n.z};
	mat4       r;
	r.at(0, 0) = c + t.x * n.x, r.at(1, 0) = t.x * n.y + s * n.z, r.at(2, 0) = t.x * n.z - s * n.y;
	r.at(0, 1) = t.y * n.x - s * n.z, r.at(1, 1) = c + t.y * n.y, r.at(2, 1) = t.y * n.z + s * n.x;
	r.at(0, 2) = t.z * n.x + s * n.y, r.at(1, 2) = t.z * n.y - s * n.x, r.at(2, 2) = c + t.z * n.z;
	return r;
}

// glm::lookAtRH
inline mat4 look_at(const vec3 &eye, const vec3 &centre, const vec3 &up)
{
	auto norm = [](vec3 v) {
		const float l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
		return vec3{v.x / l, v.y / l, v.z / l};
	};
	auto cross = [](const vec3 &a, const vec3 &b) { return vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
	auto dot   = [](const vec3 &a, const vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
	const vec3 f = norm(vec3{centre.x - eye.x, centre.y - eye.y, centre.z - eye.z});
	const vec3 s = norm(cross(f, up));
	const vec3 u = cross(s, f);
	mat4       r;
	r.at(0, 0) = s.x, r.at(0, 1) = s.y, r.at(0, 2) = s.z, r.at(0, 3) = -dot(s, eye);
	r.at(1, 0) = u.x, r.at(1, 1) = u.y, r.at(1, 2) = u.z, r.at(1, 3) = -dot(u, eye);
	r.at(2, 0) = -f.x, r.at(2, 1) = -f.y, r.at(2, 2) = -f.z, r.at(2, 3) = dot(f, eye);
	return r;
}

// vkb::PerspectiveCamera::get_projection (reverse-Z: perspectiveRH_ZO with near/far swapped) followed by
// vkb::vulkan_style_projection (y flipped); used at src/volume_render_subpass.cpp:224.
inline mat4 perspective_vulkan(float fov_y_rad, float aspect, float near_plane, float far_plane)
{
	const float t  = std::tan(fov_y_rad / 2.0f);
	const float zn = far_plane, zf = near_plane;
	mat4        r;
	for (float &v : r.m)
		v = 0.0f;
	r.at(0, 0) = 1.0f / (aspect * t);
	r.at(1, 1) = -1.0f / t;
	r.at(2, 2) = zf / (zn - zf);
	r.at(3, 2) = -1.0f;
	r.at(2, 3) = -(zf * zn) / (zf - zn);
	return r;
}

}        // namespace vkv
