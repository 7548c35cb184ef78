// host_sanitize_driver.cpp — the pure-CPU parts of the host mirror (vkvolume_amd/host/load_volume.cpp, vkv_math.hpp) under
// AddressSanitizer + UndefinedBehaviorSanitizer.  Built and run by tests/test_sanitizers_cpu.py (CPU only, never on the GPU box):
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined host_sanitize_driver.cpp ../vkvolume_amd/host/load_volume.cpp
// usage: host_sanitize_driver <scratch directory>; exit status 0 = every check passed (a sanitizer report aborts the process).
// Expected values are worked out here from the reference's formulas (src/load_volume.cpp:82-83, 165-169), not by calling the code under test.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../vkvolume_amd/host/load_volume.h"

static int g_failed = 0;
#define CHECK(cond)                                                           \
	do                                                                        \
	{                                                                         \
		if (!(cond))                                                          \
		{                                                                     \
			std::fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); \
			++g_failed;                                                       \
		}                                                                     \
	} while (0)

template <typename T>
static void write_volume(const std::string &path, const std::vector<T> &host_values, bool big)
{
	std::ofstream f(path, std::ios::binary);
	for (T v : host_values)
	{
		unsigned char b[sizeof(T)];
		std::memcpy(b, &v, sizeof(T));        // this host is little endian (checked in main)
		if (big)
			for (size_t i = 0; i < sizeof(T) / 2; ++i)
				std::swap(b[i], b[sizeof(T) - 1 - i]);
		f.write(reinterpret_cast<const char *>(b), sizeof(T));
	}
}

template <typename T>
static void loader_case(const std::string &dir, const char *type, bool big, float lo, float hi)
{
	const uint32_t W = 9, H = 5, D = 7;
	std::vector<T> values((size_t) W * H * D);
	uint32_t       state = 12345u + (uint32_t) sizeof(T) * 77u + (big ? 1u : 0u);
	for (auto &v : values)
	{
		state = state * 1664525u + 1013904223u;
		v     = (T) (state >> 13);        // wraps into the type's range, extremes included over 315 draws
	}
	values[0] = std::numeric_limits<T>::min(), values[1] = std::numeric_limits<T>::max();
	const std::string path = dir + "/vol_" + type + (big ? "_big" : "_little") + ".raw";
	write_volume(path, values, big);
	{
		std::ofstream h(path + ".header");
		h << W << " " << H << " " << D << " # extents\n0.0003 0.0003 0.0007 # voxel size\n" << lo << " " << hi << " # range\n" << type << " " << (big ? "big" : "little")
		  << " # type\n1 0 0 90 # rotation\n";
	}
	const LoadVolume::Header header = LoadVolume::load_header(path + ".header");
	CHECK(header.extent.width == W && header.extent.height == H && header.extent.depth == D);
	CHECK(header.type == type && header.endianness == (big ? "big" : "little"));
	CHECK(header.normalisation_range[0] == lo && header.normalisation_range[1] == hi);
	// image_transform = rotate(90 degrees about x) * scale(voxel_size * extent): column 0 = (sx, 0, 0), column 1 = (0, 0, sy), column 2 = (0, -sz, 0)
	const float sx = 0.0003f * (float) W, sy = 0.0003f * (float) H, sz = 0.0007f * (float) D;
	CHECK(std::fabs(header.image_transform.at(0, 0) - sx) < 1e-7f && std::fabs(header.image_transform.at(2, 1) - sy) < 1e-7f);
	CHECK(std::fabs(header.image_transform.at(1, 2) + sz) < 1e-7f && std::fabs(header.image_transform.at(1, 1)) < 1e-7f);
	const std::vector<uint8_t> data = LoadVolume::load_data(path, header);
	CHECK(data.size() == values.size());
	size_t bad = 0;
	for (size_t i = 0; i < values.size() && i < data.size(); ++i)
	{
		const float t = std::fmax(0.0f, std::fmin(1.0f, ((float) values[i] - lo) / (hi - lo)));
		bad += data[i] != (uint8_t) (255.0f * t);
	}
	CHECK(bad == 0);
	const std::vector<uint8_t> raw = LoadVolume::load_raw(path, header);
	CHECK(raw.size() == values.size() * sizeof(T));
}

template <typename F>
static bool throws(F f, const char *message)
{
	try
	{
		f();
	}
	catch (const std::runtime_error &e)
	{
		return std::strstr(e.what(), message) != nullptr;
	}
	return false;
}

int main(int argc, char **argv)
{
	if (argc < 2)
		return 2;
	const std::string dir = argv[1];
	const uint16_t    probe = 1;
	if (*reinterpret_cast<const uint8_t *>(&probe) != 1)
		return 3;        // big-endian host: write_volume would have to swap the other way

	loader_case<uint8_t>(dir, "uint8_t", false, 10.0f, 200.0f);
	loader_case<int8_t>(dir, "int8_t", true, -100.0f, 100.0f);
	loader_case<uint16_t>(dir, "uint16_t", false, 400.0f, 25380.0f);
	loader_case<uint16_t>(dir, "uint16_t", true, 400.0f, 25380.0f);
	loader_case<int16_t>(dir, "int16_t", false, -3000.0f, 12000.0f);
	loader_case<int16_t>(dir, "int16_t", true, -3000.0f, 12000.0f);
	loader_case<uint8_t>(dir, "uint8_t", true, 50.0f, 50.0f);        // degenerate range: (v - lo) / 0 = +-inf or NaN, clamped before the cast

	// error paths keep the reference's messages (src/load_volume.cpp:38,108,123,130,143)
	CHECK(throws([&] { LoadVolume::load_header(dir + "/nope.header"); }, "Failed to open header file"));
	{
		std::ofstream(dir + "/short.raw", std::ios::binary) << "0123456789A";
		std::ofstream(dir + "/short.raw.header") << "9 5 7\n1 1 1\n0 255\nuint8_t little\n1 0 0 0\n";
		const LoadVolume::Header h = LoadVolume::load_header(dir + "/short.raw.header");
		CHECK(throws([&] { LoadVolume::load_data(dir + "/short.raw", h); }, "File size does not match"));
		CHECK(throws([&] { LoadVolume::load_raw(dir + "/short.raw", h); }, "File size does not match"));
		CHECK(throws([&] { LoadVolume::load_data(dir + "/missing.raw", h); }, "Failed to open data file"));
	}
	{
		std::ofstream(dir + "/f32.raw", std::ios::binary) << "0123";
		std::ofstream(dir + "/f32.raw.header") << "1 1 1\n1 1 1\n0 255\nfloat little\n1 0 0 0\n";
		const LoadVolume::Header h = LoadVolume::load_header(dir + "/f32.raw.header");
		CHECK(throws([&] { LoadVolume::load_data(dir + "/f32.raw", h); }, "unsupported image data type"));
		CHECK(throws([&] { LoadVolume::voxel_type(h); }, "unsupported image data type"));
	}
	{        // a truncated header: missing lines leave the defaults, nothing is read out of bounds
		std::ofstream(dir + "/cut.header") << "4 4\n";
		const LoadVolume::Header h = LoadVolume::load_header(dir + "/cut.header");
		CHECK(h.extent.width == 4 && h.extent.height == 4 && h.extent.depth == 0 && h.type.empty());
	}

	// vkv_math.hpp: the glm operations of src/volume_render_subpass.cpp:226-239
	{
		const vkv::mat4 m = vkv::translate(vkv::vec3{1.0f, -2.0f, 3.0f}) * vkv::rotate(vkv::radians(37.0f), vkv::vec3{0.3f, -0.5f, 0.8f}) * vkv::scale(vkv::vec3{2.0f, 0.5f, 4.0f});
		const vkv::mat4 p = m * vkv::inverse(m);
		for (int r = 0; r < 4; ++r)
			for (int c = 0; c < 4; ++c)
				CHECK(std::fabs(p.at(r, c) - (r == c ? 1.0f : 0.0f)) < 1e-5f);
		const vkv::mat4 it = vkv::inverse_transpose(m), ti = vkv::transpose(vkv::inverse(m));
		CHECK(std::memcmp(it.m, ti.m, sizeof(it.m)) == 0);
		const vkv::vec4 v = m * vkv::vec4{1.0f, 1.0f, 1.0f, 1.0f};
		CHECK(v.w == 1.0f);
		float singular[16] = {0}, out[16];
		CHECK(!vkv::invert4x4(singular, out));        // reported, not divided by
		const vkv::mat4 view = vkv::look_at(vkv::vec3{0, 0, 5}, vkv::vec3{0, 0, 0}, vkv::vec3{0, 1, 0});
		const vkv::vec4 origin = view * vkv::vec4{0, 0, 0, 1};
		CHECK(std::fabs(origin.z + 5.0f) < 1e-6f);
		const vkv::mat4 proj = vkv::perspective_vulkan(vkv::radians(60.0f), 16.0f / 9.0f, 0.1f, 100.0f);
		CHECK(std::isfinite(proj.at(0, 0)) && std::isfinite(proj.at(2, 2)));
		const vkv::mat4 r0 = vkv::rotate(0.5f, vkv::vec3{0, 0, 0});        // zero axis (header line "0 0 0 a"): must not produce NaN-indexed anything
		(void) r0;
	}
	if (g_failed)
		std::fprintf(stderr, "%d check(s) failed\n", g_failed);
	else
		std::printf("host sanitize driver: ok\n");
	return g_failed ? 1 : 0;
}
