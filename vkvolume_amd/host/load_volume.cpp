// load_volume.cpp — see load_volume.h.  Behaviour follows src/load_volume.cpp; the implementation is new
// (stdio-free streaming read, explicit byte swap instead of Boost.Endian, vkv:: math instead of glm).
#include "load_volume.h"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>
#include <type_traits>

namespace
{

// one "<numbers> # comment" line of the header
std::istringstream next_line(std::ifstream &file)
{
	std::string line;
	std::getline(file, line);
	return std::istringstream(line);
}

template <typename T>
T from_file_order(T v, bool big_endian)
{
	if (sizeof(T) == 1)
		return v;
	const uint16_t probe         = 1;
	const bool     host_is_little = *reinterpret_cast<const uint8_t *>(&probe) == 1;
	if (big_endian != host_is_little)
		return v;        // file order == host order
	unsigned char b[sizeof(T)];
	std::memcpy(b, &v, sizeof(T));
	std::reverse(b, b + sizeof(T));
	std::memcpy(&v, b, sizeof(T));
	return v;
}

}        // namespace

LoadVolume::Header LoadVolume::load_header(std::string filename_header)
{
	std::ifstream file(filename_header);
	if (!file.is_open())
		throw std::runtime_error("Failed to open header file");

	Header header;
	// line 1: extents; 2: voxel size; 3: normalisation range; 4: type + endianness; 5: rotation axis + angle (degrees)
	next_line(file) >> header.extent.width >> header.extent.height >> header.extent.depth;
	next_line(file) >> header.voxel_size.x >> header.voxel_size.y >> header.voxel_size.z;
	next_line(file) >> header.normalisation_range[0] >> header.normalisation_range[1];
	next_line(file) >> header.type >> header.endianness;
	vkv::vec3 axis;
	float     angle_deg = 0.0f;
	next_line(file) >> axis.x >> axis.y >> axis.z >> angle_deg;

	// image_transform = rotate(angle, axis) * scale(voxel_size * extent)   (src/load_volume.cpp:82-83)
	const vkv::vec3 physical_size{header.voxel_size.x * (float) header.extent.width, header.voxel_size.y * (float) header.extent.height,
	                              header.voxel_size.z * (float) header.extent.depth};
	header.image_transform = vkv::rotate(vkv::radians(angle_deg), axis) * vkv::scale(physical_size);
	return header;
}

std::vector<uint8_t> LoadVolume::load_data(std::string filename_data, const Header &header)
{
	if (header.type == "uint8_t")
		return load_data_impl<uint8_t>(filename_data, header);
	if (header.type == "int8_t")
		return load_data_impl<int8_t>(filename_data, header);
	if (header.type == "uint16_t")
		return load_data_impl<uint16_t>(filename_data, header);
	if (header.type == "int16_t")
		return load_data_impl<int16_t>(filename_data, header);
	throw std::runtime_error("unsupported image data type");
}

int LoadVolume::voxel_type(const Header &header)
{
	if (header.type == "uint8_t") return VKV_VOXEL_UINT8;
	if (header.type == "int8_t") return VKV_VOXEL_INT8;
	if (header.type == "uint16_t") return VKV_VOXEL_UINT16;
	if (header.type == "int16_t") return VKV_VOXEL_INT16;
	throw std::runtime_error("unsupported image data type");
}

std::vector<uint8_t> LoadVolume::load_raw(std::string filename_data, const Header &header)
{
	const int    type      = voxel_type(header);
	const size_t n_voxels  = (size_t) header.extent.width * (size_t) header.extent.height * (size_t) header.extent.depth;
	const size_t file_size = n_voxels * ((type == VKV_VOXEL_UINT16 || type == VKV_VOXEL_INT16) ? 2 : 1);
	std::ifstream file(filename_data, std::ios::binary | std::ios::ate);
	if (!file.is_open())
		throw std::runtime_error("Failed to open data file");
	if ((size_t) file.tellg() != file_size)
		throw std::runtime_error("File size does not match expected size for the given image format/dimensions");
	file.seekg(0, std::ios::beg);
	std::vector<uint8_t> raw(file_size);
	file.read(reinterpret_cast<char *>(raw.data()), (std::streamsize) file_size);
	if (!file)
		throw std::runtime_error("File error");
	return raw;
}

template <typename T>
std::vector<uint8_t> LoadVolume::load_data_impl(std::string filename_data, const Header &header)
{
	const size_t n_voxels  = (size_t) header.extent.width * (size_t) header.extent.height * (size_t) header.extent.depth;
	const size_t file_size = n_voxels * sizeof(T);

	std::ifstream file(filename_data, std::ios::binary | std::ios::ate);
	if (!file.is_open())
		throw std::runtime_error("Failed to open data file");
	if ((size_t) file.tellg() != file_size)
		throw std::runtime_error("File size does not match expected size for the given image format/dimensions");
	file.seekg(0, std::ios::beg);

	// Stream the file in 64 MiB pieces and convert each piece straight to uint8: the raw T-typed volume (up to 2x the
	// output size) is never held in memory as a whole.
	const bool  big = header.endianness == "big";        // anything else is read as little endian, like the reference
	const float lo = header.normalisation_range[0], hi = header.normalisation_range[1];
	std::vector<uint8_t> out(n_voxels);
	const size_t         piece = (size_t(64) << 20) / sizeof(T);
	std::vector<T>       buf(std::min(piece, std::max<size_t>(n_voxels, 1)));
	for (size_t done = 0; done < n_voxels;)
	{
		const size_t n = std::min(piece, n_voxels - done);
		file.read(reinterpret_cast<char *>(buf.data()), (std::streamsize) (n * sizeof(T)));
		if (!file)
			throw std::runtime_error("File error");
		for (size_t i = 0; i < n; ++i)
		{
			const float v = static_cast<float>(from_file_order(buf[i], big));
			// (uint8) (255 * clamp((v - min) / (max - min), 0, 1)) — truncating (src/load_volume.cpp:165-169)
			const float t = std::max(0.0f, std::min(1.0f, (v - lo) / (hi - lo)));
			out[done + i] = static_cast<uint8_t>(std::numeric_limits<uint8_t>::max() * t);
		}
		done += n;
	}
	return out;
}
