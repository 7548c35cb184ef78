// load_volume.h — LoadVolume, the reference's raw-volume loader (src/load_volume.h:26-47) without Vulkan / glm / Boost.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/vkvolume_amd.h"
#include "vkv_math.hpp"

class LoadVolume
{
  public:
	// src/load_volume.h:29-39 (VkExtent3D -> VkvExtent3D, glm -> vkv::).  tf_range / alpha_factor are declared by the
	// reference but never filled (src/load_volume.cpp fills neither); kept for source compatibility.
	struct Header
	{
		VkvExtent3D extent{0, 0, 0};
		vkv::vec3   voxel_size;
		float       normalisation_range[2] = {0.0f, 0.0f};
		std::string type;
		std::string endianness;
		vkv::mat4   image_transform;
		float       tf_range[2] = {0.0f, 0.0f};
		float       alpha_factor = 0.0f;
	};

	// Parses the 5-line `<volume>.header` side-car (format: README.md:58-70).  Throws std::runtime_error("Failed to open header file").
	static Header load_header(std::string filename_header);

	// Reads the dense raw file, checks its size, converts endianness and normalises to uint8 with truncation
	// (src/load_volume.cpp:112-172).  Throws std::runtime_error with the reference's messages.
	static std::vector<uint8_t> load_data(std::string filename_data, const Header &header);

	// The raw file bytes, size-checked against the header (same errors as load_data); normalisation is then done on the device
	// by vkv_convert_volume (Volume::load_from_file).  Not in the reference: its loader converts on the CPU.
	static std::vector<uint8_t> load_raw(std::string filename_data, const Header &header);
	static int                  voxel_type(const Header &header);        // VkvVoxelType, or throws "unsupported image data type"

  private:
	template <typename T>
	static std::vector<uint8_t> load_data_impl(std::string filename_data, const Header &header);
};
