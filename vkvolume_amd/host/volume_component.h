// volume_component.h — Volume, the reference's scene component that owns every per-volume GPU resource
// (src/volume_component.h:31-93).  Vulkan images become linear HIP device buffers; the RenderContext / CommandBuffer
// parameters become a DeviceContext (vkv context + HIP stream).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/vkvolume_amd.h"
#include "transfer_function.h"
#include "vkv_math.hpp"

// Stand-in for vkb::RenderContext& / vkb::CommandBuffer& on this path: where to enqueue.
struct DeviceContext
{
	vkv_ctx *ctx    = nullptr;
	void *   stream = nullptr;        // hipStream_t
};

class Volume
{
  public:
	explicit Volume(const std::string &name);
	~Volume();
	Volume(const Volume &) = delete;
	Volume &operator=(const Volume &) = delete;

	// src/volume_component.cpp:55-153: header + data from disk, device buffers, volume upload.  Always returns true (errors throw).
	bool load_from_file(DeviceContext &dc, std::string filename, uint32_t distance_map_block_size = 4);
	// Same resource set-up from voxels already in host memory / generated on the device (offscreen driver, tests).
	bool load_from_memory(DeviceContext &dc, const uint8_t *voxels, VkvExtent3D extent, uint32_t distance_map_block_size = 4);
	bool load_synthetic(DeviceContext &dc, VkvExtent3D extent, uint32_t kind, uint32_t seed, uint32_t distance_map_block_size = 4);

	void set_image_transform(const vkv::mat4 &mat);
	void set_number_of_distance_maps(DeviceContext &dc, size_t n);        // grow-only (src/volume_component.cpp:155-184)

	// Volume::Options, src/volume_component.h:45-56 (same defaults)
	struct Options
	{
		float sampling_factor          = 1.0f;
		float voxel_alpha_factor       = 1.0f;
		bool  use_precomputed_gradient = true;
		float intensity_min            = 0.0f;
		float intensity_max            = 1.0f;
		float gradient_min             = 0.0f;
		float gradient_max             = 1.0f;
	} options;

	// Volume::Image: a device buffer instead of image + view + sampler
	struct Image
	{
		uint8_t *   data = nullptr;
		VkvExtent3D extent{0, 0, 0};
		uint32_t    bytes_per_texel = 1;
		size_t      size_bytes() const { return (size_t) extent.width * extent.height * extent.depth * bytes_per_texel; }
	};

	const Image &get_volume() const { return volume; }
	const Image &get_gradient() const { return gradient; }
	const Image &get_transfer_function() const { return transfer_function; }
	const Image &get_distance_map(size_t idx = 0) const { return distance_maps.at(idx); }
	const Image &get_distance_map_swap() const { return distance_map_swap; }
	size_t       get_number_of_distance_maps() const { return distance_maps.size(); }
	vkv::mat4 &  get_image_transform() { return image_transform; }
	const std::string &get_name() const { return name; }

	TransferFunctionUniform get_transfer_function_uniform();                 // src/volume_component.cpp:226-240
	void                    update_transfer_function_texture(DeviceContext &dc);        // src/volume_component.cpp:242-278

	// vkb::sg::Node stand-in: the node's world matrix (benchmark mode rescales it, src/volume_render.cpp:224-238)
	vkv::mat4 node_transform;

	// device-internal accelerators of the ray-marcher (see include/vkvolume_amd.h)
	void            pack(DeviceContext &dc);
	const void *    get_packed_volume() const { return packed; }
	const uint32_t *get_transfer_function_bits() const { return transfer_function_bits; }

  private:
	void allocate(DeviceContext &dc, VkvExtent3D extent, uint32_t block);
	void release();

	std::string        name;
	Image              volume, gradient, transfer_function;
	std::vector<Image> distance_maps;
	Image              distance_map_swap;
	void *             packed                 = nullptr;
	size_t             packed_bytes           = 0;
	uint32_t *         transfer_function_bits = nullptr;
	vkv::mat4          image_transform;
};
