// volume_component.cpp — see volume_component.h.
#include "volume_component.h"

#include <hip/hip_runtime_api.h>

#include <stdexcept>

#include "load_volume.h"

namespace
{
void hip_check(hipError_t e, const char *what)
{
	if (e != hipSuccess)
		throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void vkv_check(DeviceContext &dc, int rc, const char *what)
{
	if (rc != VKV_OK)
		throw std::runtime_error(std::string(what) + ": " + vkv_last_error(dc.ctx));
}
uint8_t *device_alloc(size_t bytes)
{
	void *p = nullptr;
	hip_check(hipMalloc(&p, bytes ? bytes : 1), "hipMalloc");
	return static_cast<uint8_t *>(p);
}
}        // namespace

Volume::Volume(const std::string &name_) : name(name_) {}

Volume::~Volume() { release(); }

void Volume::release()
{
	for (Image *im : {&volume, &gradient, &transfer_function, &distance_map_swap})
	{
		if (im->data)
			(void) hipFree(im->data);
		im->data = nullptr;
	}
	for (Image &m : distance_maps)
		if (m.data)
			(void) hipFree(m.data);
	distance_maps.clear();
	if (packed)
		(void) hipFree(packed);
	if (transfer_function_bits)
		(void) hipFree(transfer_function_bits);
	packed                 = nullptr;
	packed_bytes           = 0;
	transfer_function_bits = nullptr;
}

void Volume::allocate(DeviceContext &dc, VkvExtent3D extent, uint32_t block)
{
	(void) dc;
	release();
	if (block == 0)
		throw std::runtime_error("distance_map_block_size must be positive");
	volume.extent = extent;
	volume.data   = device_alloc(volume.size_bytes());
	if (options.use_precomputed_gradient)
	{
		gradient.extent = extent;
		gradient.data   = device_alloc(gradient.size_bytes());
	}
	transfer_function.extent          = VkvExtent3D{256, 256, 1};        // R8G8B8A8_UNORM 256x256 (src/volume_component.cpp:68-74)
	transfer_function.bytes_per_texel = 4;
	transfer_function.data            = device_alloc(transfer_function.size_bytes());
	transfer_function_bits            = reinterpret_cast<uint32_t *>(device_alloc(VKV_TF_BITS_WORDS * sizeof(uint32_t)));
	auto rnd_up                       = [](uint32_t x, uint32_t y) { return (x + y - 1) / y; };
	distance_map_swap.extent          = VkvExtent3D{rnd_up(extent.width, block), rnd_up(extent.height, block), rnd_up(extent.depth, block)};
	distance_map_swap.data            = device_alloc(distance_map_swap.size_bytes());
}

bool Volume::load_from_file(DeviceContext &dc, std::string filename, uint32_t distance_map_block_size)
{
	auto header = LoadVolume::load_header(filename + ".header");
	set_image_transform(header.image_transform);
	// the raw file goes to the device as it is; endianness + normalisation to uint8 run there (vkv_convert_volume), at HBM speed
	const int            type = LoadVolume::voxel_type(header);
	std::vector<uint8_t> raw  = LoadVolume::load_raw(filename, header);
	allocate(dc, header.extent, distance_map_block_size);
	uint8_t *d_raw = device_alloc(raw.size());
	try
	{
		hip_check(hipMemcpyAsync(d_raw, raw.data(), raw.size(), hipMemcpyHostToDevice, (hipStream_t) dc.stream), "raw volume upload");
		vkv_check(dc, vkv_convert_volume(dc.ctx, d_raw, type, header.endianness == "big", header.normalisation_range[0], header.normalisation_range[1],
		                                 (uint64_t) header.extent.width * header.extent.height * header.extent.depth, volume.data, dc.stream),
		          "volume conversion");
		hip_check(hipStreamSynchronize((hipStream_t) dc.stream), "volume upload");
	}
	catch (...)
	{
		(void) hipFree(d_raw);
		throw;
	}
	(void) hipFree(d_raw);
	return true;
}

bool Volume::load_from_memory(DeviceContext &dc, const uint8_t *voxels, VkvExtent3D extent, uint32_t distance_map_block_size)
{
	allocate(dc, extent, distance_map_block_size);
	hip_check(hipMemcpyAsync(volume.data, voxels, volume.size_bytes(), hipMemcpyHostToDevice, (hipStream_t) dc.stream), "volume upload");
	hip_check(hipStreamSynchronize((hipStream_t) dc.stream), "volume upload");        // the staging buffer is the caller's: wait like the reference's fence
	return true;
}

bool Volume::load_synthetic(DeviceContext &dc, VkvExtent3D extent, uint32_t kind, uint32_t seed, uint32_t distance_map_block_size)
{
	allocate(dc, extent, distance_map_block_size);
	vkv_check(dc, vkv_synth_volume(dc.ctx, volume.data, extent, kind, seed, dc.stream), "synthetic volume");
	return true;
}

void Volume::set_image_transform(const vkv::mat4 &mat) { image_transform = mat; }

void Volume::set_number_of_distance_maps(DeviceContext &dc, size_t n)
{
	(void) dc;
	if (n <= distance_maps.size())
		return;
	// the reference re-creates every map when growing (src/volume_component.cpp:160-183)
	for (Image &m : distance_maps)
		if (m.data)
			(void) hipFree(m.data);
	distance_maps.assign(n, Image{});
	for (Image &m : distance_maps)
	{
		m.extent = distance_map_swap.extent;
		m.data   = device_alloc(m.size_bytes());
	}
}

TransferFunctionUniform Volume::get_transfer_function_uniform()
{
	VkvVolumeOptions o{options.sampling_factor, options.voxel_alpha_factor, options.use_precomputed_gradient ? 1u : 0u,
	                   options.intensity_min,   options.intensity_max,      options.gradient_min, options.gradient_max};
	TransferFunctionUniform u;
	vkv_transfer_function_uniform(&o, &u);
	return u;
}

void Volume::update_transfer_function_texture(DeviceContext &dc)
{
	VkvVolumeOptions     o{options.sampling_factor, options.voxel_alpha_factor, options.use_precomputed_gradient ? 1u : 0u,
                       options.intensity_min,   options.intensity_max,      options.gradient_min, options.gradient_max};
	std::vector<uint8_t> tex(256 * 256 * 4);        // the CPU builds the LUT, like the reference
	vkv_transfer_function_texture(&o, tex.data());
	hip_check(hipMemcpyAsync(transfer_function.data, tex.data(), tex.size(), hipMemcpyHostToDevice, (hipStream_t) dc.stream), "TF upload");
	hip_check(hipStreamSynchronize((hipStream_t) dc.stream), "TF upload");
	const TransferFunctionUniform tf = get_transfer_function_uniform();
	vkv_check(dc, vkv_transfer_function_tables(dc.ctx, transfer_function.data, &tf, transfer_function_bits, dc.stream), "TF tables");
}

void Volume::pack(DeviceContext &dc)
{
	const size_t need = vkv_packed_volume_bytes(volume.extent);
	if (need != packed_bytes)
	{
		if (packed)
			(void) hipFree(packed);
		packed       = device_alloc(need);
		packed_bytes = need;
	}
	vkv_check(dc, vkv_pack_volume(dc.ctx, volume.data, options.use_precomputed_gradient ? gradient.data : nullptr, volume.extent, packed, dc.stream),
	          "pack volume");
}
