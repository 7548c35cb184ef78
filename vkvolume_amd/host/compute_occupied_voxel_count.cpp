#include "compute_occupied_voxel_count.h"

#include <hip/hip_runtime_api.h>

#include <stdexcept>

ComputeOccupiedVoxelCount::~ComputeOccupiedVoxelCount()
{
	if (owned)
		(void) hipFree(owned);
}

uint64_t *ComputeOccupiedVoxelCount::initialise_buffer(Volume &)
{
	if (!owned && hipMalloc((void **) &owned, sizeof(uint64_t)) != hipSuccess)
		throw std::runtime_error("ComputeOccupiedVoxelCount: hipMalloc failed");
	return owned;
}

void ComputeOccupiedVoxelCount::compute(Volume &volume, uint64_t *buffer, const TransferFunctionUniform &tf)
{
	const auto &vol = volume.get_volume();
	if (vkv_occupied_voxel_count(dc.ctx, vol.data, volume.options.use_precomputed_gradient ? volume.get_gradient().data : nullptr, &tf, vol.extent, buffer,
	                             dc.stream) != VKV_OK)
		throw std::runtime_error(std::string("ComputeOccupiedVoxelCount: ") + vkv_last_error(dc.ctx));
}

uint64_t ComputeOccupiedVoxelCount::get_result(uint64_t *buffer) const
{
	uint64_t count = 0;
	if (hipMemcpyAsync(&count, buffer, sizeof(count), hipMemcpyDeviceToHost, (hipStream_t) dc.stream) != hipSuccess ||
	    hipStreamSynchronize((hipStream_t) dc.stream) != hipSuccess)
		throw std::runtime_error("ComputeOccupiedVoxelCount: read-back failed");
	return count;
}
