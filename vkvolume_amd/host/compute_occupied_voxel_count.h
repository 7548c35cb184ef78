// compute_occupied_voxel_count.h — ComputeOccupiedVoxelCount (reference: src/compute_occupied_voxel_count.h:32-50),
// the benchmark-mode statistic "percentage of voxels with transfer-function alpha > 0".
#pragma once

#include "volume_component.h"

class ComputeOccupiedVoxelCount
{
  public:
	explicit ComputeOccupiedVoxelCount(DeviceContext &device_context) : dc(device_context) {}
	virtual ~ComputeOccupiedVoxelCount();

	// The reference sizes a buffer of one partial per subgroup (initialise_buffer) and tree-reduces it in place; here the
	// whole reduction is one kernel and the "buffer" is a single device uint64.
	uint64_t *initialise_buffer(Volume &volume);
	void      compute(Volume &volume, uint64_t *buffer, const TransferFunctionUniform &transfer_function_uniform);
	uint64_t  get_result(uint64_t *buffer) const;        // blocks until the count is on the host

  private:
	DeviceContext &dc;
	uint64_t *     owned = nullptr;
};
