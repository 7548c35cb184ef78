#include "compute_distance_map.h"

#include <stdexcept>

void ComputeDistanceMap::compute(Volume &volume, const TransferFunctionUniform &tf, VolumeRenderSubpass::SkippingType skipping_type)
{
	const bool anisotropic     = skipping_type == VolumeRenderSubpass::SkippingType::AnisotropicDistance;
	const int  n_distance_maps = anisotropic ? 8 : 1;
	volume.set_number_of_distance_maps(dc, n_distance_maps);

	uint8_t *maps[8] = {nullptr};
	for (size_t i = 0; i < volume.get_number_of_distance_maps() && i < 8; ++i)
		maps[i] = volume.get_distance_map(i).data;
	const auto &vol = volume.get_volume();
	const int   rc  = vkv_compute_distance_map(dc.ctx, vol.data, volume.options.use_precomputed_gradient ? volume.get_gradient().data : nullptr,
                                            volume.get_transfer_function().data, &tf, vol.extent, maps, volume.get_distance_map_swap().data,
                                            volume.get_distance_map_swap().extent, static_cast<int32_t>(skipping_type), dc.stream);
	if (rc != VKV_OK)
		throw std::runtime_error(std::string("ComputeDistanceMap: ") + vkv_last_error(dc.ctx));
}
