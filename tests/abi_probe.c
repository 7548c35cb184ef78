/* prints sizeof / offsetof of the public structs so the ctypes mirror (vkvolume_amd/abi.py) can be checked against the C header */
#include <stddef.h>
#include <stdio.h>
#include "../include/vkvolume_amd.h"
#define S(T) printf("sizeof %s %zu\n", #T, sizeof(T))
#define O(T, f) printf("offsetof %s.%s %zu\n", #T, #f, offsetof(T, f))
int main(void)
{
	S(VkvExtent3D); S(VkvTransferFunctionUniform); S(VkvVolumeOptions); S(VkvCameraUniform); S(VkvRayCastUniform); S(VkvRayGen);
	S(VkvRenderOptions); S(VkvTileRect); S(VkvTileSchedule); S(VkvRenderParams); S(VkvTuning);
	O(VkvTuning, feedback_period); O(VkvTuning, clamp_always); O(VkvTuning, wave_shape); O(VkvTuning, occupancy_kernel); O(VkvTuning, gradient_segment); O(VkvTuning, arena_bytes);
	O(VkvTileSchedule, compact); O(VkvTileSchedule, rect); O(VkvTileSchedule, fill_outside); O(VkvTileRect, w);
	O(VkvRenderParams, ray_cast); O(VkvRenderParams, transfer_function); O(VkvRenderParams, ray_gen); O(VkvRenderParams, options);
	O(VkvRenderParams, use_precomputed_gradient); O(VkvRenderParams, image_width); O(VkvRenderParams, tiles);
	O(VkvRenderParams, volume_extent); O(VkvRenderParams, map_extent); O(VkvRenderParams, d_volume); O(VkvRenderParams, d_gradient);
	O(VkvRenderParams, d_transfer_function); O(VkvRenderParams, d_distance_maps); O(VkvRenderParams, d_packed_volume); O(VkvRenderParams, d_transfer_function_bits); O(VkvRenderParams, d_out_color);
	O(VkvRenderParams, d_out_rgba8); O(VkvRenderParams, d_out_counts); O(VkvRenderParams, d_out_depth); O(VkvRenderParams, d_in_depth); O(VkvRenderParams, blend_over_target);
	return 0;
}
