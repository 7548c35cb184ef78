"""ctypes mirror of ``include/vkvolume_amd.h`` (struct layouts and enums only; loads no library).

Each class cites the header struct it mirrors; the header cites the reference file:line.
"""
import ctypes as C

VKV_OK = 0
VKV_E_INVALID_ARGUMENT = -1
VKV_E_UNSUPPORTED = -2
VKV_E_NO_DEVICE = -3
VKV_E_IO = -4
TF_BITS_WORDS = 2564  # VKV_TF_BITS_WORDS
MAX_BATCH = 32  # VKV_MAX_BATCH

# VolumeRenderSubpass::SkippingType / Test (src/volume_render_subpass.h:58-72)
SKIP_NONE, SKIP_BLOCK, SKIP_DISTANCE, SKIP_ANISOTROPIC_DISTANCE = 0, 1, 2, 3
VOXEL_TYPES = {"uint8_t": 0, "int8_t": 1, "uint16_t": 2, "int16_t": 3}  # VkvVoxelType by LoadVolume::Header::type
TEST_NONE, TEST_RAY_ENTRY, TEST_RAY_EXIT, TEST_NUM_TEXTURE_SAMPLES = 0, 1, 2, 3


class Extent3D(C.Structure):
    """VkvExtent3D"""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("depth", C.c_uint32)]

    def __init__(self, width=0, height=0, depth=0):
        super().__init__(int(width), int(height), int(depth))

    def as_tuple(self):
        return (self.width, self.height, self.depth)

    @property
    def count(self):
        return self.width * self.height * self.depth


class TransferFunctionUniform(C.Structure):
    """VkvTransferFunctionUniform (src/transfer_function.h:20-32)"""
    _fields_ = [("sampling_factor", C.c_float), ("voxel_alpha_factor", C.c_float),
                ("grad_magnitude_modifier", C.c_float), ("use_gradient", C.c_uint32),
                ("intensity_min", C.c_float), ("intensity_range_inv", C.c_float),
                ("gradient_min", C.c_float), ("gradient_range_inv", C.c_float)]


class VolumeOptions(C.Structure):
    """VkvVolumeOptions (Volume::Options, src/volume_component.h:45-56)"""
    _fields_ = [("sampling_factor", C.c_float), ("voxel_alpha_factor", C.c_float),
                ("use_precomputed_gradient", C.c_uint32), ("intensity_min", C.c_float),
                ("intensity_max", C.c_float), ("gradient_min", C.c_float), ("gradient_max", C.c_float)]

    def __init__(self, sampling_factor=1.0, voxel_alpha_factor=1.0, use_precomputed_gradient=True,
                 intensity_min=0.0, intensity_max=1.0, gradient_min=0.0, gradient_max=1.0):
        super().__init__(sampling_factor, voxel_alpha_factor, 1 if use_precomputed_gradient else 0,
                         intensity_min, intensity_max, gradient_min, gradient_max)


class CameraUniform(C.Structure):
    """VkvCameraUniform (src/volume_render_subpass.h:32-39)"""
    _fields_ = [("camera_view", C.c_float * 16), ("camera_proj", C.c_float * 16),
                ("camera_view_proj_inv", C.c_float * 16), ("model", C.c_float * 16),
                ("model_inv", C.c_float * 16)]


class RayCastUniform(C.Structure):
    """VkvRayCastUniform (src/volume_render_subpass.h:46-53)"""
    _fields_ = [("plane", C.c_float * 4), ("plane_tex", C.c_float * 4), ("camera_pos_tex", C.c_float * 4),
                ("block_size", C.c_float * 4), ("front_index", C.c_int32)]


class RayGen(C.Structure):
    """VkvRayGen"""
    _fields_ = [("dir00", C.c_float * 4), ("ddx", C.c_float * 4), ("ddy", C.c_float * 4)]


class RenderOptions(C.Structure):
    """VkvRenderOptions (VolumeRenderSubpass::Options, src/volume_render_subpass.h:74-81)"""
    _fields_ = [("skipping_type", C.c_int32), ("clip_distance", C.c_float), ("early_ray_termination", C.c_int32),
                ("depth_attachment", C.c_int32), ("test", C.c_int32)]

    def __init__(self, skipping_type=SKIP_DISTANCE, clip_distance=50.0, early_ray_termination=True,
                 depth_attachment=False, test=TEST_NONE):
        super().__init__(skipping_type, clip_distance, 1 if early_ray_termination else 0,
                         1 if depth_attachment else 0, test)


class TileRect(C.Structure):
    """VkvTileRect: x0, y0 = first tile column / row, w x h tiles (all zero: the whole image)"""
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32)]

    def as_tuple(self):
        return (self.x0, self.y0, self.w, self.h)

    @property
    def tiles(self):
        return self.w * self.h


class TileSchedule(C.Structure):
    """VkvTileSchedule"""
    _fields_ = [("tile_width", C.c_uint32), ("tile_height", C.c_uint32), ("tile_first", C.c_uint32),
                ("tile_stride", C.c_uint32), ("tile_count", C.c_uint32), ("compact", C.c_uint32), ("rect", TileRect),
                ("fill_outside", C.c_uint32)]


class RenderParams(C.Structure):
    """VkvRenderParams"""
    _fields_ = [("camera", CameraUniform), ("ray_cast", RayCastUniform),
                ("transfer_function", TransferFunctionUniform), ("ray_gen", RayGen), ("options", RenderOptions),
                ("use_precomputed_gradient", C.c_uint32), ("image_width", C.c_uint32), ("image_height", C.c_uint32),
                ("tiles", TileSchedule), ("volume_extent", Extent3D), ("map_extent", Extent3D),
                ("d_volume", C.c_void_p), ("d_gradient", C.c_void_p), ("d_transfer_function", C.c_void_p),
                ("d_distance_maps", C.c_void_p * 8), ("d_packed_volume", C.c_void_p),
                ("d_transfer_function_bits", C.c_void_p), ("d_out_color", C.c_void_p), ("d_out_rgba8", C.c_void_p),
                ("d_out_counts", C.c_void_p), ("d_out_depth", C.c_void_p), ("d_in_depth", C.c_void_p),
                ("blend_over_target", C.c_uint32)]


class Tuning(C.Structure):
    """VkvTuning"""
    _fields_ = [("struct_size", C.c_uint32), ("scheduler", C.c_int32), ("batch_mode", C.c_int32), ("batch_sequential", C.c_int32),
                ("tile_order_linear", C.c_int32), ("address_tables", C.c_int32), ("full_table_lds_limit", C.c_uint32),
                ("screen_cull", C.c_int32), ("feedback", C.c_int32), ("feedback_period", C.c_uint32), ("clamp_always", C.c_int32),
                ("tile_mix_heavy", C.c_float), ("tile_mix_spread", C.c_float), ("gradient_segment", C.c_uint32),
                ("pack_tile", C.c_int32), ("wave_shape", C.c_int32), ("occupancy_kernel", C.c_int32), ("arena_bytes", C.c_uint32)]


class VolumeHeader(C.Structure):
    """VkvVolumeHeader (LoadVolume::Header, src/load_volume.h:29-39)"""
    _fields_ = [("extent", Extent3D), ("voxel_size", C.c_float * 3), ("normalisation_range", C.c_float * 2),
                ("type", C.c_char * 16), ("endianness", C.c_char * 16), ("image_transform", C.c_float * 16)]


def full_frame_tiles(image_width, image_height, tile_width=16, tile_height=16, rank=0, world=1, compact=False, rect=None, fill_outside=False):
    """Tile schedule of one rank: every ``world``-th tile starting at ``rank`` (interleaved screen tiles) of the whole image or, with
    ``rect`` (a TileRect, e.g. from lib.screen_tile_rect), of that tile rectangle.  ``fill_outside`` (one rank, image-indexed outputs): the
    launch also writes the no-fragment result outside the rectangle - the whole frame without a workgroup per empty tile."""
    if rect is not None and rect.w and rect.h:
        total, r = rect.w * rect.h, TileRect(rect.x0, rect.y0, rect.w, rect.h)
    else:
        tiles_x = (image_width + tile_width - 1) // tile_width
        tiles_y = (image_height + tile_height - 1) // tile_height
        total, r = tiles_x * tiles_y, TileRect(0, 0, 0, 0)
    count = (total - rank + world - 1) // world if total > rank else 0
    return TileSchedule(tile_width, tile_height, rank, world, count, 1 if compact else 0, r, 1 if (fill_outside and r.w and not compact and world == 1) else 0)


def whole_image_rect(image_width, image_height, tile_width=16, tile_height=16):
    return TileRect(0, 0, (image_width + tile_width - 1) // tile_width, (image_height + tile_height - 1) // tile_height)
