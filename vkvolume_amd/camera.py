"""Camera matrices for the offscreen driver (stand-in for the vkb free camera the reference gets from
Vulkan-Samples, src/volume_render.cpp:169; pose and fov are this build's own definition).

All matrices are column-major 4x4 stored as numpy float32 arrays of shape (4, 4) in *memory order*
``m[col][row]`` — i.e. ``m.reshape(16)`` is the glm::mat4 byte layout the C ABI expects.
"""
import math

import numpy as np


def _colmajor(rows):
    """rows: 4x4 nested list in mathematical (row, col) order → column-major float32 [col][row]."""
    return np.ascontiguousarray(np.array(rows, np.float64).T, np.float32)


def look_at(eye, centre, up=(0.0, 1.0, 0.0)):
    """glm::lookAtRH"""
    eye, centre, up = (np.asarray(v, np.float64) for v in (eye, centre, up))
    f = centre - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    return _colmajor([[s[0], s[1], s[2], -s.dot(eye)],
                      [u[0], u[1], u[2], -u.dot(eye)],
                      [-f[0], -f[1], -f[2], f.dot(eye)],
                      [0, 0, 0, 1]])


def perspective_vulkan(fov_y_deg, aspect, near=0.1, far=1000.0):
    """vkb::PerspectiveCamera::get_projection (reverse-Z: glm::perspectiveRH_ZO(fov, aspect, far, near)) followed by
    vkb::vulkan_style_projection (y flipped), as used at src/volume_render_subpass.cpp:224."""
    t = math.tan(math.radians(fov_y_deg) / 2.0)
    zn, zf = far, near  # swapped on purpose
    return _colmajor([[1.0 / (aspect * t), 0, 0, 0],
                      [0, -1.0 / t, 0, 0],
                      [0, 0, zf / (zn - zf), -(zf * zn) / (zf - zn)],
                      [0, 0, -1, 0]])


def scale(s):
    s = np.broadcast_to(np.asarray(s, np.float64), (3,))
    return _colmajor([[s[0], 0, 0, 0], [0, s[1], 0, 0], [0, 0, s[2], 0], [0, 0, 0, 1]])


def rotate(angle_deg, axis):
    """glm::rotate(angle, axis)"""
    a = math.radians(angle_deg)
    c, s = math.cos(a), math.sin(a)
    n = np.asarray(axis, np.float64)
    n = n / np.linalg.norm(n)
    t = (1 - c) * n
    r = np.array([[c + t[0] * n[0], t[1] * n[0] - s * n[2], t[2] * n[0] + s * n[1], 0],
                  [t[0] * n[1] + s * n[2], c + t[1] * n[1], t[2] * n[1] - s * n[0], 0],
                  [t[0] * n[2] - s * n[1], t[1] * n[2] + s * n[0], c + t[2] * n[2], 0],
                  [0, 0, 0, 1]])
    return _colmajor(r)


def matmul(a, b):
    """column-major product a*b"""
    return np.ascontiguousarray((a.astype(np.float64).T @ b.astype(np.float64).T).T, np.float32)


def image_transform(voxel_size, extent_whd, axis_angle=(1.0, 0.0, 0.0, 0.0)):
    """LoadVolume::load_header's image_transform = rotate(angle, axis) * scale(voxel_size * extent)
    (src/load_volume.cpp:82-83)."""
    size = np.asarray(voxel_size, np.float64) * np.asarray(extent_whd, np.float64)
    return matmul(rotate(axis_angle[3], axis_angle[:3]), scale(size))


def benchmark_node_transform(image_xf, scale_factor=100.0):
    """Benchmark-mode node scale: the longest physical edge becomes ``scale_factor`` units
    (src/volume_render.cpp:224-238 rescales by the rotated physical size; SURVEY.md §8d keeps aspect)."""
    m = image_xf.astype(np.float64).T[:3, :3]
    longest = max(np.linalg.norm(m[:, i]) for i in range(3))
    return scale(scale_factor / longest)


def orbit_camera(azimuth_deg, elevation_deg, radius, centre=(0.0, 0.0, 0.0)):
    """Camera on a sphere around ``centre`` looking at it (8 azimuths, elevation 20 deg)."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    eye = np.asarray(centre, np.float64) + radius * np.array([math.cos(el) * math.sin(az), math.sin(el),
                                                              math.cos(el) * math.cos(az)])
    return look_at(eye, centre)
