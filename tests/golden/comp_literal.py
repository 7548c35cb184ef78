"""A second, independent statement of the reference's distance-transform compute shaders, written from the GLSL text and the host
schedule, statement by statement, in plain Python loops (small maps only) - no code shared with oracle/vkv_oracle.c:

  shaders/distance_map.comp:44-109               stages 0 / 1 / 2 of the isotropic transform (stage 0 in place, the zig-zag search with
                                                 its early exit, uint arithmetic, r8ui stores)
  src/compute_distance_map.cpp:142-175           which image is `dist` and which `dist_swap` in each dispatch
  shaders/distance_map_anisotropic.comp:31-92    the one-sided stages (dir = +1 / -1)
  src/compute_distance_map.cpp:201-252           the 14-dispatch schedule over maps 0..7 and the swap image

Arrays are uint8 [D][H][W] (x fastest), like the maps everywhere else in this repository.  Test infrastructure: only tests/ use it."""
import numpy as np


def _stage0_iso(dist, dist_swap):
    """distance_map.comp:56-71; one invocation per (y, z) row.  dist and dist_swap may be the same array (compute_distance_map.cpp:156-157)."""
    d, h, w = dist.shape
    for z in range(d):
        for y in range(h):
            gi1jk = int(dist_swap[z, y, 0])
            for x in range(1, w):        # forward
                gijk = min(gi1jk + 1, int(dist_swap[z, y, x]))
                dist[z, y, x] = gijk
                gi1jk = gijk
            for x in range(w - 2, -1, -1):        # backward
                gijk = min(gi1jk + 1, int(dist[z, y, x]))
                dist[z, y, x] = gijk
                gi1jk = gijk


def _stage12_iso(src, dst, axis):
    """distance_map.comp:72-107: `axis` 1 = stage 1 (along y, dist -> dist_swap), 2 = stage 2 (along z, dist_swap -> dist)"""
    d, h, w = src.shape
    n_axis = h if axis == 1 else d
    for z in range(d):
        for y in range(h):
            for x in range(w):
                p = y if axis == 1 else z
                D = int(src[z, y, x])
                n = 1
                while n < D:
                    if p >= n:
                        D_n = int(src[z, y - n, x]) if axis == 1 else int(src[z - n, y, x])
                        D = min(D, max(n, D_n))
                    if (p + n) < n_axis and n < D:        # "note early exit"
                        D_n = int(src[z, y + n, x]) if axis == 1 else int(src[z + n, y, x])
                        D = min(D, max(n, D_n))
                    n += 1
                dst[z, y, x] = D


def distance_map(occupancy):
    """ComputeDistanceMap::computeDistance (src/compute_distance_map.cpp:142-175): stage 0 with dist == dist_swap == the map, stage 1
    map -> swap, stage 2 swap -> map"""
    dist = np.array(occupancy, dtype=np.uint8, copy=True)
    swap = np.zeros_like(dist)
    _stage0_iso(dist, dist)
    _stage12_iso(dist, swap, 1)
    _stage12_iso(swap, dist, 2)
    return dist


def _stage0_aniso(dist, dist_swap, direction):
    """distance_map_anisotropic.comp:43-53"""
    d, h, w = dist.shape
    start = w - 1 if direction > 0 else 0
    end = -1 if direction > 0 else w
    for z in range(d):
        for y in range(h):
            gi1jk = int(dist_swap[z, y, start])
            x = start
            while x != end:
                gijk = min(gi1jk + 1, int(dist_swap[z, y, x]))
                dist[z, y, x] = gijk
                gi1jk = gijk
                x -= direction


def _stage12_aniso(src, dst, axis, direction):
    """distance_map_anisotropic.comp:55-91: one-sided search along y (stage 1, dist -> dist_swap) or z (stage 2, dist_swap -> dist)"""
    d, h, w = src.shape
    n_axis = h if axis == 1 else d
    for z in range(d):
        for y in range(h):
            for x in range(w):
                p = y if axis == 1 else z
                m_min = int(src[z, y, x])
                n = 1
                while n < m_min and n < 255:
                    t = p + direction * n
                    if t < 0 or t >= n_axis:
                        break
                    g = int(src[z, t, x]) if axis == 1 else int(src[t, y, x])
                    m = max(n, g)
                    if m < m_min:
                        m_min = m
                    n += 1
                dst[z, y, x] = m_min


def distance_map_anisotropic(occupancy):
    """ComputeDistanceMap::computeDistanceAnisotropic (src/compute_distance_map.cpp:177-252): the occupancy map lives in map 7; returns the
    eight maps, index = (z < 0) + 2 (y < 0) + 4 (x < 0) of the ray direction."""
    m = [np.zeros_like(occupancy, dtype=np.uint8) for _ in range(8)]
    m[7] = np.array(occupancy, dtype=np.uint8, copy=True)
    swap = np.zeros_like(m[7])
    stage1 = lambda idx, direction: _stage0_aniso(m[idx], m[7], direction)              # binding 0 = map idx, binding 1 = occupancy map
    stage2 = lambda idx, direction: _stage12_aniso(m[idx], swap, 1, direction)           # reads `dist` = map idx, writes `dist_swap` = swap
    stage3 = lambda idx, direction: _stage12_aniso(swap, m[idx], 2, direction)           # reads swap, writes map idx
    stage1(3, 1)
    stage2(3, 1)
    stage3(0, 1)
    stage3(1, -1)
    stage2(3, -1)
    stage3(2, 1)
    stage3(3, -1)
    stage1(7, -1)
    stage2(7, 1)
    stage3(4, 1)
    stage3(5, -1)
    stage2(7, -1)
    stage3(6, 1)
    stage3(7, -1)
    return m
