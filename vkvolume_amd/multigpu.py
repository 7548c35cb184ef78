"""Screen-tile sharding of one frame over the ranks of a node.

Rays are independent, so the path shards by screen tiles with the volume replicated per GPU and exactly one exchange
step: every rank's compact tile buffer is gathered to the frame's owner (RCCL gather over xGMI = 7 concurrent point-to-point
transfers, one per link into the root), where ``vkv_scatter_tiles`` de-interleaves them into the frame.  The owner can rotate
over the ranks launch by launch (``any_root``) or - round 6 - the frames of ONE launch can have different owners (``roots``): every
GPU then receives at the same time over its own inbound links.

Tiles are dealt round-robin (tile t -> rank t mod world) because empty-space skipping makes per-pixel cost vary by
more than 10x; contiguous strips would not balance.

Round 6: only the tiles a frame HAS are scheduled and exchanged.  ``lib.screen_tile_rect`` (vkv_screen_tile_rect) derives from the uniforms
alone the tile rectangle the clipped box projects into - the reference's rasteriser only shades fragments of the box's faces
(src/volume_render_subpass.cpp:262-293) - every rank derives the same rectangle, tiles are numbered inside it, and the de-interleave clears the
rest of the image.  A frame with rectangle R moves ceil(|R| / world) tiles per rank instead of ceil(all tiles / world).
"""
import numpy as np

from . import abi


def tiles_per_rank(rect, world):
    """tiles a rank holds of a frame whose scheduled tiles are those of `rect` (ragged deals round up: the fixed size of the collective)"""
    return (rect.w * rect.h + world - 1) // world


def launch_layout(rects, world):
    """[frame][tiles] block of one launch: (tiles per rank of every frame, tile offset of every frame inside a rank's block, the block's tiles)"""
    tpr = [tiles_per_rank(r, world) for r in rects]
    off = [0]
    for n in tpr:
        off.append(off[-1] + n)
    return tpr, off[:-1], off[-1]


def rect_ray_count(rect, rank, world, frame_size, tile):
    """in-image pixels of rank's tiles of `rect` (edge tiles of the image are partial)"""
    fw, fh = frame_size
    t = np.arange(rank, rect.w * rect.h, world, dtype=np.int64)
    x0, y0 = (rect.x0 + t % rect.w) * tile, (rect.y0 + t // rect.w) * tile
    return int((np.minimum(tile, fw - x0) * np.minimum(tile, fh - y0)).sum())


class TileGather:
    """Double-buffered gather of per-rank compact tile buffers to the frame's owner, overlapping the collective of frame k with the
    render of frame k+1.  Works with any torch.distributed backend (nccl == RCCL on ROCm; gloo in the CPU tests).  ``rect`` (a TileRect) on
    start(): only the rectangle's tiles travel; the buffers are sized for the whole image."""

    def __init__(self, dist, rank, world, frame_size, tile=16, bytes_per_pixel=4, device="cuda", n_buffers=2, any_root=False):
        import torch
        self.dist, self.rank, self.world = dist, rank, world
        self.frame_size, self.tile, self.bpp = frame_size, tile, bytes_per_pixel
        fw, fh = frame_size
        self.whole = abi.whole_image_rect(fw, fh, tile, tile)
        self.tiles_x, self.tiles_y = self.whole.w, self.whole.h
        self.total_tiles = self.tiles_x * self.tiles_y
        self.tiles_per_rank = (self.total_tiles + world - 1) // world  # capacity: a whole image's share
        self.schedule = abi.full_frame_tiles(fw, fh, tile, tile, rank, world, compact=True)
        n = self.tiles_per_rank * tile * tile
        self.buffers = [torch.zeros((n, bytes_per_pixel), dtype=torch.uint8, device=device) for _ in range(n_buffers)]
        # receive buffers live on every rank that can own a frame: rank 0 only, or all ranks (any_root) when the owner rotates -
        # xGMI is point-to-point, so frames gathered to different roots travel over disjoint links and pipeline
        self.flat = None
        if rank == 0 or any_root:
            self.flat = [torch.zeros((world, n, bytes_per_pixel), dtype=torch.uint8, device=device) for _ in range(n_buffers)]
        self.any_root = any_root
        self.works = [None] * n_buffers
        self.roots = [0] * n_buffers

    def rect_schedule(self, rect=None):
        """this rank's VkvTileSchedule of a frame whose scheduled tiles are those of `rect` (None: the whole image)"""
        fw, fh = self.frame_size
        return abi.full_frame_tiles(fw, fh, self.tile, self.tile, self.rank, self.world, compact=True, rect=rect)

    def my_ray_count(self, rect=None):
        """in-image pixels of this rank's tiles (edge tiles are partial)"""
        return rect_ray_count(rect if rect is not None and rect.w and rect.h else self.whole, self.rank, self.world, self.frame_size, self.tile)

    def start(self, b, root=0, rect=None):
        """launch the gather of buffer b to `root` (asynchronous); every rank must pass the same root and rectangle"""
        if root != 0 and not self.any_root:
            raise ValueError("TileGather was created for rank 0 as the only frame owner")
        n = tiles_per_rank(rect if rect is not None and rect.w and rect.h else self.whole, self.world) * self.tile * self.tile
        gl = [self.flat[b][r, :n] for r in range(self.world)] if self.rank == root else None
        self.roots[b] = root
        self.works[b] = self.dist.gather(self.buffers[b][:n], gl, dst=root, async_op=True)

    def finish(self, b):
        """wait for buffer b's gather; returns the root's [world, n, bpp] tensor (None on the other ranks / if nothing pending): rank r's
        tiles start at r * tiles_per_rank tiles (the capacity), whatever the rectangle"""
        if self.works[b] is None:
            return None
        self.works[b].wait()
        self.works[b] = None
        return self.flat[b] if self.rank == self.roots[b] else None


class BatchTileGather(TileGather):
    """The exchange of a whole vkv_render_batch launch through torch.distributed.  A launch's frames have their own tile rectangles
    (``rects``): a rank's block is [frame f][tiles_per_rank(rects[f]) tiles] back to back (``launch_layout``), rendered straight into
    set b at those offsets (``frame_pointer``).  One owner for the launch (``roots`` None): ONE gather of the block, the owner holds
    [rank][block] and de-interleaves frame f with vkv_scatter_tiles on the block's f-th slice (rank stride = the block's tiles).  Owners
    spread over the launch's frames (``roots``): one gather per frame to its owner, which holds [rank][tiles of the frame]."""

    def __init__(self, dist, rank, world, frame_size, tile=16, bytes_per_pixel=4, device="cuda", frames=8, n_sets=2, any_root=False, host_staging=False):
        import torch
        super().__init__(dist, rank, world, frame_size, tile, bytes_per_pixel, device, n_buffers=0, any_root=any_root)
        # host_staging: the backend cannot gather device tensors (gloo): the block goes through host memory - a functional path for tests
        # of the N > 1 orchestration (bench.py --backend gloo), not a data path anybody should measure
        self.host_staging, self._host = host_staging, [None] * n_sets
        self.frames = frames
        self.tile_bytes = tile * tile * bytes_per_pixel
        cap = frames * self.tiles_per_rank * self.tile_bytes  # bytes of a rank's block when every frame is a whole image
        self.sets = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(n_sets)]
        self.flat = None
        if rank == 0 or any_root:
            self.flat = [torch.zeros(world * cap, dtype=torch.uint8, device=device) for _ in range(n_sets)]
        self.works, self.pending = [None] * n_sets, [None] * n_sets

    def frame_pointer(self, b, tile_offset):
        """device pointer of the compact buffer of the frame that starts `tile_offset` tiles into this rank's block of set b"""
        return self.sets[b].data_ptr() + tile_offset * self.tile_bytes

    def start(self, b, root=0, rects=None, roots=None, n_frames=None):
        """gather the launch rendered into set b (asynchronous): frames with tile rectangles `rects` (None: n_frames whole images) to `root`, or
        frame f to roots[f]; every rank passes the same arguments"""
        import torch
        if rects is None:
            rects = [self.whole] * (self.frames if n_frames is None else n_frames)
        owners = [root] * len(rects) if roots is None else list(roots)
        if any(o != 0 for o in owners) and not self.any_root:
            raise ValueError("BatchTileGather was created for rank 0 as the only frame owner")
        tpr, off, total = launch_layout(rects, self.world)
        tb, W = self.tile_bytes, self.world
        one = roots is None  # one gather of the whole block, or one gather per frame (the same rule as vkv_assemble_frames)
        self.pending[b] = (rects, owners, tpr, off, total, one)
        works = []
        if one:
            spans = [(owners[0], 0, total)]  # (owner, first tile of the span in a rank's block, tiles)
        else:
            spans = [(owners[f], off[f], tpr[f]) for f in range(len(rects))]
        if self.host_staging:
            self._host[b] = []
        for owner, o, n in spans:
            mine = self.sets[b][o * tb:(o + n) * tb]
            # the owner's receive area: [rank][span] - at tile W * o of flat[b] (one owner: o = 0, the whole block per rank)
            dst = self.flat[b][W * o * tb:W * (o + n) * tb].view(W, n * tb) if self.rank == owner else None
            if self.host_staging:
                mine_h = mine.cpu()  # (waits for the render on the current stream)
                gl = [torch.empty_like(mine_h) for _ in range(W)] if self.rank == owner else None
                self._host[b].append((dst, gl))
                works.append(self.dist.gather(mine_h, gl, dst=owner, async_op=True))
            else:
                gl = [dst[r] for r in range(W)] if self.rank == owner else None
                works.append(self.dist.gather(mine, gl, dst=owner, async_op=True))
        self.works[b] = works

    def finish(self, b):
        """wait for set b's exchange; returns the frames this rank owns as a list of (frame index, source pointer, rank stride in tiles, rect)
        for vkv_scatter_tiles (empty on ranks that own nothing), or None if nothing was pending"""
        if self.works[b] is None:
            return None
        for w in self.works[b]:
            w.wait()
        self.works[b] = None
        rects, owners, tpr, off, total, one = self.pending[b]
        if self.host_staging:
            for dst, gl in self._host[b]:
                if dst is not None:
                    for r in range(self.world):
                        dst[r].copy_(gl[r])
            self._host[b] = None
        tb, W, base = self.tile_bytes, self.world, self.flat[b].data_ptr() if self.flat is not None else 0
        out = []
        for f, r in enumerate(rects):
            if owners[f] != self.rank:
                continue
            if one:
                out.append((f, base + off[f] * tb, total, r))  # [rank][block]: frame f starts off[f] tiles into every rank's block
            else:
                out.append((f, base + W * off[f] * tb, tpr[f], r))  # [rank][tiles of frame f] at tile W * off[f]
        return out


class NativeExchange:
    """The exchange step through the product's C ABI: ``vkv_assemble_frame`` = ``ncclGather`` (RCCL over xGMI) of the compact tile
    buffers to the frame's owner + de-interleave there, enqueued on a HIP stream — no torch.distributed on the data path.
    torch.distributed is only the bootstrap that carries the ncclUniqueId from rank 0 to the others."""

    def __init__(self, ctx, dist, rank, world, frame_size, tile=16, bytes_per_pixel=4, n_buffers=2, any_root=False, rccl_path=None):
        import ctypes as C
        import os
        import torch
        self.ctx, self.rank, self.world = ctx, rank, world
        self.frame_size, self.tile, self.bpp = frame_size, tile, bytes_per_pixel
        fw, fh = frame_size
        self.whole = abi.whole_image_rect(fw, fh, tile, tile)
        self.tiles_x, self.tiles_y = self.whole.w, self.whole.h
        self.total_tiles = self.tiles_x * self.tiles_y
        self.tiles_per_rank = (self.total_tiles + world - 1) // world
        self.schedule = abi.full_frame_tiles(fw, fh, tile, tile, rank, world, compact=True)
        n = self.tiles_per_rank * tile * tile
        self.buffers = [torch.zeros((n, bytes_per_pixel), dtype=torch.uint8, device="cuda") for _ in range(n_buffers)]
        own = rank == 0 or any_root
        self.flat = [torch.zeros((world, n, bytes_per_pixel), dtype=torch.uint8, device="cuda") for _ in range(n_buffers)] if own else None
        self.images = [torch.zeros((fh, fw, bytes_per_pixel), dtype=torch.uint8, device="cuda") for _ in range(n_buffers)] if own else None
        self.any_root = any_root
        # communicator of our own, created with the RCCL copy the process has loaded (the one vkv_gather_tiles resolves)
        path = rccl_path or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._rccl = C.CDLL(path)

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        uid = UniqueId()
        if rank == 0:
            rc = self._rccl.ncclGetUniqueId(C.byref(uid))
            if rc != 0:
                raise RuntimeError("ncclGetUniqueId failed: %d" % rc)
        blob = [C.string_at(C.byref(uid), 128) if rank == 0 else None]  # all 128 bytes (a c_char array reads as a NUL-terminated string)
        if world > 1:
            dist.broadcast_object_list(blob, src=0)
        C.memmove(C.byref(uid), blob[0], 128)
        self._comm = C.c_void_p()
        self._rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        rc = self._rccl.ncclCommInitRank(C.byref(self._comm), world, uid, rank)
        if rc != 0:
            raise RuntimeError("ncclCommInitRank failed: %d" % rc)

    my_ray_count = TileGather.my_ray_count
    rect_schedule = TileGather.rect_schedule

    def assemble(self, b, root, stream, scatter_stream=None, rect=None):
        """enqueue gather + de-interleave of buffer b on `stream` (a torch stream); every rank must pass the same root and rectangle.
        With ``scatter_stream`` the de-interleave runs there, behind an event recorded after the gather (the gather of the next frame
        then does not queue behind this frame's de-interleave).  Returns the stream whose completion frees buffer b."""
        if root != 0 and not self.any_root:
            raise ValueError("NativeExchange was created for rank 0 as the only frame owner")
        is_root = self.rank == root
        if scatter_stream is None:
            self.ctx.assemble_frame(self.buffers[b].data_ptr(), self.flat[b].data_ptr() if is_root else None,
                                    self.images[b].data_ptr() if is_root else None, self.frame_size, (self.tile, self.tile), self.world, self.rank,
                                    self.bpp, root, self._comm.value, stream.cuda_stream, rect=rect)
            return stream
        import torch
        tpr = tiles_per_rank(rect if rect is not None and rect.w and rect.h else self.whole, self.world)
        self.ctx.gather_tiles(self.buffers[b].data_ptr(), self.flat[b].data_ptr() if is_root else None, tpr * self.tile * self.tile * self.bpp, root,
                              self._comm.value, stream.cuda_stream)
        if not is_root:
            return stream
        gathered = torch.cuda.Event()
        gathered.record(stream)
        scatter_stream.wait_event(gathered)
        self.ctx.scatter_tiles(self.flat[b].data_ptr(), self.images[b].data_ptr(), self.frame_size, (self.tile, self.tile), self.world,
                               tpr, self.bpp, scatter_stream.cuda_stream, rect=rect)
        return scatter_stream

    def comm_count(self):
        """ranks of the exchange's communicator as RCCL itself reports them (ncclCommCount)"""
        import ctypes as C
        n = C.c_int(0)
        self._rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        rc = self._rccl.ncclCommCount(self._comm, C.byref(n))
        if rc != 0:
            raise RuntimeError("ncclCommCount failed: %d" % rc)
        return int(n.value)

    def close(self):
        if self._comm:
            self._rccl.ncclCommDestroy.argtypes = [__import__("ctypes").c_void_p]
            self._rccl.ncclCommDestroy(self._comm)
            self._comm = None


class NativeBatchExchange(NativeExchange):
    """The exchange of a whole vkv_render_batch launch through the C ABI: ``vkv_assemble_frames`` = ONE ``ncclGather`` of the launch's
    [frame][tiles] block to the launch's owner (or one group of gathers, frame f to roots[f]) + ONE de-interleave kernel on every owner,
    enqueued on a HIP stream; the same launches, rectangles and buffer sets as BatchTileGather, without torch.distributed on the data path."""

    def __init__(self, ctx, dist, rank, world, frame_size, tile=16, bytes_per_pixel=4, frames=8, n_sets=2, any_root=False, rccl_path=None):
        import torch
        super().__init__(ctx, dist, rank, world, frame_size, tile, bytes_per_pixel, n_buffers=0, any_root=any_root, rccl_path=rccl_path)
        fw, fh = frame_size
        self.frames = frames
        self.tile_bytes = tile * tile * bytes_per_pixel
        cap = frames * self.tiles_per_rank * self.tile_bytes
        self.sets = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(n_sets)]
        own = rank == 0 or any_root
        self.flat = [torch.zeros(world * cap, dtype=torch.uint8, device="cuda") for _ in range(n_sets)] if own else None
        self.images = [torch.zeros((fh, fw, bytes_per_pixel), dtype=torch.uint8, device="cuda") for _ in range(n_sets * frames)] if own else None

    frame_pointer = BatchTileGather.frame_pointer

    def assemble(self, b, root, n_frames, stream, rects=None, roots=None):
        """enqueue the exchange of the first n_frames frames of buffer set b on `stream` (a torch stream): the image of frame j lands in
        images[b * frames + j] on its owner (`root`, or roots[j]).  Every rank passes the same arguments."""
        owners = [root] * n_frames if roots is None else list(roots)
        if any(o != 0 for o in owners) and not self.any_root:
            raise ValueError("NativeBatchExchange was created for rank 0 as the only frame owner")
        mine = any(o == self.rank for o in owners)
        imgs = [self.images[b * self.frames + j].data_ptr() if owners[j] == self.rank else None for j in range(n_frames)] if mine else None
        self.ctx.assemble_frames(self.sets[b].data_ptr(), self.flat[b].data_ptr() if mine else None, imgs, n_frames, self.frame_size,
                                 (self.tile, self.tile), self.world, self.rank, self.bpp, root, self._comm.value, stream.cuda_stream, rects=rects, roots=roots)
        return stream


def deinterleave_reference(flat, frame_size, tile, world, rect=None):
    """numpy statement of vkv_scatter_tiles (tests only): flat[rank, k*tile*tile + ly*tile + lx, c] -> image[y, x, c]; tiles numbered row-major
    inside `rect` (None: the whole image), pixels outside it zero."""
    fw, fh = frame_size
    if rect is None or not (rect.w and rect.h):
        rect = abi.whole_image_rect(fw, fh, tile, tile)
    flat = np.asarray(flat)
    img = np.zeros((fh, fw, flat.shape[-1]), flat.dtype)
    y, x = np.mgrid[0:fh, 0:fw]
    tx, ty = x // tile - rect.x0, y // tile - rect.y0
    inside = (tx >= 0) & (tx < rect.w) & (ty >= 0) & (ty < rect.h)
    t = np.where(inside, ty * rect.w + tx, 0)
    val = flat[t % world, ((t // world) * tile + (y % tile)) * tile + (x % tile)]
    img[inside] = val[inside]
    return img
