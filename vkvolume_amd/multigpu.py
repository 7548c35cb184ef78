"""Screen-tile sharding of one frame over the ranks of a node.

Rays are independent, so the path shards by screen tiles with the volume replicated per GPU and exactly one exchange
step: every rank's compact tile buffer is gathered to the frame's owner (RCCL gather over xGMI = 7 concurrent point-to-point
transfers, one per link into the root), where ``vkv_scatter_tiles`` de-interleaves them into the frame.  The owner can rotate
over the ranks frame by frame (``any_root``), which spreads the inbound traffic and the de-interleave over all GPUs.

Tiles are dealt round-robin (tile t -> rank t mod world) because empty-space skipping makes per-pixel cost vary by
more than 10x; contiguous strips would not balance.
"""
import numpy as np

from . import abi


class TileGather:
    """Double-buffered gather of per-rank compact tile buffers to rank 0, overlapping the collective of frame k with the
    render of frame k+1.  Works with any torch.distributed backend (nccl == RCCL on ROCm; gloo in the CPU tests)."""

    def __init__(self, dist, rank, world, frame_size, tile=16, bytes_per_pixel=4, device="cuda", n_buffers=2, any_root=False):
        import torch
        self.dist, self.rank, self.world = dist, rank, world
        self.frame_size, self.tile, self.bpp = frame_size, tile, bytes_per_pixel
        fw, fh = frame_size
        self.tiles_x, self.tiles_y = (fw + tile - 1) // tile, (fh + tile - 1) // tile
        self.total_tiles = self.tiles_x * self.tiles_y
        self.tiles_per_rank = (self.total_tiles + world - 1) // world
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

    def my_ray_count(self):
        """in-image pixels of this rank's tiles (edge tiles are partial)"""
        fw, fh = self.frame_size
        n = 0
        for k in range(self.schedule.tile_count):
            t = self.schedule.tile_first + k * self.schedule.tile_stride
            x0, y0 = (t % self.tiles_x) * self.tile, (t // self.tiles_x) * self.tile
            n += min(self.tile, fw - x0) * min(self.tile, fh - y0)
        return n

    def start(self, b, root=0):
        """launch the gather of buffer b to `root` (asynchronous); every rank must pass the same root"""
        if root != 0 and not self.any_root:
            raise ValueError("TileGather was created for rank 0 as the only frame owner")
        gl = [self.flat[b][r] for r in range(self.world)] if self.rank == root else None
        self.roots[b] = root
        self.works[b] = self.dist.gather(self.buffers[b], gl, dst=root, async_op=True)

    def finish(self, b):
        """wait for buffer b's gather; returns the root's [world, n, bpp] tensor (None on the other ranks / if nothing pending)"""
        if self.works[b] is None:
            return None
        self.works[b].wait()
        self.works[b] = None
        return self.flat[b] if self.rank == self.roots[b] else None


class BatchTileGather(TileGather):
    """The exchange of a whole vkv_render_batch launch in ONE collective: every rank sends the compact tile buffers of the launch's
    frames as one block [frames][tiles_per_rank x tile pixels]; the owner of the launch receives [world][frames][...] and
    de-interleaves frame f with vkv_scatter_tiles on the block's f-th slice (rank stride = frames x tiles_per_rank tiles).  One
    gather and one host call per launch instead of one per frame."""

    def __init__(self, dist, rank, world, frame_size, tile=16, bytes_per_pixel=4, device="cuda", frames=8, n_sets=2, any_root=False, host_staging=False):
        import torch
        super().__init__(dist, rank, world, frame_size, tile, bytes_per_pixel, device, n_buffers=0, any_root=any_root)
        # host_staging: the backend cannot gather device tensors (gloo): the block goes through host memory - a functional path for tests
        # of the N > 1 orchestration (bench.py --backend gloo), not a data path anybody should measure
        self.host_staging, self._host = host_staging, [None] * n_sets
        n = self.tiles_per_rank * tile * tile
        self.frames = frames
        self.sets = [torch.zeros((frames, n, bytes_per_pixel), dtype=torch.uint8, device=device) for _ in range(n_sets)]
        self.buffers = [s[j] for s in self.sets for j in range(frames)]  # buffer of frame j of set b: buffers[b * frames + j]
        self.flat = None
        if rank == 0 or any_root:
            self.flat = [torch.zeros((world, frames, n, bytes_per_pixel), dtype=torch.uint8, device=device) for _ in range(n_sets)]
        self.works, self.roots, self.counts = [None] * n_sets, [0] * n_sets, [0] * n_sets

    def start(self, b, root=0, n_frames=None):
        """gather the first n_frames frames of set b to `root` (asynchronous); every rank passes the same root and count"""
        if root != 0 and not self.any_root:
            raise ValueError("BatchTileGather was created for rank 0 as the only frame owner")
        n = self.frames if n_frames is None else n_frames
        self.roots[b], self.counts[b] = root, n
        if self.host_staging:
            import torch
            mine = self.sets[b][:n].cpu()  # (waits for the render on the current stream)
            gl = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == root else None
            self._host[b] = gl
            self.works[b] = self.dist.gather(mine, gl, dst=root, async_op=True)
            return
        gl = [self.flat[b][r, :n] for r in range(self.world)] if self.rank == root else None
        self.works[b] = self.dist.gather(self.sets[b][:n], gl, dst=root, async_op=True)

    def finish(self, b):
        """wait for set b's gather; on its owner returns (flat [world, frames, n, bpp], number of frames), elsewhere None"""
        if self.works[b] is None:
            return None
        self.works[b].wait()
        self.works[b] = None
        if self.host_staging and self.rank == self.roots[b]:
            for r in range(self.world):
                self.flat[b][r, :self.counts[b]].copy_(self._host[b][r])
            self._host[b] = None
        return (self.flat[b], self.counts[b]) if self.rank == self.roots[b] else None

    def frame_source(self, flat, f):
        """(device pointer, rank stride in tiles) that make vkv_scatter_tiles read frame f of a gathered launch"""
        per_rank_bytes = self.tiles_per_rank * self.tile * self.tile * self.bpp
        return flat.data_ptr() + f * per_rank_bytes, self.frames * self.tiles_per_rank


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
        self.tiles_x, self.tiles_y = (fw + tile - 1) // tile, (fh + tile - 1) // tile
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

    def assemble(self, b, root, stream, scatter_stream=None):
        """enqueue gather + de-interleave of buffer b on `stream` (a torch stream); every rank must pass the same root.
        With ``scatter_stream`` the de-interleave runs there, behind an event recorded after the gather (the gather of the next frame
        then does not queue behind this frame's de-interleave).  Returns the stream whose completion frees buffer b."""
        if root != 0 and not self.any_root:
            raise ValueError("NativeExchange was created for rank 0 as the only frame owner")
        is_root = self.rank == root
        if scatter_stream is None:
            self.ctx.assemble_frame(self.buffers[b].data_ptr(), self.flat[b].data_ptr() if is_root else None,
                                    self.images[b].data_ptr() if is_root else None, self.frame_size, (self.tile, self.tile), self.world, self.rank,
                                    self.tiles_per_rank, self.bpp, root, self._comm.value, stream.cuda_stream)
            return stream
        import torch
        self.ctx.gather_tiles(self.buffers[b].data_ptr(), self.flat[b].data_ptr() if is_root else None, self.buffers[b].numel(), root,
                              self._comm.value, stream.cuda_stream)
        if not is_root:
            return stream
        gathered = torch.cuda.Event()
        gathered.record(stream)
        scatter_stream.wait_event(gathered)
        self.ctx.scatter_tiles(self.flat[b].data_ptr(), self.images[b].data_ptr(), self.frame_size, (self.tile, self.tile), self.world,
                               self.tiles_per_rank, self.bpp, scatter_stream.cuda_stream)
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
    [frame][tiles] block to the launch's owner + ONE de-interleave kernel there, enqueued on a HIP stream; the same launches and the
    same buffer sets as BatchTileGather, without torch.distributed on the data path."""

    def __init__(self, ctx, dist, rank, world, frame_size, tile=16, bytes_per_pixel=4, frames=8, n_sets=2, any_root=False, rccl_path=None):
        import torch
        super().__init__(ctx, dist, rank, world, frame_size, tile, bytes_per_pixel, n_buffers=0, any_root=any_root, rccl_path=rccl_path)
        fw, fh = frame_size
        n = self.tiles_per_rank * tile * tile
        self.frames = frames
        self.sets = [torch.zeros((frames, n, bytes_per_pixel), dtype=torch.uint8, device="cuda") for _ in range(n_sets)]
        self.buffers = [s[j] for s in self.sets for j in range(frames)]  # buffer of frame j of set b: buffers[b * frames + j]
        own = rank == 0 or any_root
        self.flat = [torch.zeros((world, frames, n, bytes_per_pixel), dtype=torch.uint8, device="cuda") for _ in range(n_sets)] if own else None
        self.images = [torch.zeros((fh, fw, bytes_per_pixel), dtype=torch.uint8, device="cuda") for _ in range(n_sets * frames)] if own else None

    def assemble(self, b, root, n_frames, stream):
        """enqueue the exchange of the first n_frames frames of buffer set b on `stream` (a torch stream): the images of the launch land
        in images[b * frames + j] on `root`.  Every rank passes the same root and count."""
        if root != 0 and not self.any_root:
            raise ValueError("NativeBatchExchange was created for rank 0 as the only frame owner")
        is_root = self.rank == root
        imgs = [self.images[b * self.frames + j].data_ptr() for j in range(n_frames)] if is_root else None
        self.ctx.assemble_frames(self.sets[b].data_ptr(), self.flat[b].data_ptr() if is_root else None, imgs, n_frames, self.frame_size,
                                 (self.tile, self.tile), self.world, self.rank, self.tiles_per_rank, self.bpp, root, self._comm.value, stream.cuda_stream)
        return stream


def deinterleave_reference(flat, frame_size, tile, world):
    """numpy statement of vkv_scatter_tiles (tests only): flat[rank, k*tile*tile + ly*tile + lx, c] -> image[y, x, c]."""
    fw, fh = frame_size
    tiles_x = (fw + tile - 1) // tile
    flat = np.asarray(flat)
    img = np.zeros((fh, fw, flat.shape[-1]), flat.dtype)
    y, x = np.mgrid[0:fh, 0:fw]
    t = (y // tile) * tiles_x + (x // tile)
    img[y, x] = flat[t % world, ((t // world) * tile + (y % tile)) * tile + (x % tile)]
    return img
