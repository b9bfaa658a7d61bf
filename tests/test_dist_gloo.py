"""world_size-2 gloo test of the multi-GPU path's host logic (cyclic row-tile partition,
padded gather, assembly on rank 0).  No GPU here, so each rank's part renderer is the CPU
oracle restricted to that rank's rows -- the product's ShardedRenderer is exercised as is."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OraclePartRenderer:
    def __init__(self, scene, h, w):
        self.sc, self.h, self.w = O.OracleScene(scene), h, w

    def __call__(self, part, nparts, out):
        from raytracers_amd.dist import tile_rows
        rows = tile_rows(self.h, part, nparts)
        full_rows = {}
        k = 0
        # render tile by tile (contiguous row bands) into the packed layout
        for t0 in range(0, len(rows), 8):
            band = rows[t0:t0 + 8]
            px, _ = self.sc.render(self.h, self.w, rows=(int(band[0]), int(band[-1]) + 1), threads=1)
            out[k:k + len(band)] = torch.from_numpy(px)
            k += len(band)
        return out


def _worker(rank, world, port, scene, h, w, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracers_amd.dist import ShardedRenderer
        sr = ShardedRenderer(OraclePartRenderer(scene, h, w), h, w, device="cpu")
        img = sr.render()
        img2 = sr.render()   # buffers are reused across frames
        if rank == 0:
            assert (img == img2).all()
            q.put(img.numpy().copy())
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scene,h,w", [("irreg", 52, 40), ("rgbbox", 17, 24)])
def test_two_rank_gather_equals_single_render(scene, h, w):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scene, h, w, q)) for r in range(2)]
    for p in procs:
        p.start()
    img = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full, _ = O.OracleScene(scene).render(h, w)
    assert (img == full).all()


def _step_worker(rank, world, port, frames, q, nbatch=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracers_amd.dist import ShardedStep
        st = ShardedStep([(OraclePartRenderer(s, h, w), h, w) for s, h, w in frames], device="cpu", nbatch=nbatch)
        imgs = st.render()
        imgs = st.render()   # buffers are reused across steps
        if rank == 0:
            q.put([i.numpy().copy() for i in imgs])
        else:
            assert imgs is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_one_gather_per_step_of_several_frames(world):
    """ShardedStep: the rows of two frames of different sizes travel in ONE gather."""
    frames = [("rgbbox", 27, 24), ("irreg", 41, 32)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_step_worker, args=(r, world, port, frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    imgs = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for (scene, h, w), img in zip(frames, imgs):
        full, _ = O.OracleScene(scene).render(h, w)
        assert (img == full).all()


def test_batched_step_three_frames_of_each_scene_in_one_gather():
    """ShardedStep(nbatch=3): the gather carries three frames of each scene; images[i] is [3, h, w]."""
    frames = [("rgbbox", 19, 24), ("irreg", 33, 16)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_step_worker, args=(r, 2, port, frames, q, 3)) for r in range(2)]
    for p in procs:
        p.start()
    imgs = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for (scene, h, w), img in zip(frames, imgs):
        full, _ = O.OracleScene(scene).render(h, w)
        assert img.shape == (3, h, w)
        for f in range(3):
            assert (img[f] == full).all()


def test_single_rank_path_without_process_group():
    from raytracers_amd.dist import ShardedRenderer
    h, w = 20, 16
    sr = ShardedRenderer(OraclePartRenderer("irreg", h, w), h, w, device="cpu")
    img = sr.render()
    full, _ = O.OracleScene("irreg").render(h, w)
    assert (img.numpy() == full).all()


# ---------------------------------------------------------------- the direct-store exchange, on the CPU ---------
# ShardedStep(exchange="direct"): rank dst owns the images, exports the allocation, the other ranks map it and store their rows
# in place; two one-element signals per step.  Here the "device buffer" is a POSIX shared-memory block, its "IPC handle" the
# block's name, a "device pointer" an integer that each process resolves in its own table -- the product's control flow
# (set-up, broadcast of the handle, fallback when a rank cannot map it, ordering signals, close) runs as it is.
class _ShmBuffer:
    def __init__(self, ctx, count, name=None):
        from multiprocessing import shared_memory
        self.ctx = ctx
        self.shm = shared_memory.SharedMemory(create=name is None, size=max(4, 4 * count), name=name)
        self.arr = np.ndarray((count,), dtype=np.int32, buffer=self.shm.buf)
        self.ptr = ctx.register(self)
        self.owner = name is None

    def as_torch(self, shape):
        return torch.from_numpy(self.arr[:int(np.prod(shape))]).view(*shape)

    def free(self):
        self.arr = None
        try:
            self.shm.close()
            if self.owner:
                self.shm.unlink()
        except (FileNotFoundError, BufferError):
            pass


class _ShmCtx:
    """stands in for api.Context + the rt_ipc_* entry points"""

    def __init__(self, fail_import=False):
        self.table, self.next, self.fail_import = {}, 1 << 40, fail_import

    def register(self, buf):
        base = self.next
        self.next += 1 << 36
        self.table[base] = buf
        return base

    def alloc_i32(self, count):
        return _ShmBuffer(self, count)

    def resolve(self, ptr):
        base = max(b for b in self.table if b <= ptr)
        return self.table[base].arr, (ptr - base) // 4


class _OracleInplaceRenderer(OraclePartRenderer):
    def __init__(self, scene, h, w, ctx):
        super().__init__(scene, h, w)
        self.ctx = ctx

    def inplace(self, part, nparts, nbatch, image_ptr, frame_stride):
        from raytracers_amd.dist import tile_rows
        arr, off = self.ctx.resolve(image_ptr)
        for f in range(nbatch):
            img = arr[off + f * frame_stride: off + f * frame_stride + self.h * self.w].reshape(self.h, self.w)
            rows = tile_rows(self.h, part, nparts)
            for t0 in range(0, len(rows), 8):
                band = rows[t0:t0 + 8]
                px, _ = self.sc.render(self.h, self.w, rows=(int(band[0]), int(band[-1]) + 1), threads=1)
                img[band] = px


def _direct_worker(rank, world, port, frames, q, nbatch, fail_rank):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import raytracers_amd.dist as D
        ctx = _ShmCtx(fail_import=(rank == fail_rank))
        D.api.ipc_export = lambda c, ptr: c.table[ptr].shm.name.encode().ljust(64, b"\0")

        def ipc_import(c, handle):
            if c.fail_import:
                raise RuntimeError("this rank cannot map the buffer")
            return _ShmBuffer(c, 0, name=handle.rstrip(b"\0").decode()).ptr
        D.api.ipc_import = ipc_import
        D.api.ipc_close = lambda c, ptr: c.table.pop(ptr).free()
        # (a mapped block's array covers the whole block: rebuild it at the block's real size)
        orig = _ShmBuffer.__init__

        def init(self, c, count, name=None):
            orig(self, c, count, name)
            if name is not None:
                self.arr = np.ndarray((self.shm.size // 4,), dtype=np.int32, buffer=self.shm.buf)
        _ShmBuffer.__init__ = init
        st = D.ShardedStep([(_OracleInplaceRenderer(s, h, w, ctx), h, w) for s, h, w in frames], device="cpu", nbatch=nbatch, exchange="direct")
        mode = st.exchange_mode
        if rank == 0 and mode == "direct":
            for im in st.images:
                im.fill_(-5)
        imgs = st.render()
        imgs = st.render()       # the buffers are reused across steps
        if rank == 0:
            q.put((mode, [i.numpy().copy() for i in imgs]))
        else:
            assert imgs is None
        st.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nbatch,fail_rank", [(2, 1, -1), (3, 2, -1), (2, 1, 1)])
def test_direct_store_exchange_control_flow(world, nbatch, fail_rank):
    """exchange="direct" with world 2 / 3 over gloo: every rank stores its cyclic row tiles in place into rank 0's (here:
    shared-memory) images -- single frames and a batch -- and when one rank cannot map the buffer ALL ranks fall back to
    the gather, with the same images."""
    frames = [("rgbbox", 27, 24), ("irreg", 41, 32)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000) + 3 * world + nbatch + (7 if fail_rank >= 0 else 0)
    procs = [ctx.Process(target=_direct_worker, args=(r, world, port, frames, q, nbatch, fail_rank)) for r in range(world)]
    for p in procs:
        p.start()
    mode, imgs = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert mode == ("gather" if fail_rank >= 0 else "direct")
    for (scene, h, w), img in zip(frames, imgs):
        full, _ = O.OracleScene(scene).render(h, w)
        for f in (img if nbatch > 1 else [img]):
            assert (f == full).all()
