"""Multi-GPU rendering: cyclic row-tile partition + framebuffer gather.

Pixels are independent (tabulate_2d, futhark/ray.fut:169), so the path shards by image
rows with NO data-path collective; the only exchange is the final framebuffer gather to
rank 0 (RCCL over xGMI when the backend is "nccl").  The partition is cyclic over tiles of
ROWS_PER_TILE rows -- contiguous bands would give irreg's 8 ranks 0.1 % ... 25 % of the
work each.  The BVH and camera are replicated: every rank builds them from
the same scene description.

One process per GPU (torch.distributed); the renderer of a part is injected so that the
sharding/gather logic is testable on CPU with the gloo backend.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import api


def tile_rows(h, part, nparts, rows_per_tile=api.ROWS_PER_TILE):
    """Image rows owned by `part`, in the packed order rt_render_part writes them."""
    ntiles = (h + rows_per_tile - 1) // rows_per_tile
    rows = [np.arange(t * rows_per_tile, min(h, (t + 1) * rows_per_tile)) for t in range(part, ntiles, nparts)]
    return np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)


def max_part_rows(h, nparts, rows_per_tile=api.ROWS_PER_TILE):
    return max(len(tile_rows(h, p, nparts, rows_per_tile)) for p in range(nparts))


class HipPartRenderer:
    """Renders this rank's rows with the HIP library into a torch int32 CUDA tensor, on
    torch's current stream (so torch.distributed collectives order after the kernel)."""

    def __init__(self, scene_name, h, w, device, max_depth=api.MAX_DEPTH, variant=api.VARIANT_AUTO, options=None):
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx = api.Context(self.device.index, stream)
        self.ctx.set_variant(variant)
        for k, v in (options or {}).items():
            self.ctx.set_option(k, v)
        self.scene = self.ctx.scene(scene_name)
        self.prepared = api.prepare_scene(h, w, self.scene)
        self.h, self.w, self.max_depth = h, w, max_depth

    def __call__(self, part, nparts, out=None):
        rows = api.part_rows(self.h, part, nparts)
        if out is None:
            out = torch.empty((rows, self.w), dtype=torch.int32, device=self.device)
        assert out.is_cuda and out.dtype == torch.int32 and out.is_contiguous() and out.numel() >= rows * self.w
        api.render_into(out.data_ptr(), self.h, self.w, self.prepared, self.max_depth, part, nparts)
        return out

    def batch(self, part, nparts, nbatch, out, frame_stride):
        """`nbatch` frames of this rank's rows in ONE launch (rt_render_batch): frame f at out + f * frame_stride elements."""
        assert out.is_cuda and out.dtype == torch.int32 and out.is_contiguous() and out.numel() >= nbatch * frame_stride
        api.render_batch_into(out.data_ptr(), self.h, self.w, self.prepared, nbatch, frame_stride=frame_stride,
                              max_depth=self.max_depth, part=part, nparts=nparts)
        return out

    def inplace(self, part, nparts, nbatch, image_ptr, frame_stride):
        """this rank's rows of `nbatch` frames stored at their places in the FULL image(s) at device pointer image_ptr
        (rt_render_part_inplace) -- possibly another rank's buffer (ipc_import): the stores are the exchange"""
        api.render_inplace_into(image_ptr, self.h, self.w, self.prepared, nframes=nbatch, frame_stride=frame_stride,
                                max_depth=self.max_depth, part=part, nparts=nparts)

    def place(self, part, nparts, part_tensor, image):
        api.place_part(self.ctx, self.h, self.w, part, nparts, part_tensor.data_ptr(), image.data_ptr())

    def place_batch(self, nparts, part_stride, nframes, frame_stride_in, stacked, images):
        """the parts of `nframes` frames (frame f's rows frame_stride_in elements into every part) -> images [nframes, h, w]:
        one kernel"""
        api.place_parts_batch(self.ctx, self.h, self.w, nparts, part_stride, nframes, frame_stride_in, stacked.data_ptr(),
                              images.data_ptr())

    def place_all(self, nparts, pad_rows, stacked, image, part_stride=None):
        """stacked: a tensor whose element 0 is part 0's first pixel; part p starts part_stride
        int32 elements further (default pad_rows * w)."""
        api.place_parts(self.ctx, self.h, self.w, nparts, pad_rows, stacked.data_ptr(), image.data_ptr(),
                        part_stride=part_stride)


class ShardedStep:
    """One step = one frame of each of several scenes, across the ranks of a torch.distributed
    group: every rank renders its cyclic row tiles of every frame into ONE send buffer, ONE
    gather moves them to rank `dst`, which assembles the [h][w]i32 images (one kernel each).

    frames: list of (part_renderer, h, w); part_renderer(part, nparts, out) must fill
    out[:part_rows] (out: [pad_rows, w] int32 on `device`) and may be asynchronous on the
    device's current stream.

    nbatch > 1: the step covers `nbatch` frames of EACH scene -- a renderer with a `.batch` method renders its
    nbatch frames in one launch (rt_render_batch), the single gather carries all of them, and `images[i]` is
    an [nbatch, h, w] tensor.

    exchange: "gather" -- the framebuffer gather described above (backend nccl: RCCL over xGMI), or "direct" -- NO gather:
    rank dst owns the images in a buffer it exports to the other ranks (IPC mapping, rt_ipc_export / rt_ipc_import), and
    every rank's kernel stores its pixels straight into them (rt_render_part_inplace) while it traces, over xGMI.  What is
    left of the exchange is two one-element all-reduces per step: "the image may be overwritten" (dst has enqueued its
    readers of the previous step's image; skipped with presync=False when the caller orders that itself) and "every rank's
    stores have landed".  Needs renderers with an `.inplace` method and a `.ctx`; any rank failing to map the buffer puts
    ALL ranks back on "gather" (`exchange_mode` says which one runs)."""

    def __init__(self, frames, device, group=None, dst=0, nbatch=1, exchange="gather", presync=True):
        self.frames = list(frames)
        self.group, self.dst = group, dst
        self.nbatch = int(nbatch)
        self.device = torch.device(device)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.presync = bool(presync)
        self.exchange_mode = "gather"
        self.exchange_note = None
        self._ipc_buf = None      # dst: the DeviceBuffer behind the images
        self._ipc_base = None     # other ranks: the mapped pointer
        if exchange == "direct" and (self.world > 1 or (dist.is_initialized() and os.environ.get("RT_FORCE_GATHER"))):
            if self._setup_direct():
                return
        self.pad_rows = [max_part_rows(h, self.world) for _, h, _ in self.frames]
        sizes = [self.nbatch * pr * w for pr, (_, _, w) in zip(self.pad_rows, self.frames)]
        self.offs = [int(o) for o in np.concatenate([[0], np.cumsum(sizes)])]   # int32 elements
        self.total = self.offs[-1]
        self.recv_all = None
        self.recv = None
        self.images = None
        if self.rank == dst:
            self.images = [torch.empty((h, w) if self.nbatch == 1 else (self.nbatch, h, w), dtype=torch.int32, device=self.device)
                           for _, h, w in self.frames]
        # A CPU-only backend (gloo) cannot move device memory: stage the gather through host
        # buffers then.  Only used to exercise the multi-rank control flow on a one-GPU box
        # (several ranks sharing cuda:0); the product path is backend "nccl" = RCCL over xGMI.
        self.host_staged = (self.device.type == "cuda" and dist.is_initialized()
                            and dist.get_backend(group) == "gloo")
        # RT_FORCE_GATHER=1 keeps the gather + assembly path on even for a single rank (used to
        # exercise the RCCL path on a one-GPU box)
        self.direct = self.world == 1 and not (dist.is_initialized() and os.environ.get("RT_FORCE_GATHER"))
        if self.direct:
            self.outs = self.images            # one part == the whole image: nothing to gather or assemble
        else:
            self.send = torch.zeros(self.total, dtype=torch.int32, device=self.device)
            self.outs = [self.send[self.offs[i]:self.offs[i + 1]].view(self.nbatch * self.pad_rows[i], w)
                         for i, (_, _, w) in enumerate(self.frames)]
            if self.rank == dst:
                # one contiguous buffer; gather_list entries are views of it so that a single
                # kernel per frame can scatter all parts into the image
                self.recv_all = torch.empty((self.world, self.total), dtype=torch.int32, device=self.device)
                self.recv = [self.recv_all[p] for p in range(self.world)]

    # ---- exchange = "direct": the images live on rank dst, every rank stores into them ----
    def _setup_direct(self):
        r0 = self.frames[0][0]
        if not all(hasattr(fr[0], "inplace") and hasattr(fr[0], "ctx") for fr in self.frames):
            self.exchange_note = "renderers without .inplace: gather"
            return False
        sizes = [self.nbatch * h * w for _, h, w in self.frames]
        self.img_offs = [int(o) for o in np.concatenate([[0], np.cumsum(sizes)])]
        handle, ok, err = None, 1, ""
        # Every rank makes the same two collectives whatever fails locally: rank dst's allocation / export failing must not
        # keep it out of the broadcast the other ranks are already waiting in (it then broadcasts None, which they read as
        # "no buffer"), and every rank reaches the all-gather of the outcomes below.
        if self.rank == self.dst:
            try:
                self._ipc_buf = r0.ctx.alloc_i32(self.img_offs[-1])
                handle = api.ipc_export(r0.ctx, self._ipc_buf.ptr)
            except Exception as e:   # noqa: BLE001 -- whatever went wrong, every rank must learn of it
                handle, ok, err = None, 0, f"rank {self.rank}: {e}"
        box = [handle]
        dist.broadcast_object_list(box, src=self.dst, group=self.group)
        if self.rank != self.dst:
            try:
                if box[0] is None:
                    raise RuntimeError(f"rank {self.dst} exported no buffer")
                self._ipc_base = api.ipc_import(r0.ctx, box[0])
            except Exception as e:   # noqa: BLE001
                ok, err = 0, f"rank {self.rank}: {e}"
        oks = [None] * self.world
        dist.all_gather_object(oks, (ok, err), group=self.group)
        if not all(o for o, _ in oks):
            self.exchange_note = "direct stores unavailable (" + "; ".join(e for o, e in oks if not o) + "): gather"
            # the ranks that DID map rank dst's buffer unmap it before its owner frees it (the order close() keeps): every rank
            # is here -- the all-gather above was collective -- so the barrier is too
            if self.rank != self.dst:
                self._release_direct()
            dist.barrier(group=self.group)
            self._release_direct()
            return False
        self.exchange_mode = "direct"
        self.direct = False
        self.host_staged = False
        self.cpu_backend = dist.get_backend(self.group) == "gloo"
        self.images = None
        if self.rank == self.dst:
            full = self._ipc_buf.as_torch((self.img_offs[-1],))
            self.images = [full[self.img_offs[i]:self.img_offs[i + 1]].view((h, w) if self.nbatch == 1 else (self.nbatch, h, w))
                           for i, (_, h, w) in enumerate(self.frames)]
            self.send = full   # (what a caller poisons: the whole buffer)
        else:
            self.send = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.flag = torch.zeros(1, dtype=torch.int32, device="cpu" if self.cpu_backend else self.device)
        return True

    def _release_direct(self):
        try:
            if self._ipc_base is not None:
                api.ipc_close(self.frames[0][0].ctx, self._ipc_base)
        except Exception:   # noqa: BLE001
            pass
        self._ipc_base = None
        if self._ipc_buf is not None:
            self._ipc_buf.free()
        self._ipc_buf = None

    def close(self):
        """direct mode: unmap (other ranks) before the owner frees; collective"""
        if self.exchange_mode != "direct":
            return
        self.images = None
        self.send = None            # (rank dst: a view of the buffer that is about to be freed)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        if self.rank != self.dst:
            self._release_direct()
        if dist.is_initialized():
            dist.barrier(group=self.group)
        self._release_direct()
        self.exchange_mode = "closed"

    def _signal(self):
        """one element all-reduced on the current stream: behind it every rank's earlier work on ITS stream has completed
        (gloo test mode: a host barrier behind a device sync)"""
        if self.cpu_backend:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)
        else:
            dist.all_reduce(self.flag, group=self.group)

    def _render_direct(self, events, gather_events):
        if gather_events is not None:
            gather_events[0].record()
        if self.presync:
            self._signal()          # rank dst's readers of the previous image are enqueued: it may be overwritten
        base = self._ipc_buf.ptr if self.rank == self.dst else self._ipc_base
        for i, (render_part, h, w) in enumerate(self.frames):
            if events is not None and events[i] is not None:
                events[i][0].record()
            render_part.inplace(self.rank, self.world, self.nbatch, base + 4 * self.img_offs[i], h * w)
            if events is not None and events[i] is not None:
                events[i][1].record()
        self._signal()              # every rank's stores have landed
        if gather_events is not None:
            gather_events[1].record()
        return self.images

    def render(self, events=None, gather_events=None):
        """One step.  Returns the list of full image tensors on rank dst, None elsewhere.
        `events`: optional list (one per frame) of (start, end) torch.cuda.Event pairs recorded
        around this rank's kernel of that frame; `gather_events`: an optional (start, end) pair
        recorded around the exchange (gather + assembly on rank dst)."""
        if self.exchange_mode == "direct":
            return self._render_direct(events, gather_events)
        for i, (render_part, _, _) in enumerate(self.frames):
            if events is not None and events[i] is not None:
                events[i][0].record()
            if self.nbatch == 1:
                render_part(self.rank, self.world, self.outs[i])
            elif hasattr(render_part, "batch"):
                render_part.batch(self.rank, self.world, self.nbatch, self.outs[i], self.pad_rows[i] * self.frames[i][2])
            else:
                o = self.outs[i].view(self.nbatch, -1, self.frames[i][2])
                for f in range(self.nbatch):
                    render_part(self.rank, self.world, o[f])
            if events is not None and events[i] is not None:
                events[i][1].record()
        if self.direct:
            return self.images
        if gather_events is not None:
            gather_events[0].record()
        try:
            return self._exchange()
        finally:
            if gather_events is not None:
                gather_events[1].record()

    def _exchange(self):
        if self.host_staged:
            send_h = self.send.cpu()      # synchronises with the renders on the current stream
            recv_h = [torch.empty_like(send_h) for _ in range(self.world)] if self.rank == self.dst else None
            dist.gather(send_h, recv_h, dst=self.dst, group=self.group)
            if self.rank != self.dst:
                return None
            self.recv_all.copy_(torch.stack(recv_h))
        else:
            dist.gather(self.send, self.recv if self.rank == self.dst else None, dst=self.dst, group=self.group)
            if self.rank != self.dst:
                return None
        for i in range(len(self.frames)):
            self._assemble(i)
        return self.images

    def _assemble(self, i):
        render_part, h, w = self.frames[i]
        per = self.pad_rows[i] * w
        if self.nbatch > 1 and self.images[i].is_cuda and hasattr(render_part, "place_batch"):
            # every frame of the batch in ONE launch (a launch per frame was ~0.2 ms of serial tail behind a gather at K = 20)
            render_part.place_batch(self.world, self.total, self.nbatch, per, self.recv_all[:, self.offs[i]:], self.images[i])
            return
        for f in range(self.nbatch):
            image = self.images[i] if self.nbatch == 1 else self.images[i][f]
            stacked = self.recv_all[:, self.offs[i] + f * per:self.offs[i] + (f + 1) * per]
            if image.is_cuda and hasattr(render_part, "place_all"):
                render_part.place_all(self.world, self.pad_rows[i], stacked, image, part_stride=self.total)
                continue
            for p in range(self.world):
                n = api.part_rows(h, p, self.world)
                if n:
                    idx = torch.as_tensor(tile_rows(h, p, self.world), device=image.device)
                    image[idx] = stacked[p].view(self.pad_rows[i], w)[:n]


class ShardedRenderer(ShardedStep):
    """render(h, w) of ONE frame across the ranks (a ShardedStep with a single frame)."""

    def __init__(self, part_renderer, h, w, device, group=None, dst=0):
        super().__init__([(part_renderer, h, w)], device, group=group, dst=dst)
        self.render_part, self.h, self.w = part_renderer, h, w

    @property
    def image(self):
        return self.images[0] if self.images is not None else None

    def render(self, events=None):
        """One frame.  Returns the full image tensor on rank dst, None elsewhere.  `events`:
        an optional (start, end) pair of torch.cuda.Event recorded around this rank's kernel."""
        out = super().render([events] if events is not None else None)
        return out[0] if out is not None else None
