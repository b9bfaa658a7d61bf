"""Host-side mirror of the reference's call surface for the render path,

    scene   = rgbbox() | irreg()                       futhark/ray.fut:176, :223
    prepared = prepare_scene(h, w, scene)              futhark/ray.fut:241-244
    pixels  = render(h, w, prepared)                   futhark/ray.fut:246-247
    pixels  = render_image(objs, width, height, cam)   futhark/ray.fut:166-169

over the C ABI of libray_mi355x.so (include/rt_mi355x.h).  Pixels are the reference's packed
i32 `(r<<16)|(g<<8)|b`, row-major from the top row.  Every render launches HIP kernels on the
context's stream; there is no CPU path.
"""
import ctypes as C

import numpy as np

from ._lib import lib

VARIANT_AUTO, VARIANT_PIXEL, VARIANT_PERSISTENT, VARIANT_POOLED = 0, 1, 2, 3
MAX_DEPTH = 50          # ray.fut:154
ROWS_PER_TILE = 8       # cyclic row-tile height of the multi-GPU partition


class RtError(RuntimeError):
    pass


class Context:
    """One HIP device + one stream.  stream=None: the context owns a private stream.
    Otherwise `stream` is a raw hipStream_t as an int, e.g.
    torch.cuda.current_stream().cuda_stream (0 = the default stream), so that launches order
    with torch work and torch.cuda.Event timing sees them."""

    def __init__(self, device=-1, stream=None, devices=None):
        """devices: a list of HIP device ordinals -> ONE context over several GPUs (rt_context_create_multi):
        whole frames are cut into cyclic row tiles over them and gathered on devices[0]."""
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = lib.rt_context_create_multi(C.byref(h), arr, len(devices))
        else:
            rc = lib.rt_context_create(C.byref(h), int(device), C.c_void_p(stream or 0), 0 if stream is None else 1)
        if rc != 0 or not h.value:
            raise RtError(f"rt_context_create failed (code {rc}): no usable HIP device; "
                          "raytracers_amd has no CPU fallback")
        self._h = h

    # -- plumbing
    def _check(self, rc):
        if rc != 0:
            raise RtError(lib.rt_last_error(self._h).decode() or f"error code {rc}")

    def close(self):
        if getattr(self, "_h", None):
            lib.rt_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(lib.rt_context_sync(self._h))

    @property
    def num_devices(self):
        return int(lib.rt_context_num_devices(self._h))

    @property
    def gather_mode(self):
        return lib.rt_context_gather_mode(self._h).decode()

    @property
    def last_launch(self):
        """what the last render entry enqueued: family, tickets (pixel list / ordered tiles / raster), instantiation, launch shape"""
        return lib.rt_context_last_launch(self._h).decode()

    @property
    def rccl_ranks(self):
        """ranks of the RCCL communicator behind a multi-device context's gather (0: RCCL is not what carries it)"""
        return int(lib.rt_context_rccl_ranks(self._h))

    def set_variant(self, v):
        self._check(lib.rt_context_set_variant(self._h, int(v)))

    def set_option(self, name, value):
        self._check(lib.rt_context_set_option(self._h, name.encode(), int(value)))

    def device_info(self):
        dev, cus, lds = C.c_int(), C.c_int(), C.c_int()
        name = C.create_string_buffer(64)
        self._check(lib.rt_context_device_info(self._h, C.byref(dev), C.byref(cus), C.byref(lds), name, 64))
        return {"device": dev.value, "num_cu": cus.value, "lds_bytes": lds.value, "arch": name.value.decode()}

    # -- scenes
    def _scene(self, fn, *args):
        h = C.c_void_p()
        self._check(fn(self._h, C.byref(h), *args))
        return Scene(self, h)

    def rgbbox(self):
        return self._scene(lib.rt_scene_rgbbox)

    def irreg(self):
        return self._scene(lib.rt_scene_irreg)

    def floor(self, n, k):
        return self._scene(lib.rt_scene_floor, int(n), float(k))

    def scene(self, name):
        if name == "rgbbox":
            return self.rgbbox()
        if name == "irreg":
            return self.irreg()
        if name == "big":
            return self.floor(1000, 6000.0)
        raise ValueError(f"unknown scene {name!r}")

    def scene_from_spheres(self, spheres7, look_from, look_at, fov):
        s = np.ascontiguousarray(spheres7, dtype=np.float32)
        if s.ndim != 2 or s.shape[1] != 7:
            raise ValueError("spheres7 must be (n, 7): pos.xyz, colour.rgb, radius")
        lf = (C.c_float * 3)(*look_from)
        la = (C.c_float * 3)(*look_at)
        h = C.c_void_p()
        self._check(lib.rt_scene_from_spheres(self._h, C.byref(h), s.ctypes.data, s.shape[0], lf, la, float(fov)))
        return Scene(self, h)

    # -- device memory for callers without an allocator of their own
    def alloc_i32(self, count):
        return DeviceBuffer(self, int(count) * 4)


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        ctx._check(lib.rt_device_alloc(ctx._h, C.byref(p), nbytes))
        self.ptr = p.value

    def as_torch(self, shape):
        """zero-copy torch int32 view of the buffer (the buffer must outlive it)"""
        import torch
        n = int(np.prod(shape))
        assert n * 4 <= self.nbytes
        iface = {"shape": tuple(int(x) for x in shape), "typestr": "<i4", "data": (int(self.ptr), False), "version": 2}
        holder = type("_CudaArray", (), {"__cuda_array_interface__": iface})()
        # (the device that owns the buffer, not torch's current one: a context on another GPU would get a mislabelled tensor)
        return torch.as_tensor(holder, device=torch.device("cuda", self.ctx.device_info()["device"]))

    def to_host(self, shape):
        out = np.empty(shape, dtype=np.int32)
        assert out.nbytes <= self.nbytes
        self.ctx._check(lib.rt_copy_to_host(self.ctx._h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr and self.ctx._h:
            lib.rt_device_free(self.ctx._h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Scene:
    def __init__(self, ctx, h):
        self.ctx, self._h = ctx, h

    @property
    def num_spheres(self):
        return int(lib.rt_scene_num_spheres(self._h))

    def free(self):
        if self._h and self.ctx._h:
            lib.rt_scene_free(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Prepared:
    """prepared_scene = {objs: bvh, cam: camera} (ray.fut:239), resident on the device."""

    def __init__(self, ctx, h, height, width):
        self.ctx, self._h, self.h, self.w = ctx, h, int(height), int(width)

    @property
    def num_spheres(self):
        return int(lib.rt_prepared_num_spheres(self._h))

    @property
    def height(self):
        """levels of inner nodes on the longest root-to-leaf path of the BVH"""
        return int(lib.rt_prepared_height(self._h))

    def camera(self):
        cam = np.empty(12, dtype=np.float32)
        self.ctx._check(lib.rt_prepared_get_camera(self.ctx._h, self._h, cam.ctypes.data))
        return cam

    def bvh_arrays(self):
        """Canonical {L, I} of bvh.fut:28 copied back from the device (for parity checks)."""
        n = self.num_spheres
        ni = n - 1
        A = {"L": np.empty((n, 7), np.float32), "bmin": np.empty((ni, 3), np.float32),
             "bmax": np.empty((ni, 3), np.float32), "left": np.empty(ni, np.int32),
             "right": np.empty(ni, np.int32), "parent": np.empty(ni, np.int32)}
        self.ctx._check(lib.rt_prepared_get_bvh(self.ctx._h, self._h, *[A[k].ctypes.data for k in
                                                                       ("L", "bmin", "bmax", "left", "right", "parent")]))
        return A

    def stats(self, max_depth=MAX_DEPTH):
        s = (C.c_uint64 * 3)()
        self.ctx._check(lib.rt_render_stats(self.ctx._h, self._h, self.h, self.w, int(max_depth), s))
        return {"rays": int(s[0]), "box_tests": int(s[1]), "leaf_tests": int(s[2])}

    def free(self):
        if self._h and self.ctx._h:
            lib.rt_prepared_free(self.ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def prepare_scene(h, w, scene):
    """entry prepare_scene h w scene (ray.fut:241): BVH build + camera for an h x w image."""
    ctx = scene.ctx
    p = C.c_void_p()
    ctx._check(lib.rt_prepare_scene(ctx._h, C.byref(p), int(h), int(w), scene._h))
    return Prepared(ctx, p, h, w)


def part_rows(h, part=0, nparts=1, rows_per_tile=ROWS_PER_TILE):
    return int(lib.rt_part_rows(int(h), int(rows_per_tile), int(part), int(nparts)))


def render_into(out_ptr, h, w, prepared, max_depth=MAX_DEPTH, part=0, nparts=1, rows_per_tile=ROWS_PER_TILE, cam=None):
    """Enqueue a render of part `part` of `nparts` into the device pointer `out_ptr`
    (part_rows(h, part, nparts) * w int32).  Asynchronous: ctx.sync() completes it."""
    ctx = prepared.ctx
    if cam is None:
        ctx._check(lib.rt_render_part(ctx._h, prepared._h, int(h), int(w), int(max_depth), int(rows_per_tile),
                                      int(part), int(nparts), C.c_void_p(out_ptr)))
    else:
        c = np.ascontiguousarray(cam, dtype=np.float32)
        assert c.size == 12
        ctx._check(lib.rt_render_image(ctx._h, prepared._h, int(w), int(h), c.ctypes.data, int(max_depth),
                                       int(rows_per_tile), int(part), int(nparts), C.c_void_p(out_ptr)))


def render_batch_into(out_ptr, h, w, prepared, nframes, frame_stride=None, cams=None, max_depth=MAX_DEPTH, part=0, nparts=1,
                      rows_per_tile=ROWS_PER_TILE):
    """`nframes` frames in ONE launch (rt_render_batch): frame f -> out_ptr + 4 * f * frame_stride (default: packed,
    part_rows * w elements apart), traced through cams[f] (nframes x 12 floats) or, cams=None, the prepared camera."""
    ctx = prepared.ctx
    if frame_stride is None:
        frame_stride = part_rows(h, part, nparts, rows_per_tile) * w
    cp = None
    if cams is not None:
        c = np.ascontiguousarray(cams, dtype=np.float32)
        assert c.size == 12 * nframes
        cp = c.ctypes.data
    ctx._check(lib.rt_render_batch(ctx._h, prepared._h, int(h), int(w), int(max_depth), int(rows_per_tile), int(part), int(nparts),
                                   int(nframes), C.c_void_p(cp), int(frame_stride), C.c_void_p(out_ptr)))


def render_inplace_into(image_ptr, h, w, prepared, nframes=1, frame_stride=None, cams=None, max_depth=MAX_DEPTH, part=0, nparts=1,
                        rows_per_tile=ROWS_PER_TILE):
    """Part `part` of `nparts` of `nframes` frames stored IN PLACE (rt_render_part_inplace): image_ptr is the FULL image
    (frame f at image_ptr + 4 * f * frame_stride, default h * w) -- possibly another device's memory (a peer allocation or
    an ipc_import'ed buffer): the pixel stores then are the framebuffer exchange."""
    ctx = prepared.ctx
    cp = None
    if cams is not None:
        c = np.ascontiguousarray(cams, dtype=np.float32)
        assert c.size == 12 * nframes
        cp = c.ctypes.data
    ctx._check(lib.rt_render_part_inplace(ctx._h, prepared._h, int(h), int(w), int(max_depth), int(rows_per_tile), int(part), int(nparts),
                                          int(nframes), C.c_void_p(cp), int(h * w if frame_stride is None else frame_stride),
                                          C.c_void_p(image_ptr)))


def ipc_export(ctx, dev_ptr):
    """64 opaque bytes naming the device allocation at dev_ptr (a DeviceBuffer's pointer) for the other processes of the node"""
    h = (C.c_ubyte * 64)()
    ctx._check(lib.rt_ipc_export(ctx._h, C.c_void_p(dev_ptr), h))
    return bytes(h)


def ipc_import(ctx, handle64):
    """a device pointer to another process's exported allocation (valid until ipc_close)"""
    assert len(handle64) == 64
    h = (C.c_ubyte * 64).from_buffer_copy(handle64)
    p = C.c_void_p()
    ctx._check(lib.rt_ipc_import(ctx._h, h, C.byref(p)))
    return p.value


def ipc_close(ctx, dev_ptr):
    ctx._check(lib.rt_ipc_close(ctx._h, C.c_void_p(dev_ptr)))


def render(h, w, prepared, max_depth=MAX_DEPTH):
    """entry render h w prepared (ray.fut:246): returns the [h][w]i32 image as a numpy array."""
    buf = prepared.ctx.alloc_i32(h * w)
    try:
        render_into(buf.ptr, h, w, prepared, max_depth)
        return buf.to_host((h, w))
    finally:
        buf.free()


def render_image(objs, width, height, cam, max_depth=MAX_DEPTH):
    """render_image objs width height cam (ray.fut:166): explicit camera (12 floats)."""
    buf = objs.ctx.alloc_i32(height * width)
    try:
        render_into(buf.ptr, height, width, objs, max_depth, cam=cam)
        return buf.to_host((height, width))
    finally:
        buf.free()


def render_timed(out_ptr, h, w, prepared, warmup, iters, max_depth=MAX_DEPTH, part=0, nparts=1,
                 rows_per_tile=ROWS_PER_TILE):
    """`iters` back-to-back launches timed with HIP events on the context's stream; returns ms per launch."""
    ctx = prepared.ctx
    ms = np.zeros(iters, dtype=np.float32)
    ctx._check(lib.rt_render_timed(ctx._h, prepared._h, int(h), int(w), int(max_depth), int(rows_per_tile), int(part),
                                   int(nparts), C.c_void_p(out_ptr), int(warmup), int(iters), ms.ctypes.data))
    return ms


def place_part(ctx, h, w, part, nparts, part_ptr, image_ptr, rows_per_tile=ROWS_PER_TILE):
    ctx._check(lib.rt_place_part(ctx._h, int(h), int(w), int(rows_per_tile), int(part), int(nparts),
                                 C.c_void_p(part_ptr), C.c_void_p(image_ptr)))


def place_parts(ctx, h, w, nparts, pad_rows, stacked_ptr, image_ptr, rows_per_tile=ROWS_PER_TILE, part_stride=None):
    """All gathered parts -> the h x w image, one kernel.  Part p's packed rows start at
    stacked_ptr + 4 * p * part_stride (default: pad_rows * w, i.e. nparts x pad_rows x w)."""
    if part_stride is None:
        ctx._check(lib.rt_place_parts(ctx._h, int(h), int(w), int(rows_per_tile), int(nparts), int(pad_rows),
                                      C.c_void_p(stacked_ptr), C.c_void_p(image_ptr)))
    else:
        ctx._check(lib.rt_place_parts_strided(ctx._h, int(h), int(w), int(rows_per_tile), int(nparts), int(part_stride),
                                              C.c_void_p(stacked_ptr), C.c_void_p(image_ptr)))


def place_parts_batch(ctx, h, w, nparts, part_stride, nframes, frame_stride_in, stacked_ptr, images_ptr, frame_stride_out=None,
                      rows_per_tile=ROWS_PER_TILE):
    """All gathered parts of `nframes` frames -> nframes h x w images, ONE kernel (rt_place_parts_batch): inside part p
    (at stacked_ptr + 4 * p * part_stride) frame f's packed rows start f * frame_stride_in elements in; image f at
    images_ptr + 4 * f * frame_stride_out (default h * w)."""
    ctx._check(lib.rt_place_parts_batch(ctx._h, int(h), int(w), int(rows_per_tile), int(nparts), int(part_stride), int(nframes),
                                        int(frame_stride_in), int(h * w if frame_stride_out is None else frame_stride_out),
                                        C.c_void_p(stacked_ptr), C.c_void_p(images_ptr)))
