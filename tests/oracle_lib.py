"""ctypes binding of the CPU oracle (oracle/ray_oracle.c).  TEST INFRASTRUCTURE:
imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import gzip
import io
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "build", "liboracle.so")


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("pos", Vec3), ("colour", Vec3), ("radius", C.c_float)]


class Bvh(C.Structure):
    _fields_ = [("n", C.c_int64), ("L", C.POINTER(Sphere)), ("morton", C.POINTER(C.c_uint32)),
                ("bmin", C.POINTER(C.c_float)), ("bmax", C.POINTER(C.c_float)),
                ("left", C.POINTER(C.c_int32)), ("right", C.POINTER(C.c_int32)),
                ("parent", C.POINTER(C.c_int32))]


class Camera(C.Structure):
    _fields_ = [("origin", Vec3), ("llc", Vec3), ("horizontal", Vec3), ("vertical", Vec3)]


class Scene(C.Structure):
    _fields_ = [("look_from", Vec3), ("look_at", Vec3), ("fov", C.c_float), ("n", C.c_int64),
                ("spheres", C.POINTER(Sphere))]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("steps", C.c_uint64), ("box_tests", C.c_uint64),
                ("leaf_tests", C.c_uint64), ("max_steps", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("ray_oracle.c", "rust_algo_port.c")]
        if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_scene_rgbbox.argtypes = [C.POINTER(Scene)]
        L.orc_scene_irreg.argtypes = [C.POINTER(Scene)]
        L.orc_scene_floor.argtypes = [C.POINTER(Scene), C.c_int, C.c_float]
        L.orc_scene_free.argtypes = [C.POINTER(Scene)]
        L.orc_bvh_build.argtypes = [C.POINTER(Sphere), C.c_int64, C.POINTER(Bvh)]
        L.orc_bvh_free.argtypes = [C.POINTER(Bvh)]
        L.orc_scene_camera.argtypes = [C.POINTER(Scene), C.c_int64, C.c_int64, C.POINTER(Camera)]
        L.orc_render_rows.argtypes = [C.POINTER(Bvh), C.POINTER(Camera), C.c_int64, C.c_int64, C.c_int64,
                                      C.c_int64, C.c_int32, C.c_int, C.c_void_p, C.POINTER(Counters)]
        L.orc_checksum.argtypes = [C.c_void_p, C.c_int64]
        L.orc_checksum.restype = C.c_uint32
        L.orc_num_threads.restype = C.c_int
        L.rust_bvh_build.argtypes = [C.POINTER(Sphere), C.c_int64, C.c_void_p]
        L.rust_bvh_free.argtypes = [C.c_void_p]
        L.rust_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


class OracleScene:
    """scene -> prepare_scene -> render, as the reference's entry points chain them
    (ray.fut:176/223 -> :241 -> :246)."""

    def __init__(self, name, n=None, k=None, spheres7=None, look_from=None, look_at=None, fov=None):
        L = lib()
        self.name = name
        self._owns_spheres = True
        self.scene = Scene()
        if name == "rgbbox":
            rc = L.orc_scene_rgbbox(C.byref(self.scene))
        elif name == "irreg":
            rc = L.orc_scene_irreg(C.byref(self.scene))
        elif name == "big":
            rc = L.orc_scene_floor(C.byref(self.scene), 1000 if n is None else n, 6000.0 if k is None else k)
        elif name == "floor":
            rc = L.orc_scene_floor(C.byref(self.scene), n, k)
        elif name == "custom":
            # spheres7: (n, 7) float32 {pos.xyz, colour.rgb, radius}; memory owned by numpy
            self._spheres = np.ascontiguousarray(spheres7, dtype=np.float32)
            self.scene.spheres = C.cast(self._spheres.ctypes.data, C.POINTER(Sphere))
            self.scene.n = self._spheres.shape[0]
            self.scene.look_from = Vec3(*look_from)
            self.scene.look_at = Vec3(*look_at)
            self.scene.fov = fov
            self._owns_spheres = False
            rc = 0
        else:
            raise ValueError(name)
        assert rc == 0
        self.bvh = Bvh()
        assert L.orc_bvh_build(self.scene.spheres, self.scene.n, C.byref(self.bvh)) == 0
        self.n = int(self.scene.n)

    def camera(self, h, w):
        cam = Camera()
        lib().orc_scene_camera(C.byref(self.scene), h, w, C.byref(cam))
        return cam

    def camera_floats(self, h, w):
        cam = self.camera(h, w)
        return np.frombuffer(bytes(cam), dtype=np.float32).copy()

    def render(self, h, w, max_depth=50, threads=0, rows=None, cam=None):
        """Returns (pixels[h_band, w] int32, counters dict).  cam: 12 floats {origin, llc, horizontal, vertical} instead of the
        camera prepare_scene derives (ray.fut:241-244)."""
        r0, r1 = (0, h) if rows is None else rows
        out = np.empty((r1 - r0, w), dtype=np.int32)
        cnt = Counters()
        cam = self.camera(h, w) if cam is None else Camera.from_buffer_copy(np.ascontiguousarray(cam, dtype=np.float32).tobytes())
        rc = lib().orc_render_rows(C.byref(self.bvh), C.byref(cam), w, h, r0, r1, max_depth, threads,
                                   out.ctypes.data, C.byref(cnt))
        assert rc == 0
        return out, cnt.as_dict()

    # --- canonical arrays for I/L parity checks -------------------------------------
    def arrays(self):
        n = self.n
        ni = n - 1
        b = self.bvh
        sph = np.ctypeslib.as_array(C.cast(b.L, C.POINTER(C.c_float)), shape=(n, 7)).copy()
        return {
            "L": sph,  # pos.xyz, colour.xyz, radius
            "morton": np.ctypeslib.as_array(b.morton, shape=(n,)).copy(),
            "bmin": np.ctypeslib.as_array(b.bmin, shape=(ni, 3)).copy(),
            "bmax": np.ctypeslib.as_array(b.bmax, shape=(ni, 3)).copy(),
            "left": np.ctypeslib.as_array(b.left, shape=(ni,)).copy(),
            "right": np.ctypeslib.as_array(b.right, shape=(ni,)).copy(),
            "parent": np.ctypeslib.as_array(b.parent, shape=(ni,)).copy(),
        }

    def close(self):
        L = lib()
        if self.bvh.n:
            L.orc_bvh_free(C.byref(self.bvh))
        if self.scene.n and self._owns_spheres:
            L.orc_scene_free(C.byref(self.scene))
        self.scene.n = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RustAlgoScene:
    """The reference's RUST algorithm (oracle/rust_algo_port.c): a second CPU baseline for
    timing only -- it renders a different image than the Futhark program."""

    def __init__(self, name):
        self.base = OracleScene(name)
        self.bvh = (C.c_char * 64)()
        assert lib().rust_bvh_build(self.base.scene.spheres, self.base.scene.n, C.byref(self.bvh)) == 0

    def render(self, h, w, threads=0):
        out = np.empty((h, w), dtype=np.int32)
        cam = self.base.camera_floats(h, w)
        rays = C.c_uint64()
        assert lib().rust_render(C.byref(self.bvh), cam.ctypes.data, w, h, threads, out.ctypes.data, C.byref(rays)) == 0
        return out, int(rays.value)

    def __del__(self):
        try:
            lib().rust_bvh_free(C.byref(self.bvh))
        except Exception:
            pass


def checksum(px):
    px = np.ascontiguousarray(px, dtype=np.int32)
    return int(lib().orc_checksum(px.ctypes.data, px.size))


def load_golden(scene):
    path = os.path.join(ROOT, "tests", "golden", scene + "_500.npy.gz")
    with gzip.open(path, "rb") as f:
        return np.load(io.BytesIO(f.read()))
