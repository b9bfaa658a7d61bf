"""ctypes loader for libray_mi355x.so (the HIP library; include/rt_mi355x.h, include/ray.h).

There is deliberately no fallback: if the shared library is missing or cannot be loaded the
import of this module raises, and if no HIP device is usable Context() raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libray_mi355x.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)

# every symbol include/rt_mi355x.h declares
RT_SYMBOLS = [
    "rt_context_create", "rt_device_count", "rt_context_create_multi", "rt_context_num_devices", "rt_context_gather_mode", "rt_context_rccl_ranks",
    "rt_context_destroy", "rt_last_error", "rt_context_last_launch", "rt_context_sync", "rt_context_set_variant",
    "rt_context_set_option", "rt_context_device_info",
    "rt_scene_rgbbox", "rt_scene_irreg", "rt_scene_floor", "rt_scene_from_spheres", "rt_scene_num_spheres",
    "rt_scene_free",
    "rt_prepare_scene", "rt_prepared_free", "rt_prepared_num_spheres", "rt_prepared_height", "rt_prepared_get_bvh",
    "rt_prepared_get_camera",
    "rt_render", "rt_render_part", "rt_render_image", "rt_render_batch", "rt_render_part_inplace", "rt_ipc_export", "rt_ipc_import", "rt_ipc_close", "rt_part_rows", "rt_place_part", "rt_place_parts", "rt_place_parts_strided", "rt_place_parts_batch", "rt_render_stats", "rt_render_trace",
    "rt_render_timed",
    "rt_device_alloc", "rt_device_free", "rt_copy_to_host",
]
# every symbol include/ray.h declares (the Futhark-shaped drop-in boundary)
FUTHARK_SYMBOLS = [
    "futhark_context_config_new", "futhark_context_config_free", "futhark_context_config_set_debugging",
    "futhark_context_config_set_logging", "futhark_context_config_set_profiling",
    "futhark_context_config_set_device", "futhark_context_config_set_num_threads", "futhark_context_config_set_cache_file",
    "futhark_context_new", "futhark_context_free", "futhark_context_get_error", "futhark_context_sync",
    "futhark_context_report", "futhark_context_clear_caches", "futhark_context_pause_profiling",
    "futhark_context_unpause_profiling",
    "futhark_entry_rgbbox", "futhark_entry_irreg", "futhark_entry_prepare_scene", "futhark_entry_render",
    "futhark_values_i32_2d", "futhark_free_i32_2d", "futhark_shape_i32_2d", "futhark_new_i32_2d", "futhark_values_raw_i32_2d",
    "futhark_free_opaque_prepared_scene", "futhark_free_opaque_scene",
]


def _load():
    # When torch is installed, import it FIRST: it ships its own libamdhip64.so.7 and the
    # process must have exactly one HIP runtime (ours resolves to the already-loaded soname).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP library first (`make` at the repo root, or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). raytracers_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    pf = C.POINTER(C.c_float)
    sig = {
        "rt_context_create": (C.c_int, [C.POINTER(vp), C.c_int, vp, C.c_int]),
        "rt_device_count": (C.c_int, []),
        "rt_context_create_multi": (C.c_int, [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]),
        "rt_context_num_devices": (C.c_int, [vp]),
        "rt_context_gather_mode": (C.c_char_p, [vp]),
        "rt_context_rccl_ranks": (C.c_int, [vp]),
        "rt_context_destroy": (None, [vp]),
        "rt_last_error": (C.c_char_p, [vp]),
        "rt_context_last_launch": (C.c_char_p, [vp]),
        "rt_context_sync": (C.c_int, [vp]),
        "rt_context_set_variant": (C.c_int, [vp, C.c_int]),
        "rt_context_set_option": (C.c_int, [vp, C.c_char_p, i64]),
        "rt_context_device_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                             C.c_char_p, C.c_int]),
        "rt_scene_rgbbox": (C.c_int, [vp, C.POINTER(vp)]),
        "rt_scene_irreg": (C.c_int, [vp, C.POINTER(vp)]),
        "rt_scene_floor": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_float]),
        "rt_scene_from_spheres": (C.c_int, [vp, C.POINTER(vp), vp, i64, pf, pf, C.c_float]),
        "rt_scene_num_spheres": (i64, [vp]),
        "rt_scene_free": (C.c_int, [vp, vp]),
        "rt_prepare_scene": (C.c_int, [vp, C.POINTER(vp), i64, i64, vp]),
        "rt_prepared_free": (C.c_int, [vp, vp]),
        "rt_prepared_num_spheres": (i64, [vp]),
        "rt_prepared_height": (i32, [vp]),
        "rt_prepared_get_bvh": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
        "rt_prepared_get_camera": (C.c_int, [vp, vp, vp]),
        "rt_render": (C.c_int, [vp, vp, i64, i64, vp]),
        "rt_render_part": (C.c_int, [vp, vp, i64, i64, i32, i32, i32, i32, vp]),
        "rt_render_image": (C.c_int, [vp, vp, i64, i64, vp, i32, i32, i32, i32, vp]),
        "rt_render_batch": (C.c_int, [vp, vp, i64, i64, i32, i32, i32, i32, i32, vp, i64, vp]),
        "rt_render_part_inplace": (C.c_int, [vp, vp, i64, i64, i32, i32, i32, i32, i32, vp, i64, vp]),
        "rt_ipc_export": (C.c_int, [vp, vp, vp]),
        "rt_ipc_import": (C.c_int, [vp, vp, C.POINTER(vp)]),
        "rt_ipc_close": (C.c_int, [vp, vp]),
        "rt_part_rows": (i64, [i64, i32, i32, i32]),
        "rt_place_part": (C.c_int, [vp, i64, i64, i32, i32, i32, vp, vp]),
        "rt_place_parts": (C.c_int, [vp, i64, i64, i32, i32, i64, vp, vp]),
        "rt_place_parts_strided": (C.c_int, [vp, i64, i64, i32, i32, i64, vp, vp]),
        "rt_place_parts_batch": (C.c_int, [vp, i64, i64, i32, i32, i64, i32, i64, i64, vp, vp]),
        "rt_render_stats": (C.c_int, [vp, vp, i64, i64, i32, vp]),
        "rt_render_trace": (C.c_int, [vp, vp, i64, i64, i32, vp, i32, C.POINTER(i32)]),
        "rt_render_timed": (C.c_int, [vp, vp, i64, i64, i32, i32, i32, i32, vp, i32, i32, vp]),
        "rt_device_alloc": (C.c_int, [vp, C.POINTER(vp), i64]),
        "rt_device_free": (C.c_int, [vp, vp]),
        "rt_copy_to_host": (C.c_int, [vp, vp, vp, i64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
