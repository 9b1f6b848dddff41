"""ctypes binding of librt3.so (include/rt3.h).  There is no CPU fallback: if the HIP library is missing or no gfx950
device is visible, every entry point raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("RT3_LIBRARY", PKG / "librt3.so"))  # RT3_LIBRARY: another build of the same ABI (kernel experiments)

RT3_OK = 0
E_INVALID, E_HIP, E_NO_DEVICE, E_STATE, E_UNSUPPORTED, E_DEPTH, E_COMM = -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128
TAG_BUFFER, TAG_IMAGE, TAG_ACCEL = 0, 1, 3
MISS = 0xFFFFFFFF
BACKGROUND_DEPTH = 100000.0
FORMAT_R32_SFLOAT, FORMAT_R32G32B32A32_SFLOAT, FORMAT_R32G32B32A32_UINT, FORMAT_R8G8B8A8_UNORM, FORMAT_R16_UINT = 100, 109, 107, 37, 74
F_NEE_SKY, F_BLUENOISE, F_SPECULAR, F_FACEFORWARD, F_PROBE_RADIANCE = 1, 2, 4, 8, 16
OPT_BATCH_SPP, OPT_PROFILE, OPT_COUNT_TRAVERSAL, OPT_EXTEND_VARIANT, OPT_LEAF_SIZE, OPT_NODE_WIDTH, OPT_NODE_QUANT = 1, 2, 3, 4, 5, 6, 7
OPT_WIDE_COLLAPSE, OPT_POOL_CHUNK, OPT_FUSED_TRACE, OPT_SAH_TOP, OPT_TRACE_BLOCKS, OPT_SAH_TOP_DEVICE = 8, 9, 10, 11, 12, 13

EXPORTS = [
    "rt3_create", "rt3_destroy", "rt3_last_error", "rt3_device_name", "rt3_set_option",
    "rt3_scene_set_vertices", "rt3_scene_set_indices", "rt3_scene_set_geometry", "rt3_scene_set_sky", "rt3_scene_set_bluenoise", "rt3_scene_set_texture",
    "rt3_scene_set_instances",
    "rt3_accel_build", "rt3_accel_info", "rt3_accel_download", "rt3_accel_import", "rt3_sky_download",
    "rt3_buffer_create", "rt3_image_create", "rt3_image_import", "rt3_resource_upload", "rt3_resource_download", "rt3_resource_device_ptr",
    "rt3_set_tile_partition", "rt3_tile_pixel_count", "rt3_image_pack_tiles", "rt3_image_unpack_tiles",
    "rt3_comm_version", "rt3_comm_unique_id", "rt3_comm_init", "rt3_comm_destroy", "rt3_gather_tiles", "rt3_gather_layout", "rt3_gather_unpack",
    "rt3_pass_launch", "rt3_frame_wait", "rt3_trace_rays", "rt3_selftest_eval", "rt3_stats_reset", "rt3_stats_get", "rt3_camera_gconst",
]


class GConst(C.Structure):
    """src/renderer/mod.rs:47-63 / shaders/include/datatypes.slang:28-43 (304 bytes, column-major matrices)."""

    _fields_ = [("proj", C.c_float * 16), ("view", C.c_float * 16), ("proj_inverse", C.c_float * 16), ("view_inverse", C.c_float * 16),
                ("window_size", C.c_float * 2), ("frame", C.c_uint32), ("blendfactor", C.c_float), ("bounces", C.c_uint32),
                ("samples", C.c_uint32), ("proberng", C.c_uint32), ("cell_size", C.c_float), ("mouse", C.c_uint32 * 2), ("pad", C.c_uint32 * 2)]


class Stats(C.Structure):
    _fields_ = [("extension_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("shadow_nodes_visited", C.c_uint64), ("shadow_tris_tested", C.c_uint64),
                ("extend_launches", C.c_uint64), ("extend_ms", C.c_double), ("shadow_launches", C.c_uint64), ("shadow_ms", C.c_double),
                ("shade_ms", C.c_double), ("other_ms", C.c_double),
                ("trace_launches", C.c_uint64), ("trace_ms", C.c_double), ("trace_rays", C.c_uint64 * 2), ("trace_nodes", C.c_uint64 * 2),
                ("trace_tris", C.c_uint64 * 2), ("gather_ms", C.c_double), ("nodes_visited_lds", C.c_uint64), ("shadow_nodes_visited_lds", C.c_uint64), ("accel_build_ms", C.c_double), ("accel_bulk_copies", C.c_uint64)]


class Instance(C.Structure):
    """rt3_instance: Instance{model} + Transform{Mat4} (src/renderer/world/mod.rs:34-60); transform column-major like glam's Mat4."""

    _fields_ = [("geometry_first", C.c_uint32), ("geometry_count", C.c_uint32), ("transform", C.c_float * 16)]


assert C.sizeof(GConst) == 304


def kernel_source_hash() -> str:
    """sha256 (16 hex digits) over the sources librt3.so is built from.  tools/summarize_profile.py writes it into a counter profile,
    bench.py quotes a profile only when it matches the tree it runs from: counters of other kernels are not evidence for these."""
    import hashlib

    h = hashlib.sha256()
    files = sorted((PKG / "csrc").glob("*.hip")) + sorted((PKG / "csrc").glob("*.hpp")) + [PKG / "csrc" / "Makefile", PKG.parent / "include" / "rt3.h"]
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


class Rt3Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt3 error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load librt3.so (built by __graft_entry__.build() / `make -C raytracer3_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  Import it first so that the
    # loader resolves librt3's NEEDED libamdhip64.so.7 to that already-loaded copy: two HIP runtimes in one process
    # cannot both open the GPU ("No HIP GPUs are available").  Without torch, librt3 uses /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {PKG / 'csrc'}` (hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(str(LIB_PATH))
    vp, u32, i32, f32, sz = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_size_t
    pu32 = C.POINTER(C.c_uint32)
    sig = {
        "rt3_create": (i32, [i32, C.POINTER(vp)]),
        "rt3_destroy": (None, [vp]),
        "rt3_last_error": (C.c_char_p, [vp]),
        "rt3_device_name": (i32, [vp, C.c_char_p, sz]),
        "rt3_set_option": (i32, [vp, i32, C.c_int64]),
        "rt3_scene_set_vertices": (i32, [vp, vp, u32]),
        "rt3_scene_set_indices": (i32, [vp, vp, u32]),
        "rt3_scene_set_geometry": (i32, [vp, vp, vp, u32]),
        "rt3_scene_set_sky": (i32, [vp, vp, u32, u32]),
        "rt3_scene_set_bluenoise": (i32, [vp, vp, u32, u32]),
        "rt3_scene_set_texture": (i32, [vp, u32, vp, u32, u32]),
        "rt3_scene_set_instances": (i32, [vp, vp, u32]),
        "rt3_accel_build": (i32, [vp, pu32]),
        "rt3_accel_info": (i32, [vp, pu32, pu32, pu32, pu32]),
        "rt3_accel_download": (i32, [vp, vp, sz, vp, sz]),
        "rt3_accel_import": (i32, [vp, vp, sz, vp, sz]),
        "rt3_sky_download": (i32, [vp, vp, vp, vp, vp]),
        "rt3_buffer_create": (i32, [vp, sz, pu32]),
        "rt3_image_create": (i32, [vp, u32, u32, u32, pu32]),
        "rt3_image_import": (i32, [vp, vp, u32, u32, u32, pu32]),
        "rt3_resource_upload": (i32, [vp, u32, vp, sz]),
        "rt3_resource_download": (i32, [vp, u32, vp, sz]),
        "rt3_resource_device_ptr": (i32, [vp, u32, C.POINTER(vp), C.POINTER(sz)]),
        "rt3_set_tile_partition": (i32, [vp, u32, u32, u32, u32]),
        "rt3_tile_pixel_count": (i32, [vp, u32, u32, pu32]),
        "rt3_image_pack_tiles": (i32, [vp, u32, u32, u32, vp]),
        "rt3_image_unpack_tiles": (i32, [vp, u32, u32, u32, vp]),
        "rt3_comm_version": (i32, [C.POINTER(i32)]),
        "rt3_comm_unique_id": (i32, [vp]),
        "rt3_comm_init": (i32, [vp, vp, u32, u32]),
        "rt3_comm_destroy": (i32, [vp]),
        "rt3_gather_tiles": (i32, [vp, u32, u32]),
        "rt3_gather_layout": (i32, [vp, u32, u32, u32, C.POINTER(C.c_uint64)]),
        "rt3_gather_unpack": (i32, [vp, u32, u32, u32, vp]),
        "rt3_pass_launch": (i32, [vp, C.c_char_p, C.c_char_p, u32, u32, u32, vp, sz, pu32, u32]),
        "rt3_frame_wait": (i32, [vp]),
        "rt3_trace_rays": (i32, [vp, vp, u32, i32, vp, vp, vp, vp, vp, vp, i32, C.POINTER(C.c_double)]),
        "rt3_selftest_eval": (i32, [vp, i32, vp, u32, vp]),
        "rt3_stats_reset": (i32, [vp]),
        "rt3_stats_get": (i32, [vp, C.POINTER(Stats)]),
        "rt3_camera_gconst": (None, [C.POINTER(f32), C.POINTER(f32), f32, f32, f32, f32, f32, f32, C.POINTER(GConst)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here == a symbol include/rt3.h declares is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
