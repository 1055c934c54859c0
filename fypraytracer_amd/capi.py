"""ctypes binding of the C ABI in include/fyprt.h (libfyprt.so).

The product path has no CPU fallback: if the HIP library is missing or cannot be loaded
this module raises, loudly.  Nothing here imports or touches oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "csrc" / "libfyprt.so"

# technique ids: SamplingTechniqueEnum.h:4-17
BRUTE_FORCE, UNIFORM_SAMPLING, COSINE_WEIGHTED_SAMPLING, GGX_SAMPLING, BRDF_SAMPLING = 0, 1, 2, 3, 4
LIGHT_SOURCE_SAMPLING, NEE, RESTIR_DI, RESTIR_GI = 5, 6, 7, 8
TECHNIQUE_NAMES = ["BRUTE_FORCE", "UNIFORM_SAMPLING", "COSINE_WEIGHTED_SAMPLING", "GGX_SAMPLING", "BRDF_SAMPLING",
                   "LIGHT_SOURCE_SAMPLING", "NEE", "RESTIR_DI", "RESTIR_GI"]

BUF_ACCUM, BUF_IMAGE, BUF_PAYLOAD, BUF_DEPTH, BUF_NORMAL, BUF_DI, BUF_DI_PREV, BUF_GI, BUF_GI_PREV = range(9)

# numpy views of the per-pixel records (Ray.h:13-22, ReSTIR_DI_Reservoir.cuh:9-16, ReSTIR_GI_Reservoir.cuh:9-27)
PAYLOAD_DTYPE = np.dtype([("hitDistance", "<f4"), ("worldPosition", "<f4", 3), ("worldNormal", "<f4", 3),
                          ("u", "<f4"), ("v", "<f4"), ("objectIndex", "<i4")])
DI_DTYPE = np.dtype([("indexEmissive", "<u4"), ("weightEmissive", "<f4"), ("emissivePDF", "<f4"),
                     ("weightSum", "<f4"), ("M", "<u4")])
GI_DTYPE = np.dtype([("visiblePoint", "<f4", 3), ("visibleNormal", "<f4", 2), ("samplePoint", "<f4", 3),
                     ("sampleNormal", "<f4", 2), ("Lo", "<f4", 3), ("randSeed", "<u4"), ("samplePDF", "<f4"),
                     ("weightSample", "<f4"), ("M", "<u4"), ("weightSum", "<f4")])
assert PAYLOAD_DTYPE.itemsize == 40 and DI_DTYPE.itemsize == 20 and GI_DTYPE.itemsize == 72
BUFFER_DTYPES = {BUF_ACCUM: np.dtype(("<f4", 4)), BUF_IMAGE: np.dtype("<u4"), BUF_PAYLOAD: PAYLOAD_DTYPE,
                 BUF_DEPTH: np.dtype("<f4"), BUF_NORMAL: np.dtype(("<f4", 2)), BUF_DI: DI_DTYPE, BUF_DI_PREV: DI_DTYPE,
                 BUF_GI: GI_DTYPE, BUF_GI_PREV: GI_DTYPE}

VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("uv", "<f4", 2)])
TRIANGLE_DTYPE = np.dtype([("v0", "<u4"), ("v1", "<u4"), ("v2", "<u4"), ("materialIndex", "<i4")])
MATERIAL_DTYPE = np.dtype([("isUseAlbedoMap", "<u4"), ("albedo", "<f4", 3), ("albedoMapIndex", "<u4"),
                           ("roughness", "<f4"), ("metallic", "<f4"), ("emissionColor", "<f4", 3),
                           ("emissionPower", "<f4")])
MESH_DTYPE = np.dtype([("firstTriangle", "<u4"), ("triangleCount", "<u4"), ("materialIndex", "<i4")])
LT_NODE_DTYPE = np.dtype([("energy", "<f4"), ("numEmitters", "<u4"), ("left", "<u4"), ("rightOrEmitter", "<u4"),
                          ("isLeaf", "<u4"), ("coneAxis", "<f4", 3), ("theta_o", "<f4"), ("theta_e", "<f4"),
                          ("boxLo", "<f4", 3), ("boxHi", "<f4", 3), ("boxCentroid", "<f4", 3), ("_pad", "<u4")])
assert VERTEX_DTYPE.itemsize == 32 and TRIANGLE_DTYPE.itemsize == 16 and MATERIAL_DTYPE.itemsize == 44
assert MESH_DTYPE.itemsize == 12 and LT_NODE_DTYPE.itemsize == 80
BVH_NODE_DTYPE = np.dtype([("origin", "<f4", 3), ("ex", "u1", 3), ("meta", "u1"), ("child", "<i4", 4),
                           ("qlo", "u1", (3, 4)), ("qhi", "u1", (3, 4)), ("pad", "<u4", 2)])
BVH_TRI_DTYPE = np.dtype([("v0", "<f4", 3), ("e1", "<f4", 3), ("e2", "<f4", 3), ("tri", "<u4"), ("pad", "<u4", 2)])
assert BVH_NODE_DTYPE.itemsize == 64 and BVH_TRI_DTYPE.itemsize == 48


class Settings(C.Structure):  # RenderingSettings.h:5-22 (52 B) with the reference's defaults
    _fields_ = [("to_accumulate", C.c_uint8), ("_pad0", C.c_uint8 * 3), ("light_bounces", C.c_int32),
                ("sample_count", C.c_int32), ("sky_color", C.c_float * 3), ("technique", C.c_int32),
                ("light_candidate_count", C.c_int32), ("rand_seed", C.c_uint32), ("use_temporal_reuse", C.c_uint8),
                ("use_spatial_reuse", C.c_uint8), ("_pad1", C.c_uint8 * 2), ("temporal_history_limit", C.c_int32),
                ("spatial_neighbor_num", C.c_int32), ("spatial_neighbor_radius", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        self.to_accumulate = 1
        self.light_bounces = 1
        self.sample_count = 1
        self.sky_color = (C.c_float * 3)(1.0, 1.0, 1.0)
        self.technique = BRUTE_FORCE
        self.light_candidate_count = 4
        self.rand_seed = 1
        self.use_temporal_reuse = 0
        self.use_spatial_reuse = 0
        self.temporal_history_limit = 2
        self.spatial_neighbor_num = 5
        self.spatial_neighbor_radius = 30
        for k, v in kw.items():
            if k == "sky_color":
                self.sky_color = (C.c_float * 3)(*v)
            else:
                if not hasattr(self, k):
                    raise AttributeError(k)
                setattr(self, k, v)


assert C.sizeof(Settings) == 52


class Texture(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class LightTrees(C.Structure):
    _fields_ = [("tlas_nodes", C.c_void_p), ("tlas_node_count", C.c_uint32), ("tlas_root", C.c_uint32),
                ("blas_nodes", C.c_void_p), ("blas_first", C.c_void_p), ("blas_count", C.c_void_p),
                ("blas_root", C.c_void_p)]


class SceneDesc(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("vertex_count", C.c_uint32),
                ("triangles", C.c_void_p), ("triangle_count", C.c_uint32), ("triangle_stride", C.c_uint32),
                ("materials", C.c_void_p), ("material_count", C.c_uint32),
                ("meshes", C.c_void_p), ("mesh_count", C.c_uint32),
                ("textures", C.POINTER(Texture)), ("texture_count", C.c_uint32),
                ("emissive_triangles", C.c_void_p), ("emissive_count", C.c_uint32),
                ("light_trees", C.POINTER(LightTrees))]


class CameraDesc(C.Structure):
    _fields_ = [("projection", C.c_float * 16), ("view", C.c_float * 16), ("prev_projection", C.c_float * 16),
                ("prev_view", C.c_float * 16), ("inverse_projection", C.c_float * 16), ("inverse_view", C.c_float * 16),
                ("position", C.c_float * 3), ("viewport_width", C.c_uint32), ("viewport_height", C.c_uint32)]


class FrameStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_float), ("kernel_ms_part", C.c_float * 4), ("rays", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("hits", C.c_uint64), ("part_rays", C.c_uint64 * 4),
                ("part_box_tests", C.c_uint64 * 4), ("part_tri_tests", C.c_uint64 * 4), ("part_hits", C.c_uint64 * 4),
                ("launches", C.c_uint32), ("node_visits", C.c_uint64), ("part_node_visits", C.c_uint64 * 4)]


EXPORTED_SYMBOLS = [
    "fyprt_create", "fyprt_destroy", "fyprt_last_error", "fyprt_resize", "fyprt_set_rows", "fyprt_upload_scene",
    "fyprt_set_camera", "fyprt_render", "fyprt_render_async", "fyprt_synchronize", "fyprt_readback",
    "fyprt_image_device_ptr", "fyprt_set_external_image", "fyprt_stream", "fyprt_read_buffer", "fyprt_frame_timings",
    "fyprt_reset_frame_index", "fyprt_frame_index", "fyprt_export_bvh", "fyprt_export_lighttrees", "fyprt_get_tuning", "fyprt_update_vertices",
    "fyprt_set_ray_counting", "fyprt_set_tuning", "fyprt_version",
    "fyprt_group_create", "fyprt_group_destroy", "fyprt_group_set_rows", "fyprt_group_set_halo_mode", "fyprt_group_render", "fyprt_group_gather",
    "fyprt_group_synchronize", "fyprt_comm_unique_id", "fyprt_comm_init_rank", "fyprt_comm_set_rows", "fyprt_comm_set_halo_mode", "fyprt_comm_render",
    "fyprt_comm_gather", "fyprt_comm_destroy", "fyprt_render_part", "fyprt_balance_rows", "fyprt_last_frame_ms", "fyprt_halo_plan", "fyprt_comm_ops",
    "fyprt_set_object_vertices", "fyprt_update_transforms", "fyprt_compare_image",
    "fyprt_set_row_stripes", "fyprt_group_set_interleave", "fyprt_comm_set_interleave", "fyprt_selftest_math",
]


class FyprtError(RuntimeError):
    pass


_lib = None


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """Load libfyprt.so (built by __graft_entry__.build() / csrc/build.sh). Fails loudly."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else Path(os.environ.get("FYPRT_LIB", LIB_PATH))     # FYPRT_LIB: an alternative build (A/B experiments)
    if not p.exists():
        raise FyprtError(f"HIP extension not built: {p} is missing. Run `python __graft_entry__.py build` "
                         f"(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = C.CDLL(str(p))
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
    lib.fyprt_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.fyprt_destroy.argtypes = [vp]
    lib.fyprt_destroy.restype = None
    lib.fyprt_last_error.argtypes = [vp]
    lib.fyprt_last_error.restype = C.c_char_p
    lib.fyprt_resize.argtypes = [vp, u32, u32]
    lib.fyprt_set_rows.argtypes = [vp, u32, u32, u32]
    lib.fyprt_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    lib.fyprt_set_camera.argtypes = [vp, C.POINTER(CameraDesc)]
    lib.fyprt_render.argtypes = [vp, C.POINTER(Settings), C.POINTER(FrameStats)]
    lib.fyprt_render_async.argtypes = [vp, C.POINTER(Settings)]
    lib.fyprt_synchronize.argtypes = [vp]
    lib.fyprt_frame_timings.argtypes = [vp, u32, C.POINTER(C.c_float * 4), C.POINTER(u32)]
    lib.fyprt_readback.argtypes = [vp, vp, vp]
    lib.fyprt_image_device_ptr.argtypes = [vp, C.POINTER(vp)]
    lib.fyprt_set_external_image.argtypes = [vp, vp]
    lib.fyprt_stream.argtypes = [vp, C.POINTER(vp)]
    lib.fyprt_read_buffer.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.fyprt_reset_frame_index.argtypes = [vp]
    lib.fyprt_frame_index.argtypes = [vp]
    lib.fyprt_frame_index.restype = u32
    lib.fyprt_update_vertices.argtypes = [vp, vp, u32]
    lib.fyprt_get_tuning.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    lib.fyprt_export_bvh.argtypes = [vp, vp, C.POINTER(u32), vp, C.POINTER(u32), C.POINTER(i32), C.POINTER(u32)]
    lib.fyprt_export_lighttrees.argtypes = [vp, vp, C.POINTER(u32), C.POINTER(u32), vp, C.POINTER(u32), vp, vp, vp]
    lib.fyprt_set_ray_counting.argtypes = [vp, C.c_int]
    lib.fyprt_set_tuning.argtypes = [vp, C.c_int, C.c_int]
    lib.fyprt_version.restype = C.c_char_p
    lib.fyprt_set_object_vertices.argtypes = [vp, vp, u32, C.POINTER(u32)]
    lib.fyprt_update_transforms.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_float), u32]
    lib.fyprt_compare_image.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.fyprt_group_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(u32), C.POINTER(vp)]
    lib.fyprt_group_destroy.argtypes = [vp]
    lib.fyprt_group_destroy.restype = None
    lib.fyprt_group_set_rows.argtypes = [vp, C.POINTER(u32)]
    lib.fyprt_group_set_halo_mode.argtypes = [vp, C.c_int]
    lib.fyprt_group_set_interleave.argtypes = [vp, u32]
    lib.fyprt_comm_set_interleave.argtypes = [vp, u32]
    lib.fyprt_set_row_stripes.argtypes = [vp, u32, u32, u32]
    lib.fyprt_selftest_math.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(u32)]
    lib.fyprt_group_render.argtypes = [vp, C.POINTER(Settings)]
    lib.fyprt_group_gather.argtypes = [vp, C.c_int]
    lib.fyprt_group_synchronize.argtypes = [vp]
    lib.fyprt_comm_unique_id.argtypes = [vp]
    lib.fyprt_comm_init_rank.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(u32)]
    lib.fyprt_comm_set_rows.argtypes = [vp, C.POINTER(u32)]
    lib.fyprt_comm_set_halo_mode.argtypes = [vp, C.c_int]
    lib.fyprt_comm_render.argtypes = [vp, C.POINTER(Settings)]
    lib.fyprt_comm_gather.argtypes = [vp, C.c_int]
    lib.fyprt_comm_destroy.argtypes = [vp]
    lib.fyprt_comm_destroy.restype = None
    lib.fyprt_render_part.argtypes = [vp, C.POINTER(Settings), C.c_int]
    lib.fyprt_balance_rows.argtypes = [C.POINTER(u32), C.POINTER(C.c_float), C.c_int, u32, u32, C.POINTER(u32)]
    lib.fyprt_last_frame_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.fyprt_halo_plan.argtypes = [C.POINTER(u32), C.c_int, u32, u32, C.c_int, C.POINTER(u32), C.c_int]
    if hasattr(lib, "fyprt_comm_ops"):       # (absent only in older builds loaded through FYPRT_LIB for an A/B run)
        lib.fyprt_comm_ops.argtypes = [C.c_int, C.POINTER(u32), C.POINTER(u32), C.c_int, u32, u32, C.c_int, u32, C.c_int, C.POINTER(u32), C.c_int, C.POINTER(C.c_uint64), C.c_int]
    alternative = path is not None or "FYPRT_LIB" in os.environ      # an older build loaded for an A/B run may lack the newest entry points
    for f in EXPORTED_SYMBOLS:
        if alternative and not hasattr(lib, f):
            continue
        getattr(lib, f)                                                # the product library must export every symbol of include/fyprt.h
    if path is None:
        _lib = lib
    return lib


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_scene_desc(scene, light_trees: dict | None = None):
    """Build a fyprt_scene_desc from a host `Scene` (fypraytracer_amd.scene.Scene).
    Returns (desc, keepalive) — keep `keepalive` referenced while the desc is in use."""
    keep = []
    v = np.ascontiguousarray(scene.world_vertices, dtype=VERTEX_DTYPE)
    t = np.ascontiguousarray(scene.triangles, dtype=TRIANGLE_DTYPE)
    m = np.ascontiguousarray(scene.materials_array(), dtype=MATERIAL_DTYPE)
    ms = np.ascontiguousarray(scene.meshes_array(), dtype=MESH_DTYPE)
    keep += [v, t, m, ms]
    d = SceneDesc()
    d.vertices, d.vertex_count = _ptr(v), len(v)
    d.triangles, d.triangle_count, d.triangle_stride = _ptr(t), len(t), TRIANGLE_DTYPE.itemsize
    d.materials, d.material_count = _ptr(m), len(m)
    d.meshes, d.mesh_count = _ptr(ms), len(ms)
    texs = (Texture * max(1, len(scene.textures)))()
    for i, px in enumerate(scene.textures):
        px = np.ascontiguousarray(px, dtype=np.uint32)
        keep.append(px)
        texs[i].pixels, texs[i].height, texs[i].width = _ptr(px), px.shape[0], px.shape[1]
    keep.append(texs)
    d.textures = C.cast(texs, C.POINTER(Texture))
    d.texture_count = len(scene.textures)
    d.emissive_triangles, d.emissive_count = None, 0
    if light_trees is not None:
        lt = LightTrees()
        arrs = {k: np.ascontiguousarray(light_trees[k]) for k in ("tlas", "blas", "blas_first", "blas_count", "blas_root")}
        keep.append(arrs)
        lt.tlas_nodes, lt.tlas_node_count, lt.tlas_root = _ptr(arrs["tlas"]), len(arrs["tlas"]), int(light_trees["tlas_root"])
        lt.blas_nodes, lt.blas_first = _ptr(arrs["blas"]), _ptr(arrs["blas_first"])
        lt.blas_count, lt.blas_root = _ptr(arrs["blas_count"]), _ptr(arrs["blas_root"])
        keep.append(lt)
        d.light_trees = C.pointer(lt)
    return d, keep


def make_camera_desc(cam) -> CameraDesc:
    """From fypraytracer_amd.scene.Camera (column-major float32 4x4 matrices)."""
    c = CameraDesc()
    for name, mat in (("projection", cam.projection), ("view", cam.view), ("prev_projection", cam.prev_projection),
                      ("prev_view", cam.prev_view), ("inverse_projection", cam.inverse_projection),
                      ("inverse_view", cam.inverse_view)):
        flat = np.asarray(mat, dtype=np.float32).reshape(16)   # stored column-major already (see scene.Camera)
        setattr(c, name, (C.c_float * 16)(*flat.tolist()))
    c.position = (C.c_float * 3)(*[float(x) for x in cam.position])
    c.viewport_width, c.viewport_height = int(cam.width), int(cam.height)
    return c


class Context:
    """One renderer context on one GPU — the Python face of the reference's `Renderer`
    (Renderer.h:41-56): resize / upload_scene / set_camera / render / readback."""

    def __init__(self, device: int = 0, lib: C.CDLL | None = None):
        self.lib = lib or load_library()
        h = C.c_void_p()
        rc = self.lib.fyprt_create(device, C.byref(h))
        if rc != 0:
            raise FyprtError(f"fyprt_create({device}) failed: {self.lib.fyprt_last_error(None).decode()}")
        self.h = h
        self.width = self.height = 0
        self._keep = None

    def _check(self, rc):
        if rc != 0:
            raise FyprtError(f"fyprt error {rc}: {self.lib.fyprt_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.fyprt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def resize(self, width, height):
        self._check(self.lib.fyprt_resize(self.h, width, height))
        self.width, self.height = width, height

    def set_rows(self, row_begin, row_end, halo_rows=0):
        self._check(self.lib.fyprt_set_rows(self.h, row_begin, row_end, halo_rows))

    def selftest_math(self):
        """(mismatch counts, first offending argument bits) of the lean sqrt / 1/x / 1/sqrt(x) over all 2^32 arguments each."""
        n, f = (C.c_uint64 * 3)(), (C.c_uint32 * 3)()
        self._check(self.lib.fyprt_selftest_math(self.h, n, f))
        return list(n), list(f)

    def set_row_stripes(self, stripe_rows, parts=1, part=0):
        """Interleaved split of the per-pixel techniques: this context renders stripes part, part + parts, ... (0 rows = off)."""
        self._check(self.lib.fyprt_set_row_stripes(self.h, stripe_rows, parts, part))

    def upload_scene(self, scene, light_trees=None):
        d, keep = make_scene_desc(scene, light_trees)
        self._check(self.lib.fyprt_upload_scene(self.h, C.byref(d)))

    def set_camera(self, cam):
        c = make_camera_desc(cam)
        self._check(self.lib.fyprt_set_camera(self.h, C.byref(c)))

    def render(self, settings: Settings) -> FrameStats:
        st = FrameStats()
        self._check(self.lib.fyprt_render(self.h, C.byref(settings), C.byref(st)))
        return st

    def render_async(self, settings: Settings):
        self._check(self.lib.fyprt_render_async(self.h, C.byref(settings)))

    def render_part(self, settings: Settings, part: int):
        """One of the two parts of a ReSTIR frame (asynchronous): what a host with its own transport for the halo rows calls."""
        self._check(self.lib.fyprt_render_part(self.h, C.byref(settings), part))

    def synchronize(self):
        self._check(self.lib.fyprt_synchronize(self.h))

    def frame_timings(self, frames_back=0):
        """(per-launch ms [4], launches) of the frame enqueued `frames_back` frames ago; synchronize() first."""
        ms, n = (C.c_float * 4)(), C.c_uint32()
        self._check(self.lib.fyprt_frame_timings(self.h, frames_back, C.byref(ms), C.byref(n)))
        return list(ms), n.value

    def readback(self, want_accum=True):
        n = self.width * self.height
        img = np.empty(n, dtype=np.uint32)
        acc = np.empty((n, 4), dtype=np.float32) if want_accum else None
        self._check(self.lib.fyprt_readback(self.h, _ptr(img), _ptr(acc)))
        return img.reshape(self.height, self.width), (acc.reshape(self.height, self.width, 4) if want_accum else None)

    def read_buffer(self, which) -> np.ndarray:
        dt = BUFFER_DTYPES[which]
        out = np.empty(self.width * self.height, dtype=dt)
        self._check(self.lib.fyprt_read_buffer(self.h, which, _ptr(out), out.nbytes))
        return out

    def image_device_ptr(self) -> int:
        p = C.c_void_p()
        self._check(self.lib.fyprt_image_device_ptr(self.h, C.byref(p)))
        return p.value

    def set_external_image(self, device_ptr: int):
        self._check(self.lib.fyprt_set_external_image(self.h, C.c_void_p(device_ptr)))

    def stream(self) -> int:
        p = C.c_void_p()
        self._check(self.lib.fyprt_stream(self.h, C.byref(p)))
        return p.value or 0

    def reset_frame_index(self):
        self._check(self.lib.fyprt_reset_frame_index(self.h))

    @property
    def frame_index(self) -> int:
        return int(self.lib.fyprt_frame_index(self.h))

    def set_ray_counting(self, on: bool):
        self._check(self.lib.fyprt_set_ray_counting(self.h, 1 if on else 0))

    def set_tuning(self, key: int, value: int):
        self._check(self.lib.fyprt_set_tuning(self.h, key, value))

    def update_vertices(self, scene):
        """Moved geometry, same topology: refit on the device (fyprt_update_vertices)."""
        v = np.ascontiguousarray(scene.world_vertices)
        self._check(self.lib.fyprt_update_vertices(self.h, v.ctypes.data, len(v)))

    def set_object_vertices(self, scene):
        """Object-space vertices + per-mesh vertex ranges (scene.mesh_transforms), for update_transforms."""
        v = np.ascontiguousarray(scene.vertices)
        first = [tr["vertex_start"] for tr in scene.mesh_transforms] + [len(v)]
        self._check(self.lib.fyprt_set_object_vertices(self.h, v.ctypes.data, len(v), (C.c_uint32 * len(first))(*first)))

    def update_transforms(self, scene, mesh_indices):
        """A transform edit applied on the device: 64 bytes per mesh (Scene.mesh_matrix) instead of its vertices."""
        mats = np.ascontiguousarray(np.stack([scene.mesh_matrix(m) for m in mesh_indices]).astype(np.float32).reshape(-1))
        idx = (C.c_uint32 * len(mesh_indices))(*[int(m) for m in mesh_indices])
        self._check(self.lib.fyprt_update_transforms(self.h, idx, mats.ctypes.data_as(C.POINTER(C.c_float)), len(mesh_indices)))

    def compare_image(self, reference, flip_reference_rows=False):
        """(MSE, PSNR) of the current frame against `reference` (uint32 ABGR8, H x W), reduced on the device (MisUtils::ComputeMSE)."""
        ref = np.ascontiguousarray(reference, dtype=np.uint32)
        mse, psnr = C.c_double(), C.c_double()
        self._check(self.lib.fyprt_compare_image(self.h, ref.ctypes.data, 1 if flip_reference_rows else 0, C.byref(mse), C.byref(psnr)))
        return mse.value, psnr.value

    def get_tuning(self, key: int) -> int:
        v = C.c_int()
        self._check(self.lib.fyprt_get_tuning(self.h, key, C.byref(v)))
        return v.value

    def export_bvh(self):
        nn, nt, root, depth = C.c_uint32(), C.c_uint32(), C.c_int32(), C.c_uint32()
        self._check(self.lib.fyprt_export_bvh(self.h, None, C.byref(nn), None, C.byref(nt), C.byref(root), C.byref(depth)))
        nodes = np.empty(nn.value, dtype=BVH_NODE_DTYPE)
        tris = np.empty(nt.value, dtype=BVH_TRI_DTYPE)
        self._check(self.lib.fyprt_export_bvh(self.h, _ptr(nodes), C.byref(nn), _ptr(tris), C.byref(nt), C.byref(root), C.byref(depth)))
        return {"nodes": nodes, "tris": tris, "root": root.value, "max_stack": depth.value, "stack_budget": self.get_tuning(8), "skip_dead_rays": self.get_tuning(18)}

    def export_lighttrees(self, mesh_count: int):
        tc, tr, bt = C.c_uint32(), C.c_uint32(), C.c_uint32()
        first = np.zeros(mesh_count, dtype=np.uint32)
        count = np.zeros(mesh_count, dtype=np.uint32)
        root = np.zeros(mesh_count, dtype=np.uint32)
        self._check(self.lib.fyprt_export_lighttrees(self.h, None, C.byref(tc), C.byref(tr), None, C.byref(bt), None, None, None))
        tlas = np.zeros(tc.value, dtype=LT_NODE_DTYPE)
        blas = np.zeros(bt.value, dtype=LT_NODE_DTYPE)
        self._check(self.lib.fyprt_export_lighttrees(self.h, _ptr(tlas), C.byref(tc), C.byref(tr), _ptr(blas), C.byref(bt),
                                                     _ptr(first), _ptr(count), _ptr(root)))
        return {"tlas": tlas, "tlas_root": tr.value, "blas": blas, "blas_first": first, "blas_count": count, "blas_root": root}


def _u32_array(values):
    return (C.c_uint32 * len(values))(*[int(v) for v in values])


class Group:
    """Several contexts (one per GPU) of ONE process rendering one frame in row bands: fyprt_group_* (peer copies, no RCCL)."""

    def __init__(self, contexts, row_bounds, halo_mode=0):
        self.contexts, self.lib = list(contexts), contexts[0].lib
        arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
        h = C.c_void_p()
        rc = self.lib.fyprt_group_create(arr, len(contexts), _u32_array(row_bounds), C.byref(h))
        if rc != 0:
            raise FyprtError(f"fyprt_group_create failed ({rc}): {self.lib.fyprt_last_error(contexts[0].h).decode()}")
        self.h, self.row_bounds = h, list(row_bounds)
        self.set_halo_mode(halo_mode)

    def _check(self, rc):
        if rc != 0:
            raise FyprtError(f"fyprt error {rc}: " + " | ".join(self.lib.fyprt_last_error(c.h).decode() for c in self.contexts))

    def set_halo_mode(self, mode):
        self._check(self.lib.fyprt_group_set_halo_mode(self.h, mode))

    def set_interleave(self, stripe_rows):
        self._check(self.lib.fyprt_group_set_interleave(self.h, stripe_rows))

    def set_rows(self, row_bounds):
        self._check(self.lib.fyprt_group_set_rows(self.h, _u32_array(row_bounds)))
        self.row_bounds = list(row_bounds)

    def render(self, settings):
        self._check(self.lib.fyprt_group_render(self.h, C.byref(settings)))

    def gather(self, root=0):
        self._check(self.lib.fyprt_group_gather(self.h, root))

    def synchronize(self):
        self._check(self.lib.fyprt_group_synchronize(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.fyprt_group_destroy(self.h)
            self.h = None


def balance_rows(row_bounds, band_ms, min_rows=16, max_shift=1 << 30, lib=None):
    """fyprt_balance_rows: cost-balanced band boundaries from the per-band frame times."""
    lib = lib or load_library()
    n = len(band_ms)
    out = (C.c_uint32 * (n + 1))()
    rc = lib.fyprt_balance_rows(_u32_array(row_bounds), (C.c_float * n)(*[float(x) for x in band_ms]), n, min_rows, max_shift, out)
    if rc != 0:
        raise FyprtError(f"fyprt_balance_rows failed ({rc})")
    return list(out)


def comm_ops(kind, row_bounds, rank, width, bytes_per_pixel, halo=0, height=0, wrap_row=True, new_bounds=None, lib=None):
    """fyprt_comm_ops: [(is_recv, peer, buffer, offset, bytes), ...] that `rank` issues in one RCCL group section (kind 0: halo exchange,
    kind 1: fyprt_comm_set_rows from row_bounds to new_bounds)."""
    lib = lib or load_library()
    n = len(row_bounds) - 1
    nb = _u32_array(new_bounds) if new_bounds is not None else None
    args = (kind, _u32_array(row_bounds), nb, n, halo, height, 1 if wrap_row else 0, width, rank, _u32_array(bytes_per_pixel), len(bytes_per_pixel))
    cnt = lib.fyprt_comm_ops(*args, None, 0)
    if cnt < 0:
        raise FyprtError("fyprt_comm_ops: bad arguments")
    out = (C.c_uint64 * (5 * max(cnt, 1)))()
    lib.fyprt_comm_ops(*args, out, cnt)
    return [tuple(int(v) for v in out[5 * k: 5 * k + 5]) for k in range(cnt)]


def halo_plan(row_bounds, halo, height, wrap_row=True, lib=None):
    """fyprt_halo_plan: [(receiver, owner, first row, end row), ...] of one halo exchange."""
    lib = lib or load_library()
    n = len(row_bounds) - 1
    cnt = lib.fyprt_halo_plan(_u32_array(row_bounds), n, halo, height, 1 if wrap_row else 0, None, 0)
    out = (C.c_uint32 * (4 * max(cnt, 1)))()
    lib.fyprt_halo_plan(_u32_array(row_bounds), n, halo, height, 1 if wrap_row else 0, out, cnt)
    return [tuple(out[4 * k: 4 * k + 4]) for k in range(cnt)]
