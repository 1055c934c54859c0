"""Host-side mirror of the reference's scene API for the hot path's *inputs*.

Mirrors (names and argument meaning) the parts of the reference that produce what the
kernels read: `Material` (Material.cuh:7-21), `Scene.add_new_mesh_to_scene`
(Scene::AddNewMeshToScene, Scene.cpp:9-92: world vertices, Triangle list),
`Scene.init_scene_emissive_triangles` (Scene.cpp:209-221) and `Camera`
(Camera.cpp:96-153: projection / view / inverses; the per-pixel ray directions are
generated on the GPU from the inverse matrices).  Matrices are numpy float32 arrays
indexed [column][row] (glm::mat4 memory order), so `.reshape(16)` is what crosses the
C ABI.

ONE implementation of the arithmetic: the transform of a mesh, the vertex loop, the emissive test and the
whole Camera are host/HostTypes.h (what the C++ facade uses), called here through libfyprt_host.so — this
module holds containers and bookkeeping only (the round-2 numpy copy agreed with the C++
one on the transforms but not, to the last bit, on the camera matrices).  glm itself is an un-vendored
dependency of the reference; HostTypes.h states glm's documented right-handed,
[-1,1]-depth definitions.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from .capi import MATERIAL_DTYPE, MESH_DTYPE, TRIANGLE_DTYPE, VERTEX_DTYPE

F = np.float32


@dataclass
class Material:  # Material.cuh:7-16 (defaults included)
    albedo: tuple = (1.0, 0.0, 1.0)
    roughness: float = 1.0
    metallic: float = 0.0
    emission_color: tuple = (0.0, 0.0, 0.0)
    emission_power: float = 0.0
    is_use_albedo_map: bool = False
    albedo_map_index: int = 0xFFFFFFFF

    def get_emission(self):  # Material.cu:5-8
        return np.asarray(self.emission_color, dtype=F) * F(self.emission_power)


_HOST_PATH = Path(__file__).resolve().parent / "host" / "libfyprt_host.so"
_host = None


def _host_lib():
    """libfyprt_host.so (host/build.sh): fails loudly when it is not built."""
    global _host
    if _host is None:
        if not _HOST_PATH.exists():
            raise RuntimeError(f"{_HOST_PATH} is missing: run `python __graft_entry__.py build` (host/build.sh)")
        L = C.CDLL(str(_HOST_PATH))
        vp, fp = C.c_void_p, C.c_void_p
        L.fyprt_host_mesh_matrix.argtypes = [fp, fp, fp, fp]
        L.fyprt_host_to_world.argtypes = [fp, fp, fp, C.c_uint32]
        L.fyprt_host_is_emissive.argtypes = [fp, C.c_float]
        L.fyprt_host_camera_create.restype = vp
        L.fyprt_host_camera_create.argtypes = [C.c_float, C.c_float, C.c_float]
        L.fyprt_host_camera_destroy.argtypes = [vp]
        L.fyprt_host_camera_on_resize.argtypes = [vp, C.c_uint32, C.c_uint32]
        for f in ("set_position", "set_direction", "assign_forward", "assign_position", "state"):
            getattr(L, "fyprt_host_camera_" + f).argtypes = [vp, fp]
        L.fyprt_host_camera_on_update.argtypes = [vp, C.c_float, C.c_char_p, C.c_float, C.c_float]
        L.fyprt_host_camera_commit_frame.argtypes = [vp]
        _host = L
    return _host


def _f3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=F).reshape(3))


def mesh_matrix(pos, rotation, scale_):
    """Mesh::UpdateWorldTransform: T * yawPitchRoll(radians(ry), radians(rx), radians(rz)) * S as a [col][row] float32 matrix."""
    out = np.zeros(16, dtype=F)
    p, r, sc = _f3(pos), _f3(rotation), _f3(scale_)
    _host_lib().fyprt_host_mesh_matrix(p.ctypes.data, r.ctypes.data, sc.ctypes.data, out.ctypes.data)
    return out.reshape(4, 4)


def to_world(matrix, vertices):
    """The vertex loop of Scene::AddNewMeshToScene / SceneManager (Scene.cpp:42-51, SceneManager.cpp:30-41) on VERTEX_DTYPE records."""
    m = np.ascontiguousarray(np.asarray(matrix, dtype=F).reshape(16))
    vin = np.ascontiguousarray(vertices)
    out = np.empty_like(vin)
    if len(vin):
        _host_lib().fyprt_host_to_world(m.ctypes.data, vin.ctypes.data, out.ctypes.data, len(vin))
    return out


class Camera:
    """Camera.h:9-83 / Camera.cpp — a handle to host/HostTypes.h's Camera.  `on_resize`, `set_position`, `set_direction` keep the
    reference's semantics, including the reset of the previous-frame matrices on an explicit pose change (Camera.cpp:108-116) and
    `commit_frame()` = WalnutApp.cpp:908-909.  The matrices are read back after every call as [col][row] float32 arrays."""

    _NAMES = ("projection", "view", "prev_projection", "prev_view", "inverse_projection", "inverse_view")

    def __init__(self, vertical_fov=45.0, near_clip=0.1, far_clip=100.0):
        self.vertical_fov, self.near_clip, self.far_clip = vertical_fov, near_clip, far_clip
        self._lib = _host_lib()
        self._h = C.c_void_p(self._lib.fyprt_host_camera_create(vertical_fov, near_clip, far_clip))
        self.width = self.height = 0
        self._pull()

    def __del__(self):
        try:
            if self._h:
                self._lib.fyprt_host_camera_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _pull(self):
        st = np.zeros(102, dtype=F)
        self._lib.fyprt_host_camera_state(self._h, st.ctypes.data)
        for k, n in enumerate(self._NAMES):
            setattr(self, n, st[16 * k:16 * k + 16].reshape(4, 4).copy())
        self._position, self._forward = st[96:99].copy(), st[99:102].copy()

    # Camera::GetPosition / GetDirection return references the caller may assign through: plain member writes, no view update
    @property
    def position(self):
        return self._position

    @position.setter
    def position(self, p):
        p = _f3(p)
        self._lib.fyprt_host_camera_assign_position(self._h, p.ctypes.data)
        self._position = p.copy()

    @property
    def forward(self):
        return self._forward

    @forward.setter
    def forward(self, d):
        d = _f3(d)
        self._lib.fyprt_host_camera_assign_forward(self._h, d.ctypes.data)
        self._forward = d.copy()

    def on_resize(self, width, height):  # Camera.cpp:96-106
        self._lib.fyprt_host_camera_on_resize(self._h, width, height)
        self.width, self.height = width, height
        self._pull()

    def set_position(self, p):
        p = _f3(p)
        self._lib.fyprt_host_camera_set_position(self._h, p.ctypes.data)
        self._pull()

    def set_direction(self, d):
        d = _f3(d)
        self._lib.fyprt_host_camera_set_direction(self._h, d.ctypes.data)
        self._pull()

    def commit_frame(self):
        """MainLayer::Render tail (WalnutApp.cpp:908-909): prev := current."""
        self._lib.fyprt_host_camera_commit_frame(self._h)
        self._pull()

    def on_update(self, ts, keys="", mouse_delta=(0.0, 0.0)):
        """Camera::OnUpdate (Camera.cpp:18-94) with the right mouse button held: `keys` is the set of pressed keys out of
        "WSADQE", `mouse_delta` the cursor movement in pixels since the last call.  Moves / rotates the camera and
        recalculates the view — the previous-frame matrices are NOT touched (only commit_frame / an explicit pose reset
        do that), which is what makes ReSTIR's temporal reprojection land on a different pixel.  Returns `moved`."""
        moved = bool(self._lib.fyprt_host_camera_on_update(self._h, float(ts), keys.upper().encode(), float(mouse_delta[0]), float(mouse_delta[1])))
        self._pull()
        return moved

    def ray_directions(self):
        """Camera::RecalculateRayDirections (Camera.cpp:136-153), numpy float32 — used only by
        host-side tools; the GPU regenerates these from the inverse matrices."""
        W, H = self.width, self.height
        xs = (np.arange(W, dtype=F) / F(W)) * F(2) - F(1)
        ys = (np.arange(H, dtype=F) / F(H)) * F(2) - F(1)
        cx, cy = np.meshgrid(xs, ys)
        ip, iv = self.inverse_projection, self.inverse_view
        tgt = (ip[0][None, None, :] * cx[..., None] + ip[1][None, None, :] * cy[..., None]) + (ip[2][None, None, :] + ip[3][None, None, :])
        d = tgt[..., :3] / tgt[..., 3:4]
        d = d * (F(1) / np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]))[..., None]
        out = (iv[0][None, None, :3] * d[..., 0:1] + iv[1][None, None, :3] * d[..., 1:2]) + iv[2][None, None, :3] * d[..., 2:3]
        return out.astype(F)


@dataclass
class Scene:
    """Scene.h:23-56 reduced to the containers the kernels read."""
    materials: list = field(default_factory=list)
    textures: list = field(default_factory=list)          # each: uint32 [H][W] ABGR8
    world_vertices: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=VERTEX_DTYPE))
    triangles: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=TRIANGLE_DTYPE))
    meshes: list = field(default_factory=list)            # (firstTriangle, triangleCount, materialIndex)
    emissive_triangles: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.uint32))
    vertices: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=VERTEX_DTYPE))   # Scene::vertices: object space
    mesh_transforms: list = field(default_factory=list)   # per mesh: dict(pos, rotation, scale, vertex_start, vertex_count) (Mesh.h transform members)
    scene_manager: "SceneManager" = None                  # Scene::sceneManager (Scene.h), created on first use

    @staticmethod
    def _to_world(positions, normals, pos, rotation, scale_):
        """Mesh::UpdateWorldTransform + the vertex loop of Scene::AddNewMeshToScene / SceneManager (Scene.cpp:42-51,
        SceneManager.cpp:30-41): transform = T * yawPitchRoll(ry, rx, rz) * S, position / w, normal with w = 0, normalised."""
        v = np.zeros(len(positions), dtype=VERTEX_DTYPE)
        v["position"], v["normal"] = positions, normals
        w = to_world(mesh_matrix(pos, rotation, scale_), v)
        return w["position"].copy(), w["normal"].copy()

    def mesh_matrix(self, mesh_index):
        """Mesh::worldTransformMatrix of a mesh ([col][row] float32): T * yawPitchRoll(ry, rx, rz) * S (Mesh::UpdateWorldTransform)."""
        tr = self.mesh_transforms[mesh_index]
        return mesh_matrix(tr["pos"], tr["rotation"], tr["scale"])

    def add_new_mesh_to_scene(self, positions, normals, uvs, indices, pos=(0, 0, 0), rotation=(0, 0, 0),
                              scale_=(1, 1, 1), material_index=0):
        """Scene::AddNewMeshToScene (Scene.cpp:9-92): transform = T * yawPitchRoll(ry, rx, rz) * S,
        normals by the same matrix with w = 0 then normalised (not inverse-transpose)."""
        positions = np.asarray(positions, dtype=F).reshape(-1, 3)
        normals = np.asarray(normals, dtype=F).reshape(-1, 3)
        uvs = np.asarray(uvs, dtype=F).reshape(-1, 2)
        indices = np.asarray(indices, dtype=np.uint32).reshape(-1, 3)
        wp, wn = self._to_world(positions, normals, pos, rotation, scale_)
        v = np.zeros(len(positions), dtype=VERTEX_DTYPE)
        v["position"], v["normal"], v["uv"] = wp, wn, uvs
        vstart = len(self.world_vertices)
        t = np.zeros(len(indices), dtype=TRIANGLE_DTYPE)
        t["v0"], t["v1"], t["v2"] = (indices[:, 0] + vstart, indices[:, 1] + vstart, indices[:, 2] + vstart)
        t["materialIndex"] = material_index
        first = len(self.triangles)
        lv = np.zeros(len(positions), dtype=VERTEX_DTYPE)
        lv["position"], lv["normal"], lv["uv"] = positions, normals, uvs
        self.vertices = np.concatenate([self.vertices, lv])
        self.world_vertices = np.concatenate([self.world_vertices, v])
        self.triangles = np.concatenate([self.triangles, t])
        self.meshes.append((first, len(indices), material_index))
        self.mesh_transforms.append({"pos": tuple(pos), "rotation": tuple(rotation), "scale": tuple(scale_), "vertex_start": vstart, "vertex_count": len(positions)})
        return len(self.meshes) - 1

    def manager(self):
        if self.scene_manager is None:
            self.scene_manager = SceneManager()
        return self.scene_manager

    def init_scene_emissive_triangles(self):  # Scene.cpp:209-221
        lib = _host_lib()
        em = np.array([bool(lib.fyprt_host_is_emissive(_f3(m.emission_color).ctypes.data, float(m.emission_power))) for m in self.materials], dtype=bool)
        self.emissive_triangles = np.nonzero(em[self.triangles["materialIndex"]])[0].astype(np.uint32)
        return self.emissive_triangles

    def materials_array(self):
        a = np.zeros(len(self.materials), dtype=MATERIAL_DTYPE)
        for i, m in enumerate(self.materials):
            a[i] = (1 if m.is_use_albedo_map else 0, m.albedo, m.albedo_map_index & 0xFFFFFFFF, m.roughness, m.metallic,
                    m.emission_color, m.emission_power)
        return a

    def meshes_array(self):
        a = np.zeros(len(self.meshes), dtype=MESH_DTYPE)
        for i, (f, c, mi) in enumerate(self.meshes):
            a[i] = (f, c, mi)
        return a


class SceneManager:
    """SceneManager (Classes/Managers/SceneManager.{h,cpp}): the UI queues mesh / material edits, and
    PerformAllSceneUpdates applies them before the next frame and raises the renderer's scene-dirty flag
    (WalnutApp.cpp:662, :715, :776).  Kept as the reference does it:
      * a moved mesh's world vertices are recomputed from the object-space copy with the mesh's new transform
        (SceneManager.cpp:24-41; triangle boxes :46-60 are this build's builder's business);
      * a mesh material change rewrites the material index of its triangles (:69-80);
      * the emissive-triangle list is NOT refreshed here (the reference only builds it at load, Scene.cpp:284,
        WalnutApp.cpp:510) — a material that starts or stops emitting changes NEE's light trees (rebuilt from the materials at
        upload) but not ReSTIR's candidate list until init_scene_emissive_triangles() is called again;
      * both queues start with 20 default entries (SceneManager.h:25-26: `{20}`), so the very first call always raises
        the dirty flag even if nothing was edited.
    `renderer` is anything with set_scene_to_be_updated_flag(bool) (the facade) or None."""

    def __init__(self):
        self.meshes_to_update = [(False, False, 0xFFFFFFFF)] * 20       # (transform changed, material changed, mesh index)
        self.materials_to_update = [0] * 20

    def perform_all_scene_updates(self, scene, renderer=None):
        dirty = False
        if self.materials_to_update:
            dirty = True
        for transform_changed, material_changed, mi in self.meshes_to_update:
            if not (transform_changed or material_changed):
                continue
            first, count, material_index = scene.meshes[mi]
            tr = scene.mesh_transforms[mi]
            if transform_changed:
                a, n = tr["vertex_start"], tr["vertex_count"]
                wp, wn = Scene._to_world(scene.vertices["position"][a:a + n], scene.vertices["normal"][a:a + n], tr["pos"], tr["rotation"], tr["scale"])
                scene.world_vertices["position"][a:a + n] = wp
                scene.world_vertices["normal"][a:a + n] = wn
                dirty = True
            if material_changed:
                scene.triangles["materialIndex"][first:first + count] = material_index
                dirty = True
        self.meshes_to_update = []
        self.materials_to_update = []
        if dirty and renderer is not None:
            renderer.set_scene_to_be_updated_flag(True)
        return dirty

    # the two UI edits that feed the queues (WalnutApp.cpp:620-662, :690-715)
    def set_mesh_transform(self, scene, mesh_index, pos=None, rotation=None, scale_=None):
        tr = scene.mesh_transforms[mesh_index]
        if pos is not None:
            tr["pos"] = tuple(pos)
        if rotation is not None:
            tr["rotation"] = tuple(rotation)
        if scale_ is not None:
            tr["scale"] = tuple(scale_)
        self.meshes_to_update.append((True, False, mesh_index))

    def set_mesh_material(self, scene, mesh_index, material_index):
        first, count, _ = scene.meshes[mesh_index]
        scene.meshes[mesh_index] = (first, count, material_index)
        self.meshes_to_update.append((False, True, mesh_index))

    def material_edited(self, material_index):
        self.materials_to_update.append(material_index)
