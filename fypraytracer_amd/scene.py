"""Host-side mirror of the reference's scene API for the hot path's *inputs*.

Mirrors (names and argument meaning) the parts of the reference that produce what the
kernels read: `Material` (Material.cuh:7-21), `Scene.add_new_mesh_to_scene`
(Scene::AddNewMeshToScene, Scene.cpp:9-92: world vertices, Triangle list),
`Scene.init_scene_emissive_triangles` (Scene.cpp:209-221) and `Camera`
(Camera.cpp:96-153: projection / view / inverses; the per-pixel ray directions are
generated on the GPU from the inverse matrices).  Matrices are numpy float32 arrays
indexed [column][row] (glm::mat4 memory order), so `.reshape(16)` is what crosses the
C ABI.  glm itself is an un-vendored dependency of the reference; the
formulas below are glm's documented right-handed, [-1,1]-depth definitions.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

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


def _mat_identity():
    return np.eye(4, dtype=F)


def _matmul(a, b):
    """Product of two [col][row] matrices: (a*b)[col] = sum_k a[k] * b[col][k]."""
    out = np.zeros((4, 4), dtype=F)
    for j in range(4):
        acc = np.zeros(4, dtype=F)
        for k in range(4):
            acc = acc + a[k] * b[j][k]
        out[j] = acc
    return out


def translate(v):  # glm::translate(mat4(1), v)
    m = _mat_identity()
    m[3, :3] = np.asarray(v, dtype=F)
    return m


def scale(v):  # glm::scale(mat4(1), v)
    m = _mat_identity()
    m[0, 0], m[1, 1], m[2, 2] = [F(x) for x in v]
    return m


def yaw_pitch_roll(yaw, pitch, roll):  # glm::yawPitchRoll (gtx/euler_angles)
    ch, sh = math.cos(yaw), math.sin(yaw)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cb, sb = math.cos(roll), math.sin(roll)
    m = _mat_identity()
    m[0, 0] = ch * cb + sh * sp * sb
    m[0, 1] = sb * cp
    m[0, 2] = -sh * cb + ch * sp * sb
    m[1, 0] = -ch * sb + sh * sp * cb
    m[1, 1] = cb * cp
    m[1, 2] = sb * sh + ch * sp * cb
    m[2, 0] = sh * cp
    m[2, 1] = -sp
    m[2, 2] = ch * cp
    return m.astype(F)


def perspective_fov(fov, width, height, z_near, z_far):  # glm::perspectiveFov (RH, NO)
    h = math.cos(0.5 * fov) / math.sin(0.5 * fov)
    w = h * height / width
    m = np.zeros((4, 4), dtype=F)
    m[0, 0] = w
    m[1, 1] = h
    m[2, 2] = -(z_far + z_near) / (z_far - z_near)
    m[2, 3] = -1.0
    m[3, 2] = -(2.0 * z_far * z_near) / (z_far - z_near)
    return m


def look_at(eye, center, up):  # glm::lookAt (RH)
    eye, center, up = (np.asarray(a, dtype=np.float64) for a in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = _mat_identity().astype(np.float64)
    m[0, 0], m[1, 0], m[2, 0] = s
    m[0, 1], m[1, 1], m[2, 1] = u
    m[0, 2], m[1, 2], m[2, 2] = -f
    m[3, 0], m[3, 1], m[3, 2] = -np.dot(s, eye), -np.dot(u, eye), np.dot(f, eye)
    return m.astype(F)


def inverse(m):
    # [col][row] storage == transpose of the mathematical matrix
    return np.linalg.inv(m.astype(np.float64).T).T.astype(F)


# glm quaternion helpers used by Camera::OnUpdate (gtc/quaternion: angleAxis, cross == Hamilton product, normalize, rotate)
def _quat_angle_axis(angle, axis):
    h = F(angle) * F(0.5)
    s = F(math.sin(h))
    return np.array([math.cos(h), axis[0] * s, axis[1] * s, axis[2] * s], dtype=F)      # (w, x, y, z)


def _quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2, w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2], dtype=F)


def _quat_normalize(q):
    n = F(math.sqrt(float(np.dot(q, q))))
    return (q / n).astype(F) if n > 0 else np.array([1, 0, 0, 0], dtype=F)


def _quat_rotate(q, v):
    qv = q[1:4]
    uv = np.cross(qv, v).astype(F)
    uuv = np.cross(qv, uv).astype(F)
    return (v + ((uv * q[0]) + uuv) * F(2.0)).astype(F)


class Camera:
    """Camera.h:9-83 / Camera.cpp.  `on_resize`, `set_position`, `set_direction` keep the
    reference's semantics, including the reset of the previous-frame matrices on an explicit
    pose change (Camera.cpp:108-116) and `commit_frame()` = WalnutApp.cpp:908-909."""

    def __init__(self, vertical_fov=45.0, near_clip=0.1, far_clip=100.0):
        self.vertical_fov, self.near_clip, self.far_clip = vertical_fov, near_clip, far_clip
        self.position = np.array([0, 0, 6], dtype=F)          # Camera.cpp:15-16
        self.forward = np.array([0, 0, -1], dtype=F)
        self.width = self.height = 0
        self.projection = self.view = self.prev_projection = self.prev_view = _mat_identity()
        self.inverse_projection = self.inverse_view = _mat_identity()

    def on_resize(self, width, height):  # Camera.cpp:96-106
        if width == self.width and height == self.height:
            return
        self.width, self.height = width, height
        self.projection = perspective_fov(math.radians(self.vertical_fov), float(width), float(height), self.near_clip, self.far_clip)
        self.inverse_projection = inverse(self.projection)

    def _update_view(self):  # Camera.cpp:108-134
        self.view = look_at(self.position, self.position + self.forward, (0, 1, 0))
        self.inverse_view = inverse(self.view)
        self.prev_projection, self.prev_view = self.projection.copy(), self.view.copy()

    def set_position(self, p):
        self.position = np.asarray(p, dtype=F)
        self._update_view()

    def set_direction(self, d):
        self.forward = np.asarray(d, dtype=F)
        self._update_view()

    def commit_frame(self):
        """MainLayer::Render tail (WalnutApp.cpp:908-909): prev := current."""
        self.prev_projection, self.prev_view = self.projection.copy(), self.view.copy()

    def on_update(self, ts, keys="", mouse_delta=(0.0, 0.0)):
        """Camera::OnUpdate (Camera.cpp:18-94) with the right mouse button held: `keys` is the set of pressed keys out of
        "WSADQE", `mouse_delta` the cursor movement in pixels since the last call.  Moves / rotates the camera and
        recalculates the view — the previous-frame matrices are NOT touched (only commit_frame / an explicit pose reset
        do that), which is what makes ReSTIR's temporal reprojection land on a different pixel.  Returns `moved`."""
        delta = (F(mouse_delta[0]) * F(0.002), F(mouse_delta[1]) * F(0.002))
        moved = False
        up = np.array([0, 1, 0], dtype=F)
        fwd = self.forward.astype(F)
        right = np.cross(fwd, up).astype(F)
        speed, ts = F(5.0), F(ts)
        keys = keys.upper()
        if "W" in keys:
            self.position = (self.position + fwd * speed * ts).astype(F); moved = True
        elif "S" in keys:
            self.position = (self.position - fwd * speed * ts).astype(F); moved = True
        if "A" in keys:
            self.position = (self.position - right * speed * ts).astype(F); moved = True
        elif "D" in keys:
            self.position = (self.position + right * speed * ts).astype(F); moved = True
        if "Q" in keys:
            self.position = (self.position - up * speed * ts).astype(F); moved = True
        elif "E" in keys:
            self.position = (self.position + up * speed * ts).astype(F); moved = True
        if delta[0] != 0.0 or delta[1] != 0.0:
            pitch, yaw = delta[1] * F(0.3), delta[0] * F(0.3)                    # GetRotationSpeed() = 0.3
            q = _quat_normalize(_quat_mul(_quat_angle_axis(-pitch, right), _quat_angle_axis(-yaw, up)))
            self.forward = _quat_rotate(q, fwd)
            moved = True
        if moved:                                                                # RecalculateView(): prev_* stay
            self.view = look_at(self.position, self.position + self.forward, (0, 1, 0))
            self.inverse_view = inverse(self.view)
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
        m = _matmul(_matmul(translate(pos), yaw_pitch_roll(math.radians(rotation[1]), math.radians(rotation[0]), math.radians(rotation[2]))), scale(scale_))
        p4 = (m[0][None, :] * positions[:, 0:1] + m[1][None, :] * positions[:, 1:2]) + (m[2][None, :] * positions[:, 2:3] + m[3][None, :])
        wp = (p4[:, :3] / p4[:, 3:4]).astype(F)
        n4 = (m[0][None, :] * normals[:, 0:1] + m[1][None, :] * normals[:, 1:2]) + (m[2][None, :] * normals[:, 2:3])
        nl = np.sqrt((n4[:, 0] * n4[:, 0] + n4[:, 1] * n4[:, 1]) + n4[:, 2] * n4[:, 2])
        wn = (n4[:, :3] * (F(1) / nl)[:, None]).astype(F)
        return wp, wn

    def mesh_matrix(self, mesh_index):
        """Mesh::worldTransformMatrix of a mesh ([col][row] float32): T * yawPitchRoll(ry, rx, rz) * S (Mesh::UpdateWorldTransform)."""
        tr = self.mesh_transforms[mesh_index]
        rot = tr["rotation"]
        return _matmul(_matmul(translate(tr["pos"]), yaw_pitch_roll(math.radians(rot[1]), math.radians(rot[0]), math.radians(rot[2]))), scale(tr["scale"]))

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
        em = np.array([float(np.dot(m.get_emission(), m.get_emission())) > 0.0 for m in self.materials])
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
