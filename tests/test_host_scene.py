"""the host producers of the kernel inputs — Scene::AddNewMeshToScene's transform, Mesh::UpdateWorldTransform
and SceneManager::PerformAllSceneUpdates (Scene.cpp:9-92, SceneManager.cpp:6-130).  Since r03 there is ONE implementation of their
arithmetic, host/HostTypes.h: the C++ facade uses it directly (here through host/scene_check) and the Python containers of scene.py
call it through libfyprt_host.so.  The reference has no fixtures for these and glm is unpinned, so the pins are: the Python path
(its own bookkeeping: vertex ranges, queues, material rewrites) produces what the C++ Scene / SceneManager produce, bit for bit; a mesh
moved through the SceneManager equals the same mesh added with the new transform; and the reference's queue quirks are kept."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from fypraytracer_amd import capi, scenes
from fypraytracer_amd.scene import Material, Scene

HOST = Path(__file__).resolve().parent.parent / "fypraytracer_amd" / "host"


@pytest.fixture(scope="module")
def scene_check():
    exe = HOST / "scene_check"
    if not exe.exists():
        subprocess.run(["bash", str(HOST / "build.sh")], check=True, capture_output=True)
    return str(exe)


def _fixed_mesh():
    i = np.arange(7, dtype=np.float32)
    f = np.float32
    pos = np.stack([f(0.25) * i - f(0.7), f(0.1) * i * i - f(0.3), f(1.0) - f(0.37) * i], 1).astype(np.float32)
    nrm = np.stack([f(0.3) + f(0.1) * i, f(-0.5) + f(0.2) * i, f(0.8) - f(0.15) * i], 1).astype(np.float32)
    uv = np.stack([f(0.1) * i, f(1.0) - f(0.1) * i], 1).astype(np.float32)
    idx = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6]], np.uint32)
    return pos, nrm, uv, idx


def _cpp(scene_check, *args):
    out = subprocess.run([scene_check] + [repr(float(a)) for a in args], check=True, capture_output=True, text=True).stdout.splitlines()
    head = [l for l in out if not l.startswith(("0x", "-0x"))]
    rows = np.array([[float.fromhex(t) for t in l.split()] for l in out if l.startswith(("0x", "-0x"))], dtype=np.float32)
    return head, rows


@pytest.mark.parametrize("tr", [(0, -3, 0, 90, 0, 0, 1, 1, 1), (1.5, 0.25, -2, 33, 71, -12, 0.5, 2, 1.25), (0, 0, 0, 0, 0, 0, 1, 1, 1)])
def test_cpp_scene_and_python_containers_agree(scene_check, tr):
    head, rows = _cpp(scene_check, *tr)
    assert head[:2] == ["first_call_dirty 1", "second_call_dirty 0"]          # the queues start with 20 default entries (SceneManager.h:25-26)
    sc = Scene()
    sc.materials = [Material(), Material()]
    sc.add_new_mesh_to_scene(*_fixed_mesh(), pos=tr[0:3], rotation=tr[3:6], scale_=tr[6:9], material_index=0)
    got = np.concatenate([sc.world_vertices["position"], sc.world_vertices["normal"]], 1)
    assert np.array_equal(got.view(np.uint32), rows.view(np.uint32))


def test_moving_a_mesh_equals_adding_it_there(scene_check):
    a, b = (0, -3, 0, 90, 0, 0, 1, 1, 1), (0.5, -2.5, 0.25, 45, 15, -30)
    head, rows = _cpp(scene_check, *a, *b)
    assert "moved_dirty 1 material 1" in head
    sc = Scene()
    sc.materials = [Material(), Material()]
    mi = sc.add_new_mesh_to_scene(*_fixed_mesh(), pos=a[0:3], rotation=a[3:6], scale_=a[6:9], material_index=0)
    m = sc.manager()
    assert m.perform_all_scene_updates(sc) is True and m.perform_all_scene_updates(sc) is False
    m.set_mesh_transform(sc, mi, pos=b[0:3], rotation=b[3:6])
    m.set_mesh_material(sc, mi, 1)
    assert m.perform_all_scene_updates(sc) is True
    assert (sc.triangles["materialIndex"] == 1).all()
    direct = Scene()
    direct.materials = [Material(), Material()]
    direct.add_new_mesh_to_scene(*_fixed_mesh(), pos=b[0:3], rotation=b[3:6], scale_=a[6:9], material_index=1)
    for k in ("position", "normal", "uv"):
        assert np.array_equal(sc.world_vertices[k].view(np.uint32), direct.world_vertices[k].view(np.uint32))
    got = np.concatenate([sc.world_vertices["position"], sc.world_vertices["normal"]], 1)
    assert np.array_equal(got.view(np.uint32), rows.view(np.uint32))           # and the C++ SceneManager did the same


def test_dirty_flag_drives_a_re_upload():
    """The facade re-uploads (and the library rebuilds its BVH) only when the flag is raised (Renderer.cu:61-69)."""
    sc = scenes.cornell_box()
    ctx = capi.Context(-1)
    ctx.upload_scene(sc)
    before = ctx.export_bvh()

    class Flag:
        dirty = False

        def set_scene_to_be_updated_flag(self, f):
            self.dirty = f
    r = Flag()
    m = sc.manager()
    m.perform_all_scene_updates(sc, r)
    assert r.dirty
    r.dirty = False
    m.perform_all_scene_updates(sc, r)
    assert not r.dirty                                                         # nothing queued: no rebuild
    m.set_mesh_transform(sc, len(sc.meshes) - 1, pos=(0.1, 0.0, 0.0))
    m.perform_all_scene_updates(sc, r)
    assert r.dirty
    ctx.upload_scene(sc)
    after = ctx.export_bvh()
    assert not np.array_equal(before["tris"]["v0"], after["tris"]["v0"])       # the moved mesh's triangles are where they now are
    ctx.close()


def test_camera_is_the_facades_camera():
    """scene.Camera is a handle to HostTypes.h's Camera: explicit pose -> previous-frame matrices reset (Camera.cpp:108-116), OnUpdate moves
    without touching them, commit_frame copies; the inverse matrices really are inverses."""
    cam = scenes.hall_camera(320, 180)
    assert np.array_equal(cam.prev_view, cam.view) and np.array_equal(cam.prev_projection, cam.projection)
    before = cam.view.copy()
    assert cam.on_update(0.016, "WD", (12.0, -7.0)) is True
    assert not np.array_equal(cam.view, before) and np.array_equal(cam.prev_view, before)          # reprojection now sees last frame's view
    cam.commit_frame()
    assert np.array_equal(cam.prev_view, cam.view)
    for m, inv in ((cam.view, cam.inverse_view), (cam.projection, cam.inverse_projection)):
        prod = m.astype(np.float64).T @ inv.astype(np.float64).T                                  # [col][row] storage = transposed matrices
        assert np.allclose(prod, np.eye(4), atol=1e-5)
    assert cam.on_update(0.016, "", (0.0, 0.0)) is False
