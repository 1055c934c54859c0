"""The checker's own plumbing: orc_adopt_frame moves the complete per-pixel state of one oracle instance into another (what the
call-sequence parity tests use to mirror a scene replaced or edited on a live renderer) — a sequence continued on the adopting
instance must be the sequence rendered on one instance."""
import numpy as np
import pytest

from common import SCENES, bits_equal, settings_for
from fypraytracer_amd import capi


@pytest.mark.parametrize("tech", [capi.RESTIR_DI, capi.RESTIR_GI, capi.NEE])
def test_adopted_frame_state_continues_the_sequence(oracle_built, tech):
    from oraclelib import Oracle
    mk_scene, mk_cam = SCENES["cornell"]
    sc, W, H = mk_scene(), 40, 30
    cam = mk_cam(W, H)
    st = settings_for(tech)

    def frames(orc, first, last):
        for f in range(first, last):
            st.rand_seed = f + 1
            cam.on_update(0.05, "W", (20.0, -10.0))
            orc.set_camera(cam)
            orc.render(st)

    one = Oracle(sc, W, H)
    frames(one, 0, 4)
    cam = mk_cam(W, H)
    a = Oracle(sc, W, H)
    frames(a, 0, 2)
    b = Oracle(sc, W, H)
    b.adopt_frame(a)
    a.close()
    frames(b, 2, 4)
    assert bits_equal(one.accum(), b.accum()).all() and np.array_equal(one.image(), b.image())
    one.close(); b.close()
