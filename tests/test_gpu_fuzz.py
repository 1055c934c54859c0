"""Random API sequences (tools/fuzz_sequence.py): techniques, asynchronous + pipelined vs blocking frames, frame-index resets,
SceneManager transform edits with device refits, camera moves — every buffer a context owns must end up bit-identical whether
the frames were pipelined over two streams or rendered one by one."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location("fuzz_sequence", Path(__file__).resolve().parent.parent / "tools" / "fuzz_sequence.py")
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)


@pytest.mark.parametrize("seed", [7, 21, 42])
def test_random_api_sequence(seed):
    from fypraytracer_amd import scenes
    rng = np.random.default_rng(seed)
    n_meshes = len(scenes.hall_scene_small().meshes)
    ops = []
    for _ in range(30):
        r = rng.random()
        if r < 0.75:
            ops.append(("frame", int(rng.choice(fuzz.TECHS)), bool(rng.random() < 0.8)))
        elif r < 0.82:
            ops.append(("reset",))
        elif r < 0.93:
            ops.append(("move", int(rng.integers(0, n_meshes)), tuple(rng.uniform(-0.5, 0.5, 3).tolist()), (0.0, float(rng.uniform(-30, 30)), 0.0)))
        else:
            ops.append(("camera", tuple((np.array([-18.5, 5.5, 6.5]) + rng.uniform(-0.5, 0.5, 3)).tolist())))
    assert fuzz.play(ops, 176, 104, False) == fuzz.play(ops, 176, 104, True)
