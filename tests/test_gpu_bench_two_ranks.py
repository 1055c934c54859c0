"""bench.py's own N > 1 path — band split, ExternalStream, double-buffered asynchronous gather, fence, max-over-ranks timing — driven by
TWO processes (torch.distributed.run, gloo) that share the box's one GPU: the frame rank 0 gathers must be the frame one context
renders alone.  (RCCL itself cannot be rehearsed here: it refuses two ranks on one device.  The gather through RCCL with a
one-rank communicator runs in tests/test_gpu_group.py and in `bench.py --rehearse-gather`.)"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from fypraytracer_amd import capi, scenes

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("tech", [2, 7])
def test_two_ranks_on_one_gpu_gather_the_single_context_frame(tmp_path, tech):
    W, H, warm, steps = 320, 200, 1, 3
    out = tmp_path / "frame.npy"
    if tech == 7:
        warm, steps = 0, 1                         # ReSTIR: the halo-recompute split equals the single-GPU frame on frame 1
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device", "--scene", "hall_small", "--width", str(W), "--height", str(H),
           "--technique", str(tech), "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--dump-image", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == steps and j["value"] > 0 and j["scaling"] == "strong"
    got = np.load(out)
    ctx = capi.Context(0)
    ctx.resize(W, H)
    ctx.upload_scene(scenes.hall_scene_small())
    ctx.set_camera(scenes.hall_camera(W, H))
    st = capi.Settings(technique=tech, light_bounces=2 if tech != 7 else 1, sample_count=1, sky_color=(0.0, 0.0, 0.0), light_candidate_count=4,
                       use_temporal_reuse=1, use_spatial_reuse=1, temporal_history_limit=2, spatial_neighbor_num=5, spatial_neighbor_radius=30)
    for f in range(warm + steps):
        st.rand_seed = f + 1
        ctx.render(st)
    img, _ = ctx.readback(want_accum=False)
    ctx.close()
    assert np.array_equal(got, img)
