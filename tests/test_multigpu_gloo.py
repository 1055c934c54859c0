"""N > 1 path on CPU: world_size-2 `gloo` ranks render their row bands (the oracle stands in for the GPU
kernels — same row/halo semantics as fyprt_set_rows), all-gather the RGBA8 bands with the product's
`multigpu.gather_image`, and the assembled frame is compared with a single-rank full-frame render."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tech, frames, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    from common import settings_for
    from fypraytracer_amd import multigpu, scenes
    from oraclelib import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 64, 48
    sc = scenes.cornell_box()
    cam = scenes.cornell_camera(W, H)
    st = settings_for(tech)
    r0, r1 = multigpu.band_rows(H, world, rank)
    halo = multigpu.halo_rows(st, tech, world)
    o = Oracle(sc, W, H)
    o.set_camera(cam)
    full = None
    for f in range(frames):
        st.rand_seed = f + 1
        o.render(st, rows=(r0, r1), halo=halo)
        per = (H + world - 1) // world
        band = np.zeros(per * W, dtype=np.int32)
        band[: (r1 - r0) * W] = o.image()[r0:r1].reshape(-1).view(np.int32)
        full = multigpu.gather_image(torch.from_numpy(band), H, W, world, dist)
    if rank == 0:
        np.save(out_path, full.numpy().view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tech,frames", [(2, 2), (7, 1)])
def test_two_rank_band_split_matches_single_rank(oracle_built, tmp_path, tech, frames):
    """COSINE (pure per-pixel map): exact for any number of frames.  RESTIR_DI: exact on frame 1 — with a
    halo >= the spatial radius every Part-2 neighbour was produced locally by Part 1; later frames differ
    near the band border because temporal history stays per rank (north-star design)."""
    from common import settings_for
    from fypraytracer_amd import scenes
    from oraclelib import Oracle
    out = tmp_path / "gathered.npy"
    mp.spawn(_worker, args=(2, _free_port(), tech, frames, str(out)), nprocs=2, join=True)
    got = np.load(out)
    W, H = 64, 48
    o = Oracle(scenes.cornell_box(), W, H)
    o.set_camera(scenes.cornell_camera(W, H))
    st = settings_for(tech)
    for f in range(frames):
        st.rand_seed = f + 1
        o.render(st)
    assert np.array_equal(got, o.image())


def test_band_partition_covers_the_frame():
    from fypraytracer_amd import multigpu
    for H in (1080, 2160, 1081, 7):
        for world in (1, 2, 3, 4, 8):
            rows = [multigpu.band_rows(H, world, r) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == H
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
