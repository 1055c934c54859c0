"""bench.py's output contract (the driver parses it): ONE JSON line on stdout with the agreed keys, `value` consistent with
`ms_per_step`, a roofline object whose fraction cannot exceed 1 and a CPU baseline — run here on a small frame of the small hall so
that it takes seconds (the real sizes are bench.py's defaults)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_prints_one_json_line_with_the_contract_keys():
    env = dict(os.environ, OMP_NUM_THREADS="8")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--scene", "hall_small", "--width", "256", "--height", "144", "--steps", "4", "--warmup", "2",
                        "--cpu-frames", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    b = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 4 and b["warmup"] == 2 and b["higher_is_better"] is True and b["vs_baseline"] is None
    assert b["unit"] == "Mrays/s" and b["dtype"] == "f32" and b["data"] == "synthetic" and "workload" in b["config"] and "model" not in b["config"]
    assert b["value"] > 0 and b["ms_per_step"] > 0
    rays = b["config"].get("rays_per_frame") or b.get("rays_per_frame")
    if rays:                                                                 # value = rays of a frame / time of a frame
        assert abs(b["value"] - rays / b["ms_per_step"] / 1e3) <= 0.02 * b["value"]
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == b["unit"]
