#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of fyprt.hip (hipcc -Rpass-analysis=kernel-resource-usage); runs without a GPU.
  usage: python tools/kernel_resources.py [--all] [extra hipcc flags]     (default: render kernels rt::k_* only, without the builders)"""
import re
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "fypraytracer_amd" / "csrc"


def main():
    args = [a for a in sys.argv[1:] if a != "--all"]
    show_all = "--all" in sys.argv[1:]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-Rpass-analysis=kernel-resource-usage", *args, "-c", "fyprt.hip", "-o", "/tmp/fyprt_res.o"]
    err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'kernel':44s} {'VGPR':>5s} {'spill':>6s} {'scratch':>8s} {'waves':>6s} {'LDS':>6s}")
    bad = 0
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if not show_all and (not n.startswith("rt::k_") or "lbvh" in n):
            continue
        sc = int(r.get("ScratchSize [bytes/lane]", 0))
        bad += sc > 0
        print(f"{n:44s} {r.get('VGPRs', '?'):>5s} {r.get('VGPRs Spill', r.get('VGPR Spill', '?')):>6s} {sc:8d} {r.get('Occupancy [waves/SIMD]', '?'):>6s} {r.get('LDS Size [bytes/block]', '?'):>6s}")
    print(f"kernels with scratch: {bad}")


if __name__ == "__main__":
    main()
