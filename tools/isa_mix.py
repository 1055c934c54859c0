#!/usr/bin/env python3
"""Issue cost of the traversal kernels' own instruction mix: compiles fyprt.hip to gfx950 assembly (no GPU needed), counts the vector
instructions of the kernels named on the command line (default: the two ReSTIR DI traversal kernels) by opcode and prices every opcode
with the issue cost MEASURED by tools/microbench.hip at 6 waves per SIMD (profiles/r03/microbench.jsonl; an opcode that was not
measured is priced as the slow class).  The mean is what bench.py uses as VALU_MIX_CYCLES (static mix of the node-visit body, where four fifths of
the kernels' instructions are issued; --whole: the whole kernel).   usage: python tools/isa_mix.py [--whole] [kernel-name-substring ...]"""
import collections
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "fypraytracer_amd" / "csrc"


def measured():
    cost = {}
    f = ROOT / "profiles" / "r03" / "microbench.jsonl"
    for line in f.read_text().splitlines():
        d = json.loads(line)
        if d.get("bench") == "issue" and d["waves_per_simd"] == 6 and " " not in d["instr"] and "(" not in d["instr"]:
            cost[d["instr"].replace("_sdwa", "")] = d["cycles_at_2400MHz"]
    cost.pop("v_cndmask_b32", None)            # the stand-alone loop measured a VCC read hazard, not the instruction: use the dependent pair's cost
    for line in f.read_text().splitlines():
        d = json.loads(line)
        if d.get("bench") == "issue" and d["waves_per_simd"] == 6 and d["instr"].startswith("v_cmp_le_f32+v_cndmask_b32"):
            cost["v_cndmask_b32"] = d["cycles_at_2400MHz"]
    return cost


def main():
    want = [a for a in sys.argv[1:] if not a.startswith("--")] or ["k_di_part2_traceILb0", "k_di_part1ILb0"]
    asm = "/tmp/fyprt_mix.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
                    "-S", "--cuda-device-only", "fyprt.hip", "-o", asm], cwd=CSRC, check=True, capture_output=True)
    cost = measured()
    slow = sorted(v for k, v in cost.items() if v > 3.5)
    slow_default = slow[len(slow) // 2]
    text = Path(asm).read_text()
    for w in want:
        m = re.search(r"^(_ZN2rt\w*" + re.escape(w) + r"\w*):.*?s_endpgm", text, re.S | re.M)
        if not m:
            print(json.dumps({"kernel": w, "error": "not found"}))
            continue
        lines = m.group(0).splitlines()
        if "--whole" not in sys.argv:
            # the node visit: from the node's first load (the last global_load_dwordx4 before the first byte -> float conversion) to the pop that follows the
            # pushes (the first ds_read_b32 after the last v_max_f64 of the sorting network) — the straight-line body plus its rare-path blocks
            cv = next(k for k, l in enumerate(lines) if "v_cvt_f32_ubyte" in l)
            b = max(k for k, l in enumerate(lines[:cv]) if "global_load_dwordx4" in l) - 12
            mx = max(k for k, l in enumerate(lines) if "v_max_f64" in l)
            e = next(k for k, l in enumerate(lines) if k > mx and "ds_read_b32" in l)
            lines = lines[b:e + 1]
        ops = collections.Counter()
        for line in lines:
            t = line.strip().split()
            if t and t[0].startswith("v_"):
                ops[re.sub(r"_e(32|64)$|_sdwa$|_dpp$", "", t[0])] += 1
        total = sum(ops.values())
        cyc = 0.0
        unknown = collections.Counter()
        for op, n in ops.items():
            base = op
            if base not in cost:
                fam = {"v_cmp": "v_cmp_le_f32", "v_cvt_f32_ubyte": "v_cvt_f32_ubyte1", "v_min3": "v_max3_f32", "v_max3": "v_max3_f32", "v_sub_u32": "v_add_u32", "v_subrev_u32": "v_add_u32",
                       "v_xor_b32": "v_and_b32", "v_lshrrev_b32": "v_lshlrev_b32", "v_ashrrev_i32": "v_lshlrev_b32", "v_mul_legacy": "v_mul_f32", "v_subrev_f32": "v_sub_f32"}
                base = next((v for k, v in fam.items() if op.startswith(k)), None)
            if base in cost:
                cyc += n * cost[base]
            else:
                cyc += n * slow_default
                unknown[op] += n
        print(json.dumps({"kernel": m.group(1)[:60], "vector_instructions_static": total, "mean_issue_cycles_at_2400MHz": round(cyc / total, 3),
                          "priced_as_slow_class_unmeasured": dict(unknown.most_common(12)), "top_opcodes": dict(ops.most_common(14))}))


if __name__ == "__main__":
    main()
