#!/usr/bin/env python3
"""Static issue-cost weighting of the traversal kernels' vector instructions, for the VALU issue-slot occupancy that
bench.py reports (roofline.valu_issue_occupancy).  SQ_ACTIVE_INST_VALU counts instructions, not cycles (profiles/README.md),
and the per-class counters put v_fma_mix_f32 / v_pk_fma_f32 (4.5 cycles) in the same class as v_fma_f32 (2.8), so the
instruction mix is taken from the compiled kernel instead: every VALU instruction of the kernel's body is priced with the
issue cost tools/micro/issue_rate.hip measured at 4 wavefronts per SIMD (the kernel's occupancy):
    2.8 cycles  v_fma_f32 v_fmac_f32 v_mul_f32 v_add_f32 v_sub(rev)_f32 v_mov_b32 v_add_u32 v_sub(rev)_u32
    4.5 cycles  everything else (v_fma_mix_f32, v_pk_*, v_min/max(3), v_cmp_*, v_cndmask, logic / shift / bit ops, conversions ...)
and the mean over the body is used as the cost per executed instruction (a static weighting: every instruction of the
body counts once, whatever its trip count).  Prints {kernel: {"valu_instructions", "mean_issue_cycles"}} as JSON."""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHEAP = {"v_fma_f32", "v_fmac_f32_e32", "v_fmac_f32_e64", "v_mul_f32_e32", "v_mul_f32_e64", "v_add_f32_e32", "v_add_f32_e64",
         "v_sub_f32_e32", "v_sub_f32_e64", "v_subrev_f32_e32", "v_mov_b32_e32", "v_add_u32_e32", "v_add_u32_e64",
         "v_sub_u32_e32", "v_sub_u32_e64", "v_subrev_u32_e32"}
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize".split()


def kernel_costs(src=os.path.join(ROOT, "path-tracer_amd", "csrc", "device", "kernels_trace_wide.hip")):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-device-only", "-S", src, "-o", out], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read()
    res = {}
    for m in re.finditer(r"^(_ZN3ptd12k_trace_wideILb([01])ELb([01])EEE\w*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name = "k_trace_wide_" + ("occluded" if m.group(2) == "1" else "closest") + ("_count" if m.group(3) == "1" else "")
        ops = [l.split()[0] for l in m.group(4).splitlines() if re.match(r"\s+v_", l)]
        cost = [2.8 if o in CHEAP else 4.5 for o in ops]
        res[name] = {"valu_instructions": len(ops), "mean_issue_cycles": sum(cost) / max(len(cost), 1),
                     "cheap_fraction": sum(1 for c in cost if c < 3) / max(len(cost), 1)}
    return res


if __name__ == "__main__":
    print(json.dumps(kernel_costs(), indent=1))
