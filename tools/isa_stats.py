#!/usr/bin/env python3
"""Instruction mix of the kernels in a device assembly file (hipcc ... -S --offload-device-only -o x.s yart_hip.hip):
per kernel, static counts by class (VALU / transcendental-and-quarter-rate / fp64 / SALU / VMEM / LDS / scratch) and
the register / scratch figures of its .amdhsa block.   tools/isa_stats.py x.s [kernel-substring ...]"""
import re
import sys
from collections import Counter

QUARTER = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32",
           "v_sin_f32", "v_cos_f32", "v_mad_u64_u32", "v_mad_i64_i32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")


def classify(op):
    if op.startswith(("scratch_", "buffer_")):
        return "scratch/buffer"
    if op.startswith(("global_", "flat_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "salu" if not op.startswith(("s_waitcnt", "s_nop", "s_load", "s_buffer")) else ("smem" if "load" in op else "wait")
    if op.startswith("v_"):
        if op in QUARTER or op.startswith(QUARTER):
            return "valu_quarter"
        if "_f64" in op:
            return "valu_f64"
        if op.startswith(("v_div_", "v_rcp", "v_sqrt")):
            return "valu_div"
        if op.startswith("v_cndmask"):
            return "valu_cndmask"
        if op.startswith(("v_accvgpr", "v_mov")):
            return "valu_mov"
        return "valu"
    return "other"


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    text = open(path).read()
    # kernels: "<name>:" ... ".end_amdhsa_kernel"? Use .amdhsa_kernel blocks for metadata, function labels for bodies
    bodies = {}
    cur = None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = m.group(1); bodies[cur] = []
            continue
        if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
            cur = None
        if cur and line.startswith("\t") and not line.startswith("\t."):
            op = line.strip().split()[0]
            if not op.startswith(";"):
                bodies[cur].append(op)
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        blk = m.group(2)
        g = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", blk) or [None, "?"])[1]
        meta[m.group(1)] = dict(vgpr=g("next_free_vgpr"), agpr=g("accum_offset"), sgpr=g("next_free_sgpr"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))
    for name, ops in bodies.items():
        if name not in meta or (pats and not any(p in name for p in pats)):
            continue
        c = Counter(classify(o) for o in ops)
        top = Counter(ops).most_common(14)
        print(f"== {name}\n   {meta[name]}  total {len(ops)}")
        print("   " + "  ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
        print("   top: " + ", ".join(f"{o}:{n}" for o, n in top))


if __name__ == "__main__":
    main()
