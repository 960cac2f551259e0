#!/usr/bin/env python3
"""Static issue-cost estimate of a gfx950 kernel from its assembly (hipcc -S --cuda-device-only).

Cost per wave-instruction in SIMD cycles, from tools/valu_ops.hip on MI355X at 4 waves per SIMD:
2.4 for fp32 add/sub/mul/fma/fmac, add_u32, mov, and/or/xor, lshrrev when no source is an SGPR; 8.2 for
rcp/rsq/sqrt/sin/cos/exp/log; 4.2 for every other VALU instruction (min/max, compares, cndmask, shifts left,
integer multiplies, conversions, 64-bit, packed) and for the fast ones with an SGPR source.

usage: isa_cost.py file.s kernel_symbol_substring [n_blocks]"""
import collections
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32",
         "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}


def cost(line):
    t = line.split()
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
    if not op.startswith("v_"):
        return 0.0, op
    if op in TRANS:
        return 8.2, op
    if op in FAST:
        srcs = " ".join(t[1:]).split(",")[1:]
        if any(re.match(r"\s*-?\|?(s\d+|s\[|vcc|exec|m0|ttmp)", s) for s in srcs):
            return 4.2, op + "(sgpr)"
        return 2.4, op
    return 4.2, op


def main():
    path, key = sys.argv[1], sys.argv[2]
    nshow = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if key in l and not l.startswith("\t") and l.split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ["entry", 0.0, 0, 0]
    by_op = collections.Counter()
    cyc_op = collections.Counter()
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith(".LBB") and t.split()[0].endswith(":"):
            blocks.append(cur)
            cur = [t.split(":")[0], 0.0, 0, 0]
            continue
        if t.startswith("."):
            continue
        c, op = cost(t)
        cur[3] += 1
        if c:
            cur[1] += c
            cur[2] += 1
            by_op[op] += 1
            cyc_op[op] += c
    blocks.append(cur)
    tot_c = sum(b[1] for b in blocks)
    tot_v = sum(b[2] for b in blocks)
    print(f"{key}: {sum(b[3] for b in blocks)} instructions, {tot_v} VALU, {tot_c:.0f} static issue cycles, "
          f"{tot_c / max(tot_v, 1):.2f} cycles per VALU instruction")
    print("by opcode (static cycles):")
    for op, c in cyc_op.most_common(28):
        print(f"  {op:28s} {by_op[op]:5d} insts {c:8.0f} cycles")
    print("largest blocks:")
    for b in sorted(blocks, key=lambda b: -b[1])[:nshow]:
        print(f"  {b[0]:14s} {b[3]:4d} insts {b[2]:4d} valu {b[1]:7.0f} cycles")


if __name__ == "__main__":
    main()
