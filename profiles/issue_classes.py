#!/usr/bin/env python
"""Static VALU issue-class mix of the hot kernels (no GPU needed: hipcc -S for gfx950).

    python profiles/issue_classes.py > profiles/r03/issue_classes.txt

profiles/microbench/valu_classes.hip measured two issue classes on gfx950 (nanoseconds of one SIMD per wave64 instruction
with >= 2 waves resident, profiles/microbench/valu_classes_mi355x.txt):
  fast  ~1.0 ns   v_mul_f32 v_add_f32 v_sub_f32 v_subrev_f32 v_and_b32 v_or_b32 v_xor_b32 v_add_u32 v_sub_u32 v_lshrrev_b32
                  v_mov_b32 -- with VGPR / literal operands (VOP3 forms with modifiers: 1.24)
  slow  ~1.9 ns   everything else it tried: v_fma / v_fmac, v_max / v_min, v_lshlrev_b32, all three-operand integer ops,
                  conversions, every compare, v_cndmask, all DPP forms, v_mov_b64, all v_pk_*, all f64, v_readlane, v_mbcnt --
                  and the fast ones when an operand is an SGPR; v_rcp / v_sqrt / v_exp 3.5 ns, v_swap_b32 3.6 ns
and that, in straight streams of independent instructions, a slow instruction followed by a fast one costs the slow one's
time (pairs at 2.1 ns).  This script prints, per basic block of a kernel's hot loop, how many instructions of each class it
holds, and what the stream model max(1.0 ns x all, 1.9 ns x slow) would give.

CAVEAT, measured (profiles/r03/exp_depth_bound/): that model did NOT predict the fine rasterizer.  A variant that removed
~50 slow-class instructions from part of the evaluations at the price of ~19 fast-class ones on all of them executed 4 %
MORE instructions and ran 2 % SLOWER; interleaving fast moves among the queue's slow ones (+5 % instructions) changed
nothing.  With four waves per SIMD that spend 42 % of their time parked (barriers between the four sub-tile waves of a
workgroup, LDS round trips) the kernel's time follows the instruction COUNT: a single wave issues one VALU instruction
per ~6 nominal cycles whatever its class (the W = 1 columns of the microbenchmarks), and there are not enough ready waves
for the classes' pipe costs to be what binds.  Read the class columns as a map of the code, the estimate as an upper bound
on what class-aware rewriting could buy if the kernel were bound by the VALU pipes.
"""
import collections
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import static_counts as S  # noqa: E402

FAST = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_lshrrev_b32", "v_mov_b32"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_swap_b32"}
T_FAST, T_SLOW, T_QUARTER = 1.0, 1.9, 3.5


def classify(line):
    """'fast' | 'slow' | 'quarter' | None (not a VALU instruction)"""
    t = line.strip()
    op = t.split()[0]
    if not op.startswith("v_"):
        return None
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in QUARTER:
        return "quarter"
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return "slow"
    if base in FAST:
        operands = t[len(op):].split(";")[0]
        srcs = operands.split(",")[1:]
        if any(re.match(r"\s*-?\|?(s\d+|s\[|vcc|exec|ttmp|m0)", s) for s in srcs):
            return "slow"  # an SGPR operand: measured at the slow rate
        return "fast"
    return "slow"


def blocks_of_loop(lines, loop_of):
    """[(label, [instruction lines])] of the innermost loop whose header block contains `loop_of`; whole kernel if None"""
    hdr = None
    if loop_of:
        for i, l in enumerate(lines):
            if loop_of in l:
                j = i
                while j > 0 and not re.match(r"^\.LBB\d+_\d+:", lines[j]):
                    j -= 1
                hdr = re.match(r"^\.(LBB\d+_\d+):", lines[j]).group(1)
                break
    out, cur, inside = [], None, hdr is None
    if hdr is None:
        cur = ("entry", [])
        out.append(cur)
    for l in lines:
        m = re.match(r"^\.(LBB\d+_\d+):(.*)", l)
        m2 = re.match(r"^; (%bb\.\d+):(.*)", l)
        if m or m2:
            label = m.group(1) if m else m2.group(1)
            c = m.group(2) if m else m2.group(2)
            inside = hdr is None or ("Header=" + hdr[1:] in c) or bool(m and m.group(1) == hdr)
            cur = (label, [])
            if inside:
                out.append(cur)
            continue
        if inside and cur is not None and S.is_instr(l):
            cur[1].append(l)
    return out


def report(title, blocks, min_valu=6):
    print(title)
    tot = collections.Counter()
    for label, ins in blocks:
        c = collections.Counter(filter(None, (classify(l) for l in ins)))
        other = sum(1 for l in ins if classify(l) is None)
        n = c["fast"] + c["slow"] + c["quarter"]
        tot.update(c)
        if n < min_valu:
            continue
        slow_ns = T_SLOW * c["slow"] + T_QUARTER * c["quarter"]
        est = max(T_FAST * n, slow_ns)
        slow_ops = collections.Counter(l.strip().split()[0] for l in ins if classify(l) in ("slow", "quarter"))
        top = ", ".join(f"{k} {v}" for k, v in slow_ops.most_common(7))
        print(f"  {label:<12} VALU {n:4d} = fast {c['fast']:4d} + slow {c['slow']:4d} + quarter-rate {c['quarter']:2d} | other {other:3d} | "
              f"~{est:6.1f} ns ({'slow-bound' if slow_ns > T_FAST * n else 'count-bound'}) | {top}")
    n = tot["fast"] + tot["slow"] + tot["quarter"]
    print(f"  {'all blocks':<12} VALU {n:4d} = fast {tot['fast']:4d} + slow {tot['slow']:4d} + quarter-rate {tot['quarter']:2d}\n")


def main():
    with tempfile.TemporaryDirectory() as tmp:
        lines, _ = S.compile_s("raster_mesh.hip", [], tmp)
        ks = S.kernels(lines)
        name = [n for n in ks if re.search(r"mesh_raster_kernel<TopKPairs<8, true, 4>, 8, true, true, true, 4, true, false>", n)][0]
        report("# mesh_fine, K = 8, perspective + clip kernel: blocks of the innermost candidate loop (one iteration = one candidate face "
               "against the wave's 64 pixels)", blocks_of_loop(ks[name], "s_ff1_i32_b64"))
        lines, _ = S.compile_s("raster_mesh_bwd.hip", [], tmp)
        ks = S.kernels(lines)
        name = [n for n in ks if "mesh_backward_rows_kernel<8, true>" in n][0]
        report("# mesh_backward_rows_kernel<8, to vertices>: all blocks (one pass of the step loop = 64 samples)", blocks_of_loop(ks[name], None), 10)
        lines, _ = S.compile_s("raster_points.hip", [], tmp)
        ks = S.kernels(lines)
        names = [n for n in ks if re.search(r"point_raster_kernel<TopKPairs<10, true, 0>, 10, true, true", n)]
        if names:
            report("# points_fine, K = 10: all blocks", blocks_of_loop(ks[names[0]], None), 10)


if __name__ == "__main__":
    main()
