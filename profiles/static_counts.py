#!/usr/bin/env python
"""Static instruction counts of the hot kernels per build-flag variant (no GPU needed: hipcc -S for gfx950).

    python profiles/static_counts.py > profiles/r02_static_counts.txt

For the fine rasterizer the count is over the innermost candidate loop of the K = 8 perspective + clip kernel (the blocks
LLVM marks as belonging to the loop that starts with `s_ff1_i32_b64` + `v_readlane_b32`: one iteration = one candidate
face against the wave's 64 pixels); for the backward and the point rasterizer over the whole kernel (both are straight
unrolled code around short loops).  VGPRs / scratch from -Rpass-analysis=kernel-resource-usage."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch3d_amd import build as B  # noqa: E402


def compile_s(src, flags, tmp):
    out = os.path.join(tmp, "k.s")
    cmd = [B._hipcc()] + B.FLAGS + flags + ["-x", "hip", "--cuda-device-only", "-S", os.path.join(B.CSRC, src), "-o", out,
                                            "-Rpass-analysis=kernel-resource-usage"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise SystemExit(res.stderr[-3000:])
    return open(out).read().split("\n"), res.stderr


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("p3d::(anonymous namespace)::", "").replace("p3d::", "")


def kernels(lines):
    """{demangled name: list of instruction lines with their block labels}"""
    out, cur = {}, None
    for l in lines:
        m = re.match(r"^(_Z\S+):\s", l)
        if m and "@" in l:
            cur = demangle(m.group(1))
            out[cur] = []
            continue
        if cur and ".end_amdhsa_kernel" in l:
            cur = None
        if cur is not None:
            out[cur].append(l)
    return out


def resources(stderr):
    res = {}
    for b in re.split(r"remark: [^\n]*Function Name: ", stderr)[1:]:
        name = demangle(b.split("\n")[0].strip().split()[0])
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]  # noqa: E731
        res[name] = (g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"))
    return res


def is_instr(l):
    t = l.strip()
    return l.startswith("\t") and t and not t.startswith(";") and not t.startswith(".")


def count(lines, loop_of=None):
    """Counter of mnemonics; loop_of: only the blocks of the innermost loop whose header contains that mnemonic."""
    hdr = None
    if loop_of:
        for i, l in enumerate(lines):
            if loop_of in l:
                j = i
                while j > 0 and not re.match(r"^\.LBB\d+_\d+:", lines[j]):
                    j -= 1
                hdr = re.match(r"^\.(LBB\d+_\d+):", lines[j]).group(1)
                break
    ops, inside = collections.Counter(), hdr is None
    for l in lines:
        m = re.match(r"^\.(LBB\d+_\d+):(.*)", l)
        m2 = re.match(r"^; %bb\.\d+:(.*)", l)
        if hdr and (m or m2):
            c = m.group(2) if m else m2.group(1)
            inside = ("Header=" + hdr[1:] in c) or bool(m and m.group(1) == hdr)
            continue
        if inside and is_instr(l):
            ops[l.strip().split()[0]] += 1
    return ops


def summary(ops):
    valu = sum(n for k, n in ops.items() if k.startswith("v_"))
    return "total %5d  VALU %5d  (v_cndmask %4d, v_pk_* %4d, v_cmp* %4d, f64 %3d)  SALU %4d  DS %3d" % (
        sum(ops.values()), valu, sum(n for k, n in ops.items() if k.startswith("v_cndmask")),
        sum(n for k, n in ops.items() if k.startswith("v_pk_")), sum(n for k, n in ops.items() if k.startswith("v_cmp")),
        sum(n for k, n in ops.items() if "f64" in k), sum(n for k, n in ops.items() if k.startswith("s_")),
        sum(n for k, n in ops.items() if k.startswith("ds_")))


def main():
    """Product build only (the round-2 build-flag variants this script compared are folded in or deleted: the pair queues
    are the product since round 3; profiles/r02_static_counts.txt keeps the before / after record)."""
    with tempfile.TemporaryDirectory() as tmp:
        print("# mesh_fine, K = 8, perspective + clip kernel: innermost candidate loop (per wave and candidate face)")
        lines, err = compile_s("raster_mesh.hip", [], tmp)
        ks, rs = kernels(lines), resources(err)
        name = [n for n in ks if re.search(r"mesh_raster_kernel<TopKPairs<8, true, 4>, 8, true, true, true, 4, true, false>", n)][0]
        print("%-22s %s  VGPR %s scratch %s" % ("product", summary(count(ks[name], "s_ff1_i32_b64")), rs[name][0], rs[name][2]))
        print("\n# mesh_backward, K = 8 (whole kernel)")
        lines, err = compile_s("raster_mesh_bwd.hip", [], tmp)
        ks, rs = kernels(lines), resources(err)
        name = [n for n in ks if "mesh_backward_rows_kernel<8, true>" in n][0]  # per-vertex output: what the autograd nodes run
        print("%-22s %s  VGPR %s scratch %s" % ("product", summary(count(ks[name])), rs[name][0], rs[name][2]))
        print("\n# points_fine (whole kernel) per queue capacity")
        lines, err = compile_s("raster_points.hip", [], tmp)
        ks, rs = kernels(lines), resources(err)
        for K in (10, 32, 50, 100):
            names = [n for n in ks if re.search(r"point_raster_kernel<TopKPairs<%d, true, 0>, %d, true, true" % (K, K), n)]
            if not names:
                continue
            r = rs[names[0]]
            print("%-26s K=%-3d %s  VGPR %s AGPR %s scratch %s waves/SIMD %s" % ("product", K, summary(count(ks[names[0]])), r[0], r[1], r[2], r[3]))


if __name__ == "__main__":
    main()
