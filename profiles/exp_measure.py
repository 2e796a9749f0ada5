#!/usr/bin/env python
"""Measure build-flag variants of the two headline kernels in ONE process, on the bench batch (BASELINE configs[2]).

    python profiles/exp_measure.py [--iters 40] [--batch 64] name=path/to/libp3d_<name>.so ...

Every variant library is opened with ctypes next to the product library and driven through the C ABI directly
(p3d_rasterize_meshes, p3d_rasterize_meshes_backward_verts) on the same device buffers; per variant:
  * per-kernel milliseconds from the library's own HIP-event profiler (p3d_profile_*: events on the launch stream);
  * forward parity against the PRODUCT library's outputs: pix_to_face equal, zbuf / bary / dists bit-equal (the product
    library is what the whole GPU test-suite pins to the oracle and the reference's device code);
  * backward parity: max relative deviation of grad_verts from the product library's (atomics make it order-dependent).
One JSON line per variant on stdout; a table on stderr.  Variants that fail parity are marked, not hidden.
"""
import argparse
import ctypes
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def open_lib(path):
    from pytorch3d_amd import _lib

    lib = ctypes.CDLL(path)
    for name in ("p3d_rasterize_meshes_workspace_bytes", "p3d_rasterize_meshes_with_cover", "p3d_rasterize_meshes_backward_verts_with_cover",
                 "p3d_rasterize_meshes_backward_workspace_bytes",
                 "p3d_profile_enable", "p3d_profile_collect", "p3d_profile_num_entries", "p3d_profile_entry", "p3d_profile_reset"):
        res, args = _lib._SIGNATURES[name]
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name in ("p3d_rasterize_meshes_short_workspace_bytes", "p3d_rasterize_meshes_workspace_need_offset"):  # round 4+
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = _lib._SIGNATURES[name]
    return lib


def snapshot(lib):
    from pytorch3d_amd._lib import c_i64

    lib.p3d_profile_collect()
    out = {}
    for i in range(lib.p3d_profile_num_entries()):
        n = c_i64(0)
        ms = ctypes.c_double(0.0)
        name = lib.p3d_profile_entry(i, ctypes.byref(n), ctypes.byref(ms))
        if name is not None and n.value > 0:
            out[name.decode()] = ms.value / n.value
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--faces-per-pixel", type=int, default=8)
    ap.add_argument("--torus-div", type=float, default=1.0, help="hetero_batch torus_div: 1.0 = SURVEY 8(d) config 3 as written (the bench headline), 1.5 = the lighter batch of rounds 1-3")
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()

    import _util as U
    import pytorch3d_amd as p3d
    from pytorch3d_amd import _lib

    d = torch.device("cuda:0")
    B, H, K = args.batch, args.image_size, args.faces_per_pixel
    blur = math.log(1.0 / 1e-4 - 1.0) * 1e-4
    verts, faces = U.hetero_batch(B, seed=0, torus_div=args.torus_div)
    m = p3d.PackedMeshes([v.to(d) for v in verts], [f.to(d) for f in faces])
    vp, fp = m.verts_packed().contiguous(), m.faces_packed().contiguous()
    fv = vp[fp].contiguous()
    F, V = int(fv.shape[0]), int(vp.shape[0])
    first, count = m.mesh_to_faces_packed_first_idx().contiguous(), m.num_faces_per_mesh().contiguous()
    nbr = torch.full((F,), -1, dtype=torch.int64, device=d)
    gen = torch.Generator().manual_seed(231)
    gz = torch.randn((B, H, H, K), generator=gen).to(d)
    gb = torch.randn((B, H, H, K, 3), generator=gen).to(d)
    gd = torch.randn((B, H, H, K), generator=gen).to(d)
    bin_size, M = 32, int(max(10000, F / 5))
    stream = ctypes.c_void_p(torch.cuda.current_stream(d).cuda_stream)

    def outputs():
        return (torch.empty((B, H, H, K), dtype=torch.int64, device=d), torch.empty((B, H, H, K), device=d),
                torch.empty((B, H, H, K, 3), device=d), torch.empty((B, H, H, K), device=d))

    cover = torch.empty((B, (H + 15) // 16, (H + 15) // 16), dtype=torch.int32, device=d)
    bws = torch.empty((int(_lib.load().p3d_rasterize_meshes_backward_workspace_bytes(B, H, H)),), dtype=torch.uint8, device=d)

    def run(lib, out, gv, ws):
        # what the L2 mirror (and bench.py) runs: the forward writes the row cover, the backward walks it
        rc = lib.p3d_rasterize_meshes_with_cover(fv.data_ptr(), first.data_ptr(), count.data_ptr(), nbr.data_ptr(), F, B, H, H, blur, K,
                                                 bin_size, M, 1, 1, 0, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                                 out[3].data_ptr(), cover.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        assert rc == 0, rc
        rc = lib.p3d_rasterize_meshes_backward_verts_with_cover(fv.data_ptr(), fp.data_ptr(), out[0].data_ptr(), gz.data_ptr(),
                                                                gb.data_ptr(), gd.data_ptr(), cover.data_ptr(), F, V, B, H, H, K, 1, 1,
                                                                gv.data_ptr(), bws.data_ptr(), bws.numel(), stream)
        assert rc == 0, rc

    product = _lib.load()
    libs = [("product", _lib.LIB_PATH, product)]
    short = {}  # variant -> list entries of a SHORT workspace (include/p3d_amd.h): name=path@entries; 0 = 1.25 x what the batch needs
    for spec in args.variants:
        name, path = spec.split("=", 1)
        if "@" in path:
            path, entries = path.split("@", 1)
            short[name] = int(entries)
        if not os.path.exists(path):
            print(f"[exp_measure] {name}: {path} missing, skipped", file=sys.stderr)
            continue
        libs.append((name, path, open_lib(path)))

    ref_out, ref_gv = None, None
    rows = []
    for name, path, lib in libs:
        ws = torch.empty((int(lib.p3d_rasterize_meshes_workspace_bytes(F, B, H, H, bin_size, M)),), dtype=torch.uint8, device=d)
        out, gv = outputs(), torch.empty((V, 3), device=d)
        ws_note = None
        if name in short:
            run(lib, out, gv, ws)
            torch.cuda.synchronize()
            at = int(lib.p3d_rasterize_meshes_workspace_need_offset(F, B, H, H, bin_size, M))
            need = int(ws[at:at + 8].view(torch.int64)[0])
            entries = short[name] or need + need // 4
            worst = ws.numel()
            del ws
            ws = torch.empty((int(lib.p3d_rasterize_meshes_short_workspace_bytes(F, B, H, H, bin_size, M, entries)),), dtype=torch.uint8,
                             device=d)
            ws_note = {"needed_entries": need, "list_entries": entries, "bytes": ws.numel(), "worst_case_bytes": worst}
        for _ in range(3):
            run(lib, out, gv, ws)
        torch.cuda.synchronize()
        lib.p3d_profile_reset()
        lib.p3d_profile_enable(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run(lib, out, gv, ws)
        e1.record()
        torch.cuda.synchronize()
        lib.p3d_profile_enable(0)
        kern = snapshot(lib)
        rec = {"variant": name, "lib": os.path.basename(path), "ms_per_step": e0.elapsed_time(e1) / args.iters,
               "kernels_ms": {k: round(v, 4) for k, v in sorted(kern.items())}}
        if ws_note is not None:
            rec["short_workspace"] = ws_note
        if ref_out is None:
            ref_out, ref_gv = out, gv
        else:
            rec["p2f_equal"] = bool(torch.equal(out[0], ref_out[0]))
            rec["floats_bit_equal"] = [bool(torch.equal(a.view(torch.int32), b.view(torch.int32))) for a, b in zip(out[1:], ref_out[1:])]
            if not rec["p2f_equal"]:
                rec["p2f_mismatches"] = int((out[0] != ref_out[0]).sum())
            # per vertex, against the vertex's largest component (edge-on faces reach 1e24: a global scale would be vacuous)
            per = ref_gv.abs().amax(dim=1, keepdim=True)
            med = float(per[per > 0].median())
            rec["grad_max_rel_dev"] = float(((gv - ref_gv).abs() / (per + 1e-6 * med)).amax())
            rec["grad_beyond_rtol5e-3"] = int(((gv - ref_gv).abs() > 5e-3 * per + 1e-6 * med).sum())
            del out, gv
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        del ws
        torch.cuda.empty_cache()
    print(f"{'variant':<14}{'step ms':>9}{'mesh_fine':>11}{'mesh_bwd':>10}  parity", file=sys.stderr)
    for r in rows:
        k = r["kernels_ms"]
        par = "" if r["variant"] == "product" else (
            f"p2f {'=' if r['p2f_equal'] else 'DIFF'} floats {'=' if all(r['floats_bit_equal']) else 'DIFF'} "
            f"grad dev {r['grad_max_rel_dev']:.1e} ({r['grad_beyond_rtol5e-3']} out)")
        print(f"{r['variant']:<14}{r['ms_per_step']:>9.3f}{k.get('mesh_fine', float('nan')):>11.4f}{k.get('mesh_backward', float('nan')):>10.4f}  {par}",
              file=sys.stderr)


if __name__ == "__main__":
    main()
