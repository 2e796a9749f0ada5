"""The `pytorch3d._C` operator surface for the rasterization hot path, on MI355X.

Same names, positional signatures, return conventions and error behaviour as the reference's
pybind module (pytorch3d/csrc/ext.cpp:38-73); each function validates like the reference's
dispatcher, allocates outputs with torch (caching allocator, uninitialised -- the kernels write
every element), and calls the C ABI of libp3d_amd.so on torch's current HIP stream.

GPU tensors only: there is no CPU implementation and no fallback in this package.
"""
import ctypes
import os

import torch

from . import _lib

# constants pytorch3d/renderer/points/pulsar/renderer.py reads at import (ext.cpp:180-185)
EPS = 1e-6
MAX_FLOAT = 3.4e38
MAX_INT = 2147483647
MAX_UINT = 4294967295
MAX_USHORT = 65535
PULSAR_MAX_GRAD_SPHERES = 128

kMaxPointsPerPixel = 150
kMaxItemsPerBin = 22


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() > 0 else None)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _need_gpu(t, name):
    if not t.is_cuda:
        # mirrors CHECK_CUDA (pytorch3d_cutils.h:12-21) -- but there is no CPU path to fall to.
        raise RuntimeError(f"{name} must be a CUDA/HIP tensor: pytorch3d_amd implements the GPU path only.")


def _same_device(*named):
    dev = named[0][1].device
    for name, t in named:
        _need_gpu(t, name)
        if t.device != dev:
            raise RuntimeError(f"Expected all tensors to be on the same GPU, but {name} is on {t.device} "
                               f"and {named[0][0]} is on {dev}")
    return dev


def _check_fragments(who, pix_to_face, **floats):
    """The fragment tensors a fused kernel is about to reinterpret: pix_to_face must be int64 (rasterize_points returns
    int32 idx -- a mix-up would be misread, not rejected, by a kernel that takes a raw pointer), the float tensors
    float32, and everything on pix_to_face's GPU."""
    _need_gpu(pix_to_face, "pix_to_face")
    if pix_to_face.dtype != torch.int64:
        raise RuntimeError(f"{who}: pix_to_face must be int64, got {pix_to_face.dtype}")
    for name, t in floats.items():
        _need_gpu(t, name)
        if t.dtype != torch.float32:
            raise RuntimeError(f"{who}: {name} must be float32, got {t.dtype}")
        if t.device != pix_to_face.device:
            raise RuntimeError(f"Expected all tensors to be on the same GPU, but {name} is on {t.device} and pix_to_face is on "
                               f"{pix_to_face.device}")


def _c(t, dtype):
    if t.dtype != dtype:
        raise RuntimeError(f"expected dtype {dtype}, got {t.dtype}")
    return t.contiguous()


def _hw(image_size):
    h, w = image_size
    return int(h), int(w)


def _check_k(K):
    if K > kMaxPointsPerPixel:
        raise RuntimeError(f"Must have points_per_pixel <= {kMaxPointsPerPixel}")


def _num_bins(H, W, bin_size, who):
    by, bx = 1 + (H - 1) // bin_size, 1 + (W - 1) // bin_size
    if by >= kMaxItemsPerBin or bx >= kMaxItemsPerBin:
        raise RuntimeError(f"In {who} got num_bins_y: {by}, num_bins_x: {bx}, ; that's too many!")
    return by, bx


def _workspace(nbytes, device):
    """Scratch for the coarse stage, sized by the C ABI (include/p3d_amd.h).  A request the allocator cannot serve is reported
    with what to do about it instead of as a bare out-of-memory error."""
    n = max(int(nbytes), 256)
    try:
        return torch.empty((n,), dtype=torch.uint8, device=device)
    except RuntimeError as e:  # torch.OutOfMemoryError is a RuntimeError
        if "out of memory" not in str(e).lower():
            raise
        raise RuntimeError(f"the coarse stage's workspace of {n / 2**30:.1f} GiB does not fit on {device}: rasterize the "
                           "batch in smaller pieces (pytorch3d_amd.sharding.partition, or bench.py --jobs style sub-batches of 64 "
                           "meshes), pass a smaller max_faces_per_bin (the worst case is capped by N * bins * max_faces_per_bin * 4 "
                           "bytes) or set pytorch3d_amd._C.SHORT_WORKSPACE = 'always'") from e


# Short workspaces for rasterize_meshes and rasterize_points (include/p3d_amd.h "Short workspaces"; csrc/binning.h).  The worst case of the bin
# lists is 100-1000 x what a batch needs (bench batch: 1.3 GB against 8.4 MB; 512 such meshes in one call: 42 GB), and
# what it needs is only known on the device.  With a short workspace the call sizes the lists from what the SAME call shape
# needed before (read back asynchronously: no host sync anywhere) plus a quarter of headroom, and the library decides on
# the device whether they fit (if not, its naive kernel writes the same bits, once; the next call has the new size).
#   'auto'    short when the worst case is above SHORT_WORKSPACE_ABOVE bytes (the stand-by launch costs 0.026 ms on the 65 536
#             tiles of the bench batch, 1 % of its step: profiles/r04/r04c8/measure.json)
#   'always'  every binned call        'never'  always the worst case
SHORT_WORKSPACE = os.environ.get("P3D_SHORT_WORKSPACE", "auto")
SHORT_WORKSPACE_ABOVE = 2 << 30
SHORT_WORKSPACE_FIRST_GUESS = None  # tests: list entries of a shape's first call (default 32 per face + 256k)
WORKSPACE_STATS = {"short_calls": 0, "last_bytes": 0, "last_entries": None}
_NEEDS = {}  # call shape -> _Need


class _Need:
    """What one call shape needed, refreshed from the device without ever waiting for it.

    `entries` is a running maximum with slow decay (a report below it pulls it down by an eighth of the difference), not the
    last report: a scene whose lists grow -- mesh fitting, a camera zooming in -- would otherwise size every call by a stale
    smaller number.  A report is requested on every call while the last one was within 20 % of the capacity the call was given
    (or above it: the naive kernel ran), and every 8th call otherwise (up to seven consecutive calls could
    run the every-tile-tests-every-face stand-by).  Caveat (documented in INTEGRATION.md): the stand-by is bit-identical to the
    binned path only while no bin reaches max_faces_per_bin -- the binned path drops a bin's faces beyond that limit as the
    reference does (rasterize_coarse.cu:186-201), the naive kernel has no bins."""

    def __init__(self):
        self.entries = None   # running maximum (slow decay) of what the calls that reported back needed
        self.last = None      # the last report itself
        self.capacity = None  # list entries of the call that produced the pending / last report
        self.calls = 0
        self.pinned = None
        self.event = None

    def collect(self):
        if self.event is not None and self.event.query():
            self.last = int(self.pinned[0])
            self.entries = self.last if self.entries is None or self.last >= self.entries else self.entries - (self.entries - self.last) // 8
            self.event = None

    def tight(self):
        """The last report used more than 80 % of what its call was given (or overflowed it)."""
        return self.last is not None and self.capacity is not None and self.last * 5 > self.capacity * 4

    def report_later(self, ws, offset, capacity=None):
        # the first calls of a shape and calls with tight lists report every time, the others every 8th: an 8-byte copy is
        # still a copy on the stream
        self.calls += 1
        if self.event is not None or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            return
        if not (self.calls <= 4 or self.calls % 8 == 0 or self.tight()):
            return
        if self.pinned is None:
            self.pinned = torch.empty((1,), dtype=torch.int64, pin_memory=True)
        self.pinned.copy_(ws[offset:offset + 8].view(torch.int64), non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        if capacity is not None:
            self.capacity = int(capacity)


def _mesh_workspace(lib, F, N, H, W, bin_size, M, dev, what="meshes", extra=0):
    """(workspace, _Need or None, byte offset of the call's needed-entries word, list entries the workspace was sized for or None);
    what: "meshes" (F faces) or "points"; extra: bytes the entry point takes off the END of the workspace (the marks of the CUDA
    tie order) -- p3d_rasterize_meshes_workspace_bytes counts them in already (include/p3d_amd.h), the short size does not."""
    worst_fn = getattr(lib, f"p3d_rasterize_{what}_workspace_bytes")
    short_fn = getattr(lib, f"p3d_rasterize_{what}_short_workspace_bytes")
    at_fn = getattr(lib, f"p3d_rasterize_{what}_workspace_need_offset")
    worst = worst_fn(F, N, H, W, bin_size, M)
    if SHORT_WORKSPACE == "never" or (SHORT_WORKSPACE != "always" and worst <= SHORT_WORKSPACE_ABOVE):
        WORKSPACE_STATS["last_bytes"], WORKSPACE_STATS["last_entries"] = worst, None
        return _workspace(worst, dev), None, 0, None
    key = (what, dev.index, F, N, H, W, bin_size, M)
    need = _NEEDS.get(key)
    if need is None:
        if len(_NEEDS) >= 64:
            _NEEDS.pop(next(iter(_NEEDS)))
        need = _NEEDS[key] = _Need()
    if not torch.cuda.is_current_stream_capturing():
        need.collect()
    if need.entries is not None:
        entries = need.entries + need.entries // 4 + 4096
    else:
        entries = SHORT_WORKSPACE_FIRST_GUESS if SHORT_WORKSPACE_FIRST_GUESS is not None else 32 * F + (1 << 18)
    nbytes = min(short_fn(F, N, H, W, bin_size, M, int(entries)) + extra, worst)
    WORKSPACE_STATS["short_calls"] += 1
    WORKSPACE_STATS["last_bytes"], WORKSPACE_STATS["last_entries"] = nbytes, int(entries)
    # (the entries go back to the caller, who hands them to report_later: the module-wide stats are another thread's too)
    return _workspace(nbytes, dev), need, at_fn(F, N, H, W, bin_size, M), int(entries)


# ----------------------------------------------------------------------------------------------
# meshes
# ----------------------------------------------------------------------------------------------
# CUDA tie order (include/p3d_amd.h: p3d_rasterize_meshes_cuda_order).  False: the K nearest faces of a pixel under the total
# order (depth, face index), as the reference's CPU and Python implementations return them.  True: where faces tie EXACTLY in
# depth at the K-th place, the survivors the reference's CUDA kernels keep (bit-identical pix_to_face to a CUDA render; the fine
# kernel marks the pixels concerned and a replay kernel rewrites those: profiles/tie_order_timing.py).  rasterize_points obeys it too (there the reference's CUDA kernels also ORDER tied
# entries by array position: rasterize_points.cu:26-28).  Also: P3D_CUDA_TIE_ORDER=1 in the environment.
CUDA_TIE_ORDER = os.environ.get("P3D_CUDA_TIE_ORDER", "0") not in ("", "0")


def _check_face_verts(face_verts):
    if not (face_verts.dim() == 3 and face_verts.size(1) == 3 and face_verts.size(2) == 3):
        raise RuntimeError("face_verts must have dimensions (num_faces, 3, 3)")


def _mesh_outputs(N, H, W, K, device):
    p2f = torch.empty((N, H, W, K), dtype=torch.int64, device=device)
    zbuf = torch.empty((N, H, W, K), dtype=torch.float32, device=device)
    bary = torch.empty((N, H, W, K, 3), dtype=torch.float32, device=device)
    dists = torch.empty((N, H, W, K), dtype=torch.float32, device=device)
    return p2f, zbuf, bary, dists


def rasterize_meshes(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, image_size,
                     blur_radius, faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct,
                     clip_barycentric_coords, cull_backfaces):
    """RasterizeMeshes, rasterize_meshes.h:513-562.  Returns (pix_to_face, zbuf, bary, dists).

    The row cover of the output (include/p3d_amd.h: p3d_rasterize_meshes_with_cover) is written as well and remembered for the
    pix_to_face tensor returned here (_remember_cover): the reference's autograd node hands that very tensor to
    `rasterize_meshes_backward` (renderer/mesh/rasterize_meshes.py:312-357), which then walks only the rows that hold a face --
    the operator signatures stay the reference's."""
    out, cover = _rasterize_meshes_covered(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx,
                                           image_size, blur_radius, faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct,
                                           clip_barycentric_coords, cull_backfaces, want_cover=True)
    if cover is not None:
        _remember_cover(out[0], cover)
    return out


# The row cover of a pix_to_face tensor returned by rasterize_meshes rides ON that tensor object (a Python attribute: it lives
# exactly as long as the object does, and nothing else can ever be mistaken for it -- until round 5 a module-wide map keyed by
# the tensor's address played this part).  The reference's autograd node saves the very object and hands it back to
# `rasterize_meshes_backward` (renderer/mesh/rasterize_meshes.py:291-357: pix_to_face is a non-differentiable output, autograd
# keeps the original).  The cover is used only if the tensor has not been written in place since (version counter) and still has
# the shape the cover was made for; a copy, a view, another tensor: no cover, the backward reads every row.
#
# Writes that bypass the version counter (`pix_to_face.data[...] = x`, an external kernel writing through data_ptr()) are not seen
# by that test: a row the stale cover calls empty would be skipped and the faces written into it would get no gradient.  Three
# ways to be safe: CHECK_COVERS (P3D_CHECK=1) verifies every recalled cover on the device before it is trusted
# (p3d_rasterize_meshes_cover_check: one read of slot 0 of every pixel and a host sync -- about half of what the cover saves) and
# falls back to reading every row, with a warning, when the tensor holds a face the cover does not know of;
# `forget_cover(pix_to_face)` after such a write; RECALL_COVERS = False (P3D_RECALL_COVERS=0) switches the recall off (+0.5 ms on
# the bench batch).  Pinned by tests/test_gpu_cover.py.
RECALL_COVERS = os.environ.get("P3D_RECALL_COVERS", "1") not in ("", "0")
# the forward appends the non-empty words of its cover to a list behind it (p3d_rasterize_meshes_with_cover_list); 0: the plain cover,
# the backward builds the list itself (mesh_backward_areas)
COVER_LIST = os.environ.get("P3D_COVER_LIST", "1") not in ("", "0")
# the backward of perspective + clip launches with K = 4 / 8 reads per-face reciprocals (include/p3d_amd.h: p3d_rasterize_meshes_backward_pre)
FACE_PRE = os.environ.get("P3D_FACE_PRE", "1") not in ("", "0")
CHECK_COVERS = os.environ.get("P3D_CHECK", "0") not in ("", "0")
COVER_RECALLS = [0, 0]  # backward calls without an explicit cover: [found the forward's, found none] (read by tests / profiles)
COVER_CHECKS = [0, 0]   # CHECK_COVERS: [covers verified, of which stale]
_COVER_ATTR = "_p3d_row_cover"


def _remember_cover(p2f, cover):
    setattr(p2f, _COVER_ATTR, (cover, p2f._version))


def forget_cover(pix_to_face):
    """Drop the row cover that rides on a pix_to_face tensor (after writing into it behind autograd's back: see above)."""
    if hasattr(pix_to_face, _COVER_ATTR):
        delattr(pix_to_face, _COVER_ATTR)


def cover_is_current(p2f, cover):
    """CHECK_COVERS: True unless `p2f` holds a face in a 16-pixel row segment that `cover` calls empty (device check + host sync)."""
    N, H, W, K = p2f.shape
    lib = _lib.load()
    with torch.cuda.device(p2f.device):
        stale = torch.empty((1,), dtype=torch.int32, device=p2f.device)
        rc = lib.p3d_rasterize_meshes_cover_check(_ptr(p2f), _ptr(cover), N, H, W, K, _ptr(stale), _stream(p2f.device))
        _lib.check(rc, "rasterize_meshes_cover_check")
        bad = bool(int(stale.item()))
    COVER_CHECKS[0] += 1
    COVER_CHECKS[1] += int(bad)
    return not bad


def checked_cover(p2f, cover):
    """For the autograd nodes that carry their cover themselves (pytorch3d_amd/rasterize_meshes.py): `cover`, or None when CHECK_COVERS
    finds it stale."""
    if cover is None or not CHECK_COVERS or torch.cuda.is_current_stream_capturing() or cover_is_current(p2f, cover):
        return cover
    import warnings

    warnings.warn("pytorch3d_amd: pix_to_face holds faces its forward's row cover does not know of (written through .data or by "
                  "another kernel since the forward?); the backward reads every row instead", RuntimeWarning)
    return None


def _recall_cover(p2f):
    e = getattr(p2f, _COVER_ATTR, None) if RECALL_COVERS and p2f.dtype == torch.int64 and p2f.is_contiguous() and p2f.dim() == 4 else None
    ok = e is not None and p2f._version == e[1] and e[0].device == p2f.device and \
        tuple(e[0].shape) == (p2f.shape[0], (p2f.shape[1] + 15) // 16, (p2f.shape[2] + 15) // 16)
    if ok and CHECK_COVERS and not torch.cuda.is_current_stream_capturing() and not cover_is_current(p2f, e[0]):
        import warnings

        warnings.warn("pytorch3d_amd: pix_to_face holds faces its forward's row cover does not know of (written through .data or by "
                      "another kernel since the forward?); the backward reads every row instead", RuntimeWarning)
        forget_cover(p2f)
        ok = False
    if not ok:
        COVER_RECALLS[1] += 1
        return None
    COVER_RECALLS[0] += 1
    return e[0]


def _rasterize_meshes_covered(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx, image_size,
                              blur_radius, faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct,
                              clip_barycentric_coords, cull_backfaces, want_cover=True):
    """rasterize_meshes + the row cover of its output (include/p3d_amd.h: p3d_rasterize_meshes_with_cover) for the autograd
    nodes of this package: ((pix_to_face, zbuf, bary, dists), cover or None)."""
    dev = _same_device(("face_verts", face_verts), ("mesh_to_face_first_idx", mesh_to_face_first_idx),
                       ("num_faces_per_mesh", num_faces_per_mesh),
                       ("clipped_faces_neighbor_idx", clipped_faces_neighbor_idx))
    _check_face_verts(face_verts)
    if num_faces_per_mesh.size(0) != mesh_to_face_first_idx.size(0):
        raise RuntimeError("num_faces_per_mesh must have save size first dimension as mesh_to_faces_packed_first_idx")
    if clipped_faces_neighbor_idx.size(0) != face_verts.size(0):
        raise RuntimeError("clipped_faces_neighbor_idx must have save size first dimension as face_verts")
    K = int(faces_per_pixel)
    _check_k(K)
    H, W = _hw(image_size)
    bin_size, M = int(bin_size), int(max_faces_per_bin)
    binned = bin_size > 0 and M > 0
    if binned:
        _num_bins(H, W, bin_size, "RasterizeCoarseCuda")
    fv = _c(face_verts, torch.float32)
    first, count = _c(mesh_to_face_first_idx, torch.int64), _c(num_faces_per_mesh, torch.int64)
    nb = _c(clipped_faces_neighbor_idx, torch.int64)
    N, F = count.size(0), fv.size(0)
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = _mesh_outputs(N, H, W, K, dev)
        if out[0].numel() == 0:
            return out, None
        # (the naive launch needs no workspace; with the CUDA tie order it takes one for the lane masks of the marked pixels --
        # without it the replay finds its pixels by reading a pix_to_face entry of every pixel)
        marks = N * ((H + 7) // 8) * ((W + 7) // 8) * 8 + 1024 if CUDA_TIE_ORDER else 0
        ws, need, need_at, entries = _mesh_workspace(lib, F, N, H, W, bin_size, M, dev, extra=marks) if binned else (_workspace(marks, dev), None, 0, None)
        cover = None
        if want_cover and not CUDA_TIE_ORDER and COVER_LIST:
            # the cover with the list of its non-empty words behind it (include/p3d_amd.h: p3d_rasterize_meshes_with_cover_list): what is
            # handed on is the (N, CY, CX) view of the buffer's first words -- a plain cover to everybody; the backward wrappers of this
            # package see the room behind it (cover_has_list) and take the list
            words = N * ((H + 15) // 16) * ((W + 15) // 16)
            buf = torch.empty((lib.p3d_rasterize_meshes_cover_list_bytes(N, H, W) // 4,), dtype=torch.int32, device=dev)
            cover = buf[:words].view(N, (H + 15) // 16, (W + 15) // 16)
        elif want_cover:
            cover = torch.empty((N, (H + 15) // 16, (W + 15) // 16), dtype=torch.int32, device=dev)
        entry = lib.p3d_rasterize_meshes_cuda_order if CUDA_TIE_ORDER else (lib.p3d_rasterize_meshes_with_cover_list if (want_cover and COVER_LIST)
                                                                            else lib.p3d_rasterize_meshes_with_cover)
        rc = entry(
            _ptr(fv), _ptr(first), _ptr(count), _ptr(nb), F, N, H, W, float(blur_radius), K, bin_size if binned else 0,
            M if binned else 0, int(bool(perspective_correct)), int(bool(clip_barycentric_coords)), int(bool(cull_backfaces)),
            _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _ptr(cover) if want_cover else None, _ptr(ws), ws.numel(),
            _stream(dev))
        _lib.check(rc, "rasterize_meshes")
        if need is not None:
            need.report_later(ws, need_at, entries)
    return out, cover


def _rasterize_meshes_naive(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx,
                            image_size, blur_radius, faces_per_pixel, perspective_correct, clip_barycentric_coords,
                            cull_backfaces):
    """RasterizeMeshesNaive, rasterize_meshes.h:108-156."""
    return rasterize_meshes(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, clipped_faces_neighbor_idx,
                            image_size, blur_radius, faces_per_pixel, 0, 0, perspective_correct,
                            clip_barycentric_coords, cull_backfaces)


def _rasterize_meshes_coarse(face_verts, mesh_to_face_first_idx, num_faces_per_mesh, image_size, blur_radius, bin_size,
                             max_faces_per_bin):
    """RasterizeMeshesCoarse, rasterize_meshes.h:292-329.  Returns bin_faces (N,BH,BW,M) int32."""
    dev = _same_device(("face_verts", face_verts), ("mesh_to_face_first_idx", mesh_to_face_first_idx),
                       ("num_faces_per_mesh", num_faces_per_mesh))
    _check_face_verts(face_verts)
    H, W = _hw(image_size)
    bin_size, M = int(bin_size), int(max_faces_per_bin)
    BH, BW = _num_bins(H, W, bin_size, "RasterizeCoarseCuda")
    fv = _c(face_verts, torch.float32)
    first, count = _c(mesh_to_face_first_idx, torch.int64), _c(num_faces_per_mesh, torch.int64)
    N, F = count.size(0), fv.size(0)
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = torch.empty((N, BH, BW, M), dtype=torch.int32, device=dev)
        if out.numel() == 0:
            return out
        ws = _workspace(lib.p3d_rasterize_meshes_workspace_bytes(F, N, H, W, bin_size, M), dev)
        rc = lib.p3d_rasterize_meshes_coarse(_ptr(fv), _ptr(first), _ptr(count), F, N, H, W, float(blur_radius),
                                             bin_size, M, _ptr(out), _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "_rasterize_meshes_coarse")
    return out


def _rasterize_meshes_fine(face_verts, bin_faces, clipped_faces_neighbor_idx, image_size, blur_radius, bin_size,
                           faces_per_pixel, perspective_correct, clip_barycentric_coords, cull_backfaces):
    """RasterizeMeshesFine, rasterize_meshes.h:406-441."""
    dev = _same_device(("face_verts", face_verts), ("bin_faces", bin_faces),
                       ("clipped_faces_neighbor_idx", clipped_faces_neighbor_idx))
    _check_face_verts(face_verts)
    if bin_faces.dim() != 4:
        raise RuntimeError("bin_faces must have 4 dimensions")
    if clipped_faces_neighbor_idx.size(0) != face_verts.size(0):
        raise RuntimeError("clipped_faces_neighbor_idx must have the same first dimension as face_verts")
    K = int(faces_per_pixel)
    if K > kMaxPointsPerPixel:
        raise RuntimeError("Must have num_closest <= 150")
    H, W = _hw(image_size)
    fv = _c(face_verts, torch.float32)
    bf = _c(bin_faces, torch.int32)
    nb = _c(clipped_faces_neighbor_idx, torch.int64)
    N, BH, BW, M = bf.shape
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = _mesh_outputs(N, H, W, K, dev)
        if out[0].numel() == 0:
            return out
        ws = _workspace(lib.p3d_rasterize_fine_workspace_bytes(N, BH, BW, M), dev)
        rc = lib.p3d_rasterize_meshes_fine(_ptr(fv), _ptr(bf), _ptr(nb), fv.size(0), N, BH, BW, M, H, W,
                                           float(blur_radius), int(bin_size), K, int(bool(perspective_correct)),
                                           int(bool(clip_barycentric_coords)), int(bool(cull_backfaces)), _ptr(out[0]),
                                           _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "_rasterize_meshes_fine")
    return out


def cover_ptr(cover, N, H, W):
    """Pointer of a row cover after checking that it is the cover of an (N, H, W, .) rasterization; None -> null."""
    if cover is None:
        return None
    if cover.dtype != torch.int32 or tuple(cover.shape) != (N, (H + 15) // 16, (W + 15) // 16) or not cover.is_contiguous():
        raise RuntimeError("row cover does not belong to this pix_to_face")
    return _ptr(cover)


def cover_has_list(cover, N, H, W):
    """Is `cover` the front of a buffer of p3d_rasterize_meshes_cover_list_bytes, as _rasterize_meshes_covered makes them (the list of
    its non-empty words behind it)?  A clone, a cover from elsewhere, a slice: no."""
    if cover is None or cover.storage_offset() != 0 or not cover.is_contiguous():
        return False
    need = _lib.load().p3d_rasterize_meshes_cover_list_bytes(N, H, W)
    return need > 0 and cover.untyped_storage().nbytes() == need


def backward_workspace(cover, N, H, W, dev):
    """Scratch of the covered backward (the list of areas with work); empty without a cover, or when the cover brings its list."""
    n = _lib.load().p3d_rasterize_meshes_backward_workspace_bytes(N, H, W) if (cover is not None and not cover_has_list(cover, N, H, W)) else 0
    return _workspace(n, dev)


def rasterize_meshes_backward(face_verts, pix_to_face, grad_zbuf, grad_bary, grad_dists, perspective_correct,
                              clip_barycentric_coords, _cover=None):
    """RasterizeMeshesBackward, rasterize_meshes.h:211-252.  Returns grad_face_verts (F,3,3).
    _cover (not part of the reference's signature): the row cover of THIS pix_to_face from _rasterize_meshes_covered."""
    dev = _same_device(("face_verts", face_verts), ("pix_to_face", pix_to_face), ("grad_zbuf", grad_zbuf),
                       ("grad_bary", grad_bary), ("grad_dists", grad_dists))
    # float atomics: the accumulation order is not deterministic (rasterize_meshes.cu:587)
    if torch.are_deterministic_algorithms_enabled() and not torch.is_deterministic_algorithms_warn_only_enabled():
        raise RuntimeError("RasterizeMeshesBackwardCuda does not have a deterministic implementation")
    fv = _c(face_verts, torch.float32)
    if _cover is None:
        _cover = _recall_cover(pix_to_face)  # the forward of this very tensor left its row cover (see rasterize_meshes)
    p2f = _c(pix_to_face, torch.int64)
    gz, gb, gd = _c(grad_zbuf, torch.float32), _c(grad_bary, torch.float32), _c(grad_dists, torch.float32)
    N, H, W, K = p2f.shape
    F = fv.size(0)
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = torch.empty((F, 3, 3), dtype=torch.float32, device=dev)
        if F == 0:
            return out
        if FACE_PRE and perspective_correct and clip_barycentric_coords and K in (4, 8):
            # the per-face reciprocals first (one small launch into a scratch tensor), then the kernel that gathers them per sample
            has_list = cover_has_list(_cover, N, H, W)
            ws = _workspace(0, dev) if has_list else backward_workspace(_cover, N, H, W, dev)
            pre = torch.empty((F, 4), dtype=torch.float32, device=dev)
            rc = lib.p3d_rasterize_meshes_backward_pre(
                _ptr(fv), _ptr(p2f), _ptr(gz), _ptr(gb), _ptr(gd), cover_ptr(_cover, N, H, W), int(has_list), F, N, H, W, K, 1, 1,
                _ptr(out), _ptr(pre), _ptr(ws), ws.numel(), _stream(dev))
        elif cover_has_list(_cover, N, H, W):
            rc = lib.p3d_rasterize_meshes_backward_with_cover_list(
                _ptr(fv), _ptr(p2f), _ptr(gz), _ptr(gb), _ptr(gd), cover_ptr(_cover, N, H, W), F, N, H, W, K,
                int(bool(perspective_correct)), int(bool(clip_barycentric_coords)), _ptr(out), _stream(dev))
        else:
            ws = backward_workspace(_cover, N, H, W, dev)
            rc = lib.p3d_rasterize_meshes_backward_with_cover(
                _ptr(fv), _ptr(p2f), _ptr(gz), _ptr(gb), _ptr(gd), cover_ptr(_cover, N, H, W), F, N, H, W, K,
                int(bool(perspective_correct)), int(bool(clip_barycentric_coords)), _ptr(out), _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "rasterize_meshes_backward")
    return out


# ----------------------------------------------------------------------------------------------
# point clouds
# ----------------------------------------------------------------------------------------------
def _check_points(points):
    if not (points.dim() == 2 and points.size(1) == 3):
        raise RuntimeError("points must have dimensions (num_points, 3)")


def _point_outputs(N, H, W, K, device):
    idx = torch.empty((N, H, W, K), dtype=torch.int32, device=device)
    zbuf = torch.empty((N, H, W, K), dtype=torch.float32, device=device)
    dists = torch.empty((N, H, W, K), dtype=torch.float32, device=device)
    return idx, zbuf, dists


def rasterize_points(points, cloud_to_packed_first_idx, num_points_per_cloud, image_size, radius, points_per_pixel,
                     bin_size, max_points_per_bin):
    """RasterizePoints, rasterize_points.h:343-374.  Returns (idxs int32, zbuf, dists2)."""
    dev = _same_device(("points", points), ("cloud_to_packed_first_idx", cloud_to_packed_first_idx),
                       ("num_points_per_cloud", num_points_per_cloud), ("radius", radius))
    _check_points(points)
    if radius.dim() != 1 or radius.size(0) != points.size(0):
        raise RuntimeError("radius must be of shape (P,)")
    K = int(points_per_pixel)
    _check_k(K)
    H, W = _hw(image_size)
    bin_size, M = int(bin_size), int(max_points_per_bin)
    binned = bin_size > 0 and M > 0
    if binned:
        _num_bins(H, W, bin_size, "RasterizeCoarseCuda")
    pts, rad = _c(points, torch.float32), _c(radius, torch.float32)
    first, count = _c(cloud_to_packed_first_idx, torch.int64), _c(num_points_per_cloud, torch.int64)
    N, P = count.size(0), pts.size(0)
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = _point_outputs(N, H, W, K, dev)
        if out[0].numel() == 0:
            return out
        ws, need, need_at, entries = _mesh_workspace(lib, P, N, H, W, bin_size, M, dev, "points") if binned else (_workspace(0, dev), None, 0, None)
        entry = lib.p3d_rasterize_points_cuda_order if CUDA_TIE_ORDER else lib.p3d_rasterize_points
        rc = entry(_ptr(pts), _ptr(first), _ptr(count), _ptr(rad), P, N, H, W, K, bin_size if binned else 0, M if binned else 0,
                   _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "rasterize_points")
        if need is not None:
            need.report_later(ws, need_at, entries)
    return out


def _rasterize_points_naive(points, cloud_to_packed_first_idx, num_points_per_cloud, image_size, radius,
                            points_per_pixel):
    """RasterizePointsNaive, rasterize_points.h:70-99."""
    return rasterize_points(points, cloud_to_packed_first_idx, num_points_per_cloud, image_size, radius,
                            points_per_pixel, 0, 0)


def _rasterize_points_coarse(points, cloud_to_packed_first_idx, num_points_per_cloud, image_size, radius, bin_size,
                             max_points_per_bin):
    """RasterizePointsCoarse, rasterize_points.h:146-191.  Returns bin_points (N,BH,BW,M) int32."""
    dev = _same_device(("points", points), ("cloud_to_packed_first_idx", cloud_to_packed_first_idx),
                       ("num_points_per_cloud", num_points_per_cloud), ("radius", radius))
    _check_points(points)
    H, W = _hw(image_size)
    bin_size, M = int(bin_size), int(max_points_per_bin)
    BH, BW = _num_bins(H, W, bin_size, "RasterizeCoarseCuda")
    pts, rad = _c(points, torch.float32), _c(radius, torch.float32)
    first, count = _c(cloud_to_packed_first_idx, torch.int64), _c(num_points_per_cloud, torch.int64)
    N, P = count.size(0), pts.size(0)
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = torch.empty((N, BH, BW, M), dtype=torch.int32, device=dev)
        if out.numel() == 0:
            return out
        ws = _workspace(lib.p3d_rasterize_points_workspace_bytes(P, N, H, W, bin_size, M), dev)
        rc = lib.p3d_rasterize_points_coarse(_ptr(pts), _ptr(first), _ptr(count), _ptr(rad), P, N, H, W, bin_size, M,
                                             _ptr(out), _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "_rasterize_points_coarse")
    return out


def _rasterize_points_fine(points, bin_points, image_size, radius, bin_size, points_per_pixel):
    """RasterizePointsFine, rasterize_points.h:222-247."""
    dev = _same_device(("points", points), ("bin_points", bin_points), ("radius", radius))
    _check_points(points)
    K = int(points_per_pixel)
    if K > kMaxPointsPerPixel:
        raise RuntimeError("Must have num_closest <= 150")
    H, W = _hw(image_size)
    pts, rad = _c(points, torch.float32), _c(radius, torch.float32)
    bp = _c(bin_points, torch.int32)
    N, BH, BW, M = bp.shape
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = _point_outputs(N, H, W, K, dev)
        if out[0].numel() == 0:
            return out
        ws = _workspace(lib.p3d_rasterize_fine_workspace_bytes(N, BH, BW, M), dev)
        rc = lib.p3d_rasterize_points_fine(_ptr(pts), _ptr(bp), _ptr(rad), pts.size(0), N, BH, BW, M, H, W,
                                           int(bin_size), K, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(ws),
                                           ws.numel(), _stream(dev))
        _lib.check(rc, "_rasterize_points_fine")
    return out


def rasterize_points_backward(points, idxs, grad_zbuf, grad_dists):
    """RasterizePointsBackward, rasterize_points.h:281-305.  Returns grad_points (P,3)."""
    dev = _same_device(("points", points), ("idxs", idxs), ("grad_zbuf", grad_zbuf), ("grad_dists", grad_dists))
    if torch.are_deterministic_algorithms_enabled() and not torch.is_deterministic_algorithms_warn_only_enabled():
        raise RuntimeError("RasterizePointsBackwardCuda does not have a deterministic implementation")
    pts = _c(points, torch.float32)
    ix = _c(idxs, torch.int32)
    gz, gd = _c(grad_zbuf, torch.float32), _c(grad_dists, torch.float32)
    N, H, W, K = ix.shape
    P = pts.size(0)
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = torch.empty((P, 3), dtype=torch.float32, device=dev)
        if P == 0:
            return out
        rc = lib.p3d_rasterize_points_backward(_ptr(pts), _ptr(ix), _ptr(gz), _ptr(gd), P, N, H, W, K, _ptr(out),
                                               _stream(dev))
        _lib.check(rc, "rasterize_points_backward")
    return out


def inv_r2_of(radius):
    """float32(1) / float32(r * r): `dists / (r * r)` with a Python scalar is a multiplication by this on the device
    (renderer/points/renderer.py:62-64 under torch's div-by-scalar kernel)."""
    import numpy as np

    return float(np.float32(1.0) / np.float32(float(radius) * float(radius)))


_SPLAT_MODES = {"alpha": 0, "norm": 1}  # P3D_COMPOSITE_ALPHA / _NORM_SUM: AlphaCompositor / NormWeightedCompositor


def rasterize_points_composite(points, cloud_to_packed_first_idx, num_points_per_cloud, image_size, radius, features, inv_r2,
                               points_per_pixel, bin_size, max_points_per_bin, mode="alpha"):
    """include/p3d_amd.h: p3d_rasterize_points_composite.  rasterize_points's arguments + features (P, C), C in 1..4, inv_r2
    (inv_r2_of) and the compositor ("alpha" / "norm").  Returns (idxs int32, zbuf, dists2, images (N, H, W, C))."""
    dev = _same_device(("points", points), ("cloud_to_packed_first_idx", cloud_to_packed_first_idx),
                       ("num_points_per_cloud", num_points_per_cloud), ("radius", radius), ("features", features))
    _check_points(points)
    if radius.dim() != 1 or radius.size(0) != points.size(0):
        raise RuntimeError("radius must be of shape (P,)")
    if features.dim() != 2 or features.size(0) != points.size(0) or not 1 <= features.size(1) <= 4:
        raise RuntimeError("features must be of shape (P, C) with C in 1..4")
    K = int(points_per_pixel)
    _check_k(K)
    H, W = _hw(image_size)
    bin_size, M = int(bin_size), int(max_points_per_bin)
    binned = bin_size > 0 and M > 0
    if binned:
        _num_bins(H, W, bin_size, "RasterizeCoarseCuda")
    pts, rad, feats = _c(points, torch.float32), _c(radius, torch.float32), _c(features, torch.float32)
    first, count = _c(cloud_to_packed_first_idx, torch.int64), _c(num_points_per_cloud, torch.int64)
    N, P, C = count.size(0), pts.size(0), feats.size(1)
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = _point_outputs(N, H, W, K, dev)
        images = torch.empty((N, H, W, C), dtype=torch.float32, device=dev)
        if images.numel() == 0:
            return out + (images,)
        ws, need, need_at, entries = _mesh_workspace(lib, P, N, H, W, bin_size, M, dev, "points") if binned else (_workspace(0, dev), None, 0, None)
        rc = lib.p3d_rasterize_points_composite(_SPLAT_MODES[mode], _ptr(pts), _ptr(first), _ptr(count), _ptr(rad), _ptr(feats), P, C, N, H, W, K,
                                                bin_size if binned else 0, M if binned else 0, float(inv_r2), _ptr(out[0]), _ptr(out[1]),
                                                _ptr(out[2]), _ptr(images), _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "rasterize_points_composite")
        if need is not None:
            need.report_later(ws, need_at, entries)
    return out + (images,)


def rasterize_points_composite_backward(points, features, idxs, dists, grad_images, inv_r2, mode="alpha"):
    """include/p3d_amd.h: p3d_rasterize_points_composite_backward.  Returns (grad_points (P, 3), grad_features (P, C))."""
    dev = _same_device(("points", points), ("features", features), ("idxs", idxs), ("dists", dists), ("grad_images", grad_images))
    if torch.are_deterministic_algorithms_enabled() and not torch.is_deterministic_algorithms_warn_only_enabled():
        raise RuntimeError("RasterizePointsBackwardCuda does not have a deterministic implementation")
    pts, feats = _c(points, torch.float32), _c(features, torch.float32)
    ix, ds, gi = _c(idxs, torch.int32), _c(dists, torch.float32), _c(grad_images, torch.float32)
    N, H, W, K = ix.shape
    P, C = pts.size(0), feats.size(1)
    if tuple(gi.shape) != (N, H, W, C):
        raise RuntimeError("grad_images must be of shape (N, H, W, C)")
    lib = _lib.load()
    with torch.cuda.device(dev):
        gp = torch.empty((P, 3), dtype=torch.float32, device=dev)
        gf = torch.empty((P, C), dtype=torch.float32, device=dev)
        if P == 0:
            return gp, gf
        rc = lib.p3d_rasterize_points_composite_backward(_SPLAT_MODES[mode], _ptr(pts), _ptr(feats), _ptr(ix), _ptr(ds), _ptr(gi), P, C, N, H, W, K,
                                                         float(inv_r2), _ptr(gp), _ptr(gf), _stream(dev))
        _lib.check(rc, "rasterize_points_composite_backward")
    return gp, gf


# ----------------------------------------------------------------------------------------------
# compositors
# ----------------------------------------------------------------------------------------------
_ALPHA, _NORM, _SUM = 0, 1, 2


def _strides4(t):
    return (ctypes.c_int64 * 4)(*[int(s) for s in t.stride()])


def _composite_check(features, alphas, points_idx):
    dev = _same_device(("features", features), ("alphas", alphas), ("points_idx", points_idx))
    if features.dim() != 2:
        raise RuntimeError("features must have 2 dimensions (C, P)")
    if alphas.dim() != 4 or points_idx.dim() != 4 or alphas.shape != points_idx.shape:
        raise RuntimeError("alphas and points_idx must both have shape (N, K, H, W)")
    if features.dtype != torch.float32 or alphas.dtype != torch.float32 or points_idx.dtype != torch.int64:
        raise RuntimeError("features/alphas must be float32 and points_idx int64")
    return dev


def _feature_layout(features):
    """(tensor to hand over, its (channel, point) element strides, interleaved?).  Two layouts go to the kernels as they are: the
    contiguous (C, P) tensor of the reference's operators, and the transposed view of a contiguous (P, C) tensor -- what
    PointsRenderer passes (renderer/points/renderer.py:67); there a point's channels are adjacent and the gathers of a pixel cost
    one memory request per entry instead of C (include/p3d_amd.h: p3d_composite_forward_strided).  Anything else is copied."""
    C, P = features.shape
    st = features.stride()
    if C > 1 and P > 1 and st[0] == 1 and st[1] == C:
        return features, (1, C), True
    f = features.contiguous()
    return f, (P, 1), False


def _strides2(st):
    return (ctypes.c_int64 * 2)(int(st[0]), int(st[1]))


def _composite_forward(mode, name, features, alphas, points_idx):
    dev = _composite_check(features, alphas, points_idx)
    feats, fst, _ = _feature_layout(features)
    N, K, H, W = alphas.shape
    C, P = feats.shape
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=dev)
        if out.numel() == 0:
            return out
        rc = lib.p3d_composite_forward_strided(mode, _ptr(feats), _strides2(fst), _ptr(alphas), _ptr(points_idx), N, C, P, K, H, W,
                                               _strides4(alphas), _strides4(points_idx), _ptr(out), _stream(dev))
        _lib.check(rc, name)
    return out


def _composite_backward(mode, name, grad_outputs, features, alphas, points_idx):
    dev = _composite_check(features, alphas, points_idx)
    _need_gpu(grad_outputs, "grad_outputs")
    feats, fst, interleaved = _feature_layout(features)
    go = _c(grad_outputs, torch.float32)
    N, K, H, W = alphas.shape
    C, P = feats.shape
    lib = _lib.load()
    with torch.cuda.device(dev):
        # grad_features in the layout of the features: (C, P) planes, or -- for the renderers' transposed (P, C) view -- the
        # transposed view of a (P, C) tensor (the permute in front of the operator then hands autograd a contiguous gradient)
        if interleaved:
            gf = torch.empty((P, C), dtype=torch.float32, device=dev).t()
        else:
            gf = torch.empty((C, P), dtype=torch.float32, device=dev)
        ga = torch.empty((N, K, H, W), dtype=torch.float32, device=dev)
        rc = lib.p3d_composite_backward_strided(mode, _ptr(go), _ptr(feats), _strides2(fst), _ptr(alphas), _ptr(points_idx), N, C, P, K,
                                                H, W, _strides4(alphas), _strides4(points_idx), _ptr(gf), _strides2(fst), _ptr(ga),
                                                _stream(dev))
        _lib.check(rc, name)
    return gf, ga


def accum_alphacomposite(features, alphas, points_idx):
    """alphaCompositeForward, compositing/alpha_composite.h:59-82."""
    return _composite_forward(_ALPHA, "accum_alphacomposite", features, alphas, points_idx)


def accum_alphacomposite_backward(grad_outputs, features, alphas, points_idx):
    """alphaCompositeBackward, alpha_composite.h:84-115.  Returns (grad_features, grad_alphas)."""
    return _composite_backward(_ALPHA, "accum_alphacomposite_backward", grad_outputs, features, alphas, points_idx)


def accum_weightedsumnorm(features, alphas, points_idx):
    """weightedSumNormForward, compositing/norm_weighted_sum.h:57-80."""
    return _composite_forward(_NORM, "accum_weightedsumnorm", features, alphas, points_idx)


def accum_weightedsumnorm_backward(grad_outputs, features, alphas, points_idx):
    return _composite_backward(_NORM, "accum_weightedsumnorm_backward", grad_outputs, features, alphas, points_idx)


def accum_weightedsum(features, alphas, points_idx):
    """weightedSumForward, compositing/weighted_sum.h:55-78."""
    return _composite_forward(_SUM, "accum_weightedsum", features, alphas, points_idx)


def accum_weightedsum_backward(grad_outputs, features, alphas, points_idx):
    return _composite_backward(_SUM, "accum_weightedsum_backward", grad_outputs, features, alphas, points_idx)


# ----------------------------------------------------------------------------------------------
# interpolate_face_attributes
# ----------------------------------------------------------------------------------------------
_DTYPES = {torch.float32: 0, torch.float64: 1}


def interp_face_attrs_forward(pix_to_face, barycentric_coords, face_attrs):
    """InterpFaceAttrsForward, interp_face_attrs/interp_face_attrs.h:46-66.  Returns pix_attrs (P, D)."""
    dev = _same_device(("pix_to_face", pix_to_face), ("barycentric_coords", barycentric_coords),
                       ("face_attributes", face_attrs))
    if barycentric_coords.dtype != face_attrs.dtype or face_attrs.dtype not in _DTYPES:
        raise RuntimeError("barycentric_coords and face_attributes must have the same floating dtype")
    P = pix_to_face.size(0)
    if barycentric_coords.dim() != 2 or barycentric_coords.size(0) != P or barycentric_coords.size(1) != 3:
        raise RuntimeError("barycentric_coords must have size (P, 3)")
    if face_attrs.dim() != 3 or face_attrs.size(1) != 3:
        raise RuntimeError("face_attrs must have size (F, 3, D)")
    p2f = _c(pix_to_face, torch.int64)
    bary, attrs = barycentric_coords.contiguous(), face_attrs.contiguous()
    F, _, D = attrs.shape
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = torch.empty((P, D), dtype=attrs.dtype, device=dev)
        if out.numel() == 0:
            return out
        rc = lib.p3d_interp_face_attrs_forward(_DTYPES[attrs.dtype], _ptr(p2f), _ptr(bary), _ptr(attrs), P, F, D,
                                               _ptr(out), _stream(dev))
        _lib.check(rc, "interp_face_attrs_forward")
    return out


def interp_face_attrs_backward(pix_to_face, barycentric_coords, face_attrs, grad_pix_attrs, image_shape=None):
    """InterpFaceAttrsBackward, interp_face_attrs.h:95-116.  Returns (grad_bary (P,3), grad_face_attrs (F,3,D)).
    image_shape=(N,H,W,K) (ours, optional): the P samples are image-shaped fragments -> tile-mapped kernel."""
    dev = _same_device(("pix_to_face", pix_to_face), ("barycentric_coords", barycentric_coords),
                       ("face_attributes", face_attrs), ("pix_attrs", grad_pix_attrs))
    if not (barycentric_coords.dtype == face_attrs.dtype == grad_pix_attrs.dtype) or face_attrs.dtype not in _DTYPES:
        raise RuntimeError("barycentric_coords, face_attributes and pix_attrs must have the same floating dtype")
    if torch.are_deterministic_algorithms_enabled() and not torch.is_deterministic_algorithms_warn_only_enabled():
        raise RuntimeError("InterpFaceAttrsBackwardCuda does not have a deterministic implementation")
    P = pix_to_face.size(0)
    if barycentric_coords.dim() != 2 or barycentric_coords.size(0) != P or barycentric_coords.size(1) != 3:
        raise RuntimeError("barycentric_coords must have size (P, 3)")
    if face_attrs.dim() != 3 or face_attrs.size(1) != 3:
        raise RuntimeError("face_attrs must have size (F, 3, D)")
    F, _, D = face_attrs.shape
    if grad_pix_attrs.dim() != 2 or grad_pix_attrs.size(0) != P or grad_pix_attrs.size(1) != D:
        raise RuntimeError("grad_pix_attrs must have size (P, D)")
    p2f = _c(pix_to_face, torch.int64)
    bary, attrs, g = barycentric_coords.contiguous(), face_attrs.contiguous(), grad_pix_attrs.contiguous()
    lib = _lib.load()
    with torch.cuda.device(dev):
        gb = torch.empty((P, 3), dtype=attrs.dtype, device=dev)
        gf = torch.empty((F, 3, D), dtype=attrs.dtype, device=dev)
        if (image_shape is not None and attrs.dtype == torch.float32 and 1 <= D <= 4 and F > 0
                and image_shape[0] * image_shape[1] * image_shape[2] * image_shape[3] == P):
            n_, h_, w_, k_ = (int(x) for x in image_shape)
            rc = lib.p3d_interp_face_attrs_backward_nhwk(_ptr(p2f), _ptr(bary), _ptr(attrs), _ptr(g), n_, h_, w_, k_, F, D,
                                                         _ptr(gb), _ptr(gf), _stream(dev))
            _lib.check(rc, "interp_face_attrs_backward")
            return gb, gf
        rc = lib.p3d_interp_face_attrs_backward(_DTYPES[attrs.dtype], _ptr(p2f), _ptr(bary), _ptr(attrs), _ptr(g), P, F,
                                                D, _ptr(gb), _ptr(gf), _stream(dev))
        _lib.check(rc, "interp_face_attrs_backward")
    return gb, gf


# ----------------------------------------------------------------------------------------------
# blending
# ----------------------------------------------------------------------------------------------
def sigmoid_alpha_blend(distances, pix_to_face, sigma):
    """SigmoidAlphaBlend, blending/sigmoid_alpha_blend.h:73-84.  distances, pix_to_face (N,H,W,K) -> alphas (N,H,W)."""
    dev = _same_device(("distances", distances), ("pix_to_face", pix_to_face))
    if distances.dim() != 4 or pix_to_face.shape != distances.shape:
        raise RuntimeError("distances and pix_to_face must both have shape (N, H, W, K)")
    d, p2f = _c(distances, torch.float32), _c(pix_to_face, torch.int64)
    N, H, W, K = d.shape
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = torch.empty((N, H, W), dtype=torch.float32, device=dev)
        if out.numel() == 0:
            return out
        rc = lib.p3d_sigmoid_alpha_blend_forward(_ptr(d), _ptr(p2f), float(sigma), N * H * W, K, _ptr(out), _stream(dev))
        _lib.check(rc, "sigmoid_alpha_blend")
    return out


def sigmoid_alpha_blend_backward(grad_alphas, alphas, distances, pix_to_face, sigma):
    """SigmoidAlphaBlendBackward, sigmoid_alpha_blend.h:86-103.  Returns grad_distances (N,H,W,K)."""
    dev = _same_device(("distances", distances), ("pix_to_face", pix_to_face), ("alphas", alphas),
                       ("grad_alphas", grad_alphas))
    d, p2f = _c(distances, torch.float32), _c(pix_to_face, torch.int64)
    ga, al = _c(grad_alphas, torch.float32), _c(alphas, torch.float32)
    N, H, W, K = d.shape
    lib = _lib.load()
    with torch.cuda.device(dev):
        out = torch.empty((N, H, W, K), dtype=torch.float32, device=dev)
        if out.numel() == 0:
            return out
        rc = lib.p3d_sigmoid_alpha_blend_backward(_ptr(ga), _ptr(al), _ptr(d), _ptr(p2f), float(sigma), N * H * W, K,
                                                  _ptr(out), _stream(dev))
        _lib.check(rc, "sigmoid_alpha_blend_backward")
    return out


HOT_PATH_EXPORTS = (
    "rasterize_meshes", "rasterize_meshes_backward", "_rasterize_meshes_naive", "_rasterize_meshes_coarse",
    "_rasterize_meshes_fine", "rasterize_points", "rasterize_points_backward", "_rasterize_points_naive",
    "_rasterize_points_coarse", "_rasterize_points_fine", "accum_alphacomposite", "accum_alphacomposite_backward",
    "accum_weightedsumnorm", "accum_weightedsumnorm_backward", "accum_weightedsum", "accum_weightedsum_backward",
    "interp_face_attrs_forward", "interp_face_attrs_backward", "sigmoid_alpha_blend", "sigmoid_alpha_blend_backward",
)
