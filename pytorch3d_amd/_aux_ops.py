"""Four small `pytorch3d._C` operators ABOVE the rasterization boundary, as plain torch formulations.

Not part of the hot path and not HIP kernels: `Meshes.faces_normals_packed()` / `faces_areas_packed()`
(pytorch3d/structures/meshes.py:868-880 -> ops/mesh_face_areas_normals.py:48,63) and `packed_to_padded` /
`padded_to_packed` (ops/packed_to_padded.py:52-62,142-152) are what the reference's mesh classes and flat shading call on
the way to the renderer.  They are F- / V-sized and run once per mesh batch; providing them lets the UNMODIFIED
`Meshes` + `MeshRenderer(MeshRasterizer, HardFlatShader)` work through the shim without the reference's own extension.
Semantics follow the reference's kernels (csrc/face_areas_normals/face_areas_normals.cu:14-70: area = |(v1 - v0) x (v2 -
v0)| / 2, normal = cross / max(|cross|, 1e-6); csrc/packed_to_padded_tensor/*: zero padding, rows first_idxs[n] ..).
"""
import torch


def _cross(verts, faces):
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    return torch.cross(v1 - v0, v2 - v0, dim=1)


def face_areas_normals_forward(verts, faces):
    c = _cross(verts, faces)
    norm = c.norm(dim=1)
    return norm / 2.0, c / norm.clamp_min(1e-6)[:, None]


def face_areas_normals_backward(grad_areas, grad_normals, verts, faces):
    """face_areas_normals.cu:64-216.  With c = (v1 - v0) x (v2 - v0), t = dc / d(vertex coordinate) and s = t . c:

        grad = grad_area s / (2 |c|) + sum_j grad_normal_j (t_j - c_j s / |c|^2) / |c|        (|c| clamped at 1e-6)

    -- the derivative of area and unit normal -- with ONE deviation kept from the reference: in d/d(v1.z) the j = y term
    multiplies by c_x where the derivative has c_y (face_areas_normals.cu:183-184, the same in its CPU kernel).  A
    drop-in has to return what the reference returns."""
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    a, b = v1 - v0, v2 - v0
    c = torch.cross(a, b, dim=1)
    norm = c.norm(dim=1).clamp_min(1e-6)
    eye = torch.eye(3, dtype=verts.dtype, device=verts.device)
    # t1[f, d, :] = e_d x b,  t2[f, d, :] = a x e_d,  t0 = -(t1 + t2)
    t1 = torch.cross(eye[None].expand(a.shape[0], 3, 3), b[:, None, :].expand(-1, 3, 3), dim=2)
    t2 = torch.cross(a[:, None, :].expand(-1, 3, 3), eye[None].expand(a.shape[0], 3, 3), dim=2)
    t = torch.stack([-(t1 + t2), t1, t2], 1)  # (F, vertex, d, j)
    sdot = (t * c[:, None, None, :]).sum(-1)  # (F, vertex, d)
    cj = c[:, None, None, :].expand(-1, 3, 3, 3).clone()
    cj[:, 1, 2, 1] = c[:, 0]  # the reference's c_x in place of c_y
    inv = 1.0 / norm
    dn = (t - cj * (sdot * (inv * inv)[:, None, None])[..., None]) * inv[:, None, None, None]
    g = grad_areas[:, None, None] * sdot * (0.5 * inv)[:, None, None] + (dn * grad_normals[:, None, None, :]).sum(-1)
    out = torch.zeros_like(verts)
    out.index_add_(0, faces.reshape(-1), g.reshape(-1, 3))
    return out


def packed_to_padded(inputs_packed, first_idxs, max_size):
    """(F, D), (N,) -> (N, max_size, D), zero padded."""
    F, D = inputs_packed.shape
    N = first_idxs.shape[0]
    out = torch.zeros((N, int(max_size), D), dtype=inputs_packed.dtype, device=inputs_packed.device)
    if F == 0 or N == 0:
        return out
    ends = torch.cat([first_idxs[1:], first_idxs.new_tensor([F])])
    row = torch.arange(F, device=inputs_packed.device)
    n = torch.searchsorted(first_idxs.contiguous(), row, right=True) - 1
    j = row - first_idxs[n]
    keep = (j < int(max_size)) & (row < ends[n])
    out[n[keep], j[keep]] = inputs_packed[keep]
    return out


def padded_to_packed(inputs_padded, first_idxs, num_inputs):
    """(N, max_size, D), (N,) -> (num_inputs, D)."""
    N, M, D = inputs_padded.shape
    F = int(num_inputs)
    out = torch.zeros((F, D), dtype=inputs_padded.dtype, device=inputs_padded.device)
    if F == 0 or N == 0:
        return out
    row = torch.arange(F, device=inputs_padded.device)
    n = torch.searchsorted(first_idxs.contiguous(), row, right=True) - 1
    j = row - first_idxs[n]
    keep = j < M
    out[row[keep]] = inputs_padded[n[keep], j[keep]]
    return out
