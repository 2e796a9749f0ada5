"""Minimal packed-batch containers with the accessor names the reference's L2 functions use.

The reference's `Meshes` / `Pointclouds` (pytorch3d/structures/meshes.py:960-1035,
pointclouds.py:487-600) are OUT OF SCOPE and used unmodified when the reference is installed;
these two classes only give bench.py, the tests and the sharding driver something with the same
accessors (`verts_packed()`, `faces_packed()`, `mesh_to_faces_packed_first_idx()`, ...) on a box
where the reference is absent.  Any object exposing those accessors works with
pytorch3d_amd.rasterize_meshes / rasterize_points.
"""
from typing import List, Sequence

import torch


class PackedMeshes:
    """A heterogeneous batch of triangle meshes in packed layout."""

    def __init__(self, verts: Sequence[torch.Tensor], faces: Sequence[torch.Tensor]):
        if len(verts) != len(faces):
            raise ValueError("verts and faces must have the same length")
        self._N = len(verts)
        self.device = verts[0].device if self._N else torch.device("cpu")
        nv = torch.tensor([v.shape[0] for v in verts], dtype=torch.int64)
        nf = torch.tensor([f.shape[0] for f in faces], dtype=torch.int64)
        v_first = torch.cumsum(nv, 0) - nv
        f_first = torch.cumsum(nf, 0) - nf
        self._verts_list = list(verts)
        self._faces_list = list(faces)
        self._nv_host = [int(x) for x in nv.tolist()]  # host copies: no device sync when re-splitting
        self._verts_packed = torch.cat(list(verts), 0) if self._N else torch.zeros((0, 3))
        self._faces_packed = (torch.cat([f + int(o) for f, o in zip(faces, v_first)], 0)
                              if self._N else torch.zeros((0, 3), dtype=torch.int64))
        self._num_verts = nv.to(self.device)
        self._num_faces = nf.to(self.device)
        self._v_first = v_first.to(self.device)
        self._f_first = f_first.to(self.device)
        self._F = int(nf.max()) if self._N else 0  # max faces per mesh, as Meshes._F
        self._V = int(nv.max()) if self._N else 0

    def __len__(self):
        return self._N

    def verts_packed(self):
        return self._verts_packed

    def faces_packed(self):
        return self._faces_packed

    def verts_list(self) -> List[torch.Tensor]:
        if self._verts_list is None:
            self._verts_list = list(torch.split(self._verts_packed, self._nv_host, 0))
        return self._verts_list

    def faces_list(self) -> List[torch.Tensor]:
        return self._faces_list

    def mesh_to_faces_packed_first_idx(self):
        return self._f_first

    def mesh_to_verts_packed_first_idx(self):
        return self._v_first

    def num_faces_per_mesh(self):
        return self._num_faces

    def num_verts_per_mesh(self):
        return self._num_verts

    def verts_normals_packed(self):
        """Area-weighted vertex normals as Meshes._compute_vertex_normals (structures/meshes.py:884-927): plain
        torch ops, differentiable; computed on every call (this container caches nothing that depends on verts)."""
        verts, faces = self._verts_packed, self._faces_packed
        fv = verts[faces]
        fn = torch.cross(fv[:, 2] - fv[:, 1], fv[:, 0] - fv[:, 1], dim=1)
        vn = torch.zeros_like(verts)
        for c in range(3):
            vn = vn.index_add(0, faces[:, c], fn)
        return torch.nn.functional.normalize(vn, eps=1e-6, dim=1)

    def faces_normals_packed(self):
        """Unit face normals cross(v1 - v0, v2 - v0) / max(|.|, 1e-6) as mesh_face_areas_normals
        (csrc/face_areas_normals/face_areas_normals_cpu.cpp:44-62), with torch ops."""
        fv = self._verts_packed[self._faces_packed]
        c = torch.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=1)
        return c / c.norm(dim=1, keepdim=True).clamp_min(1e-6)

    def verts_packed_to_mesh_idx(self):
        return torch.repeat_interleave(torch.arange(self._N, device=self.device), self._num_verts)

    def update_verts_packed(self, new_verts_packed):
        """Same topology, new vertex positions (like Meshes.update_padded, for camera transforms)."""
        out = object.__new__(PackedMeshes)
        out.__dict__.update(self.__dict__)
        out._verts_packed = new_verts_packed
        out._verts_list = None  # split lazily in verts_list()
        return out

    def slice(self, start: int, stop: int) -> "PackedMeshes":
        """Meshes start..stop-1 as their own batch (the unit of multi-GPU sharding)."""
        return PackedMeshes(self.verts_list()[start:stop], self._faces_list[start:stop])


class PackedPointclouds:
    """A heterogeneous batch of point clouds in packed layout."""

    def __init__(self, points: Sequence[torch.Tensor]):
        self._N = len(points)
        self.device = points[0].device if self._N else torch.device("cpu")
        n = torch.tensor([p.shape[0] for p in points], dtype=torch.int64)
        self._points_list = list(points)
        self._points_packed = torch.cat(list(points), 0) if self._N else torch.zeros((0, 3))
        self._num_points = n.to(self.device)
        self._first = (torch.cumsum(n, 0) - n).to(self.device)
        self._P = int(n.max()) if self._N else 0  # max points per cloud, as Pointclouds._P

    def __len__(self):
        return self._N

    def points_packed(self):
        return self._points_packed

    def points_list(self):
        return self._points_list

    def cloud_to_packed_first_idx(self):
        return self._first

    def num_points_per_cloud(self):
        return self._num_points

    def padded_to_packed_idx(self):
        idx = [torch.arange(int(c), device=self.device) + i * self._P for i, c in enumerate(self._num_points.tolist())]
        return torch.cat(idx, 0) if idx else torch.zeros((0,), dtype=torch.int64, device=self.device)
