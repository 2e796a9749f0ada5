"""Install pytorch3d_amd._C as `pytorch3d._C`, so the UNMODIFIED reference Python package
(pytorch3d.renderer.MeshRasterizer, PointsRasterizer, AlphaCompositor, NormWeightedCompositor,
interpolate_face_attributes, ...) runs on the MI355X kernels.

    import pytorch3d_amd.shim as shim
    shim.install()                       # reference already importable, or
    shim.install("/path/to/pytorch3d")   # a source checkout (no build needed: only _C is native)
    from pytorch3d.renderer import MeshRasterizer   # drop-in

`pytorch3d/__init__.py` does not import `_C`; the sub-packages do (`from pytorch3d import _C`),
so the module must be in sys.modules before they are imported.  Operators outside the hot path
(knn, point_mesh, pulsar, ...) raise NotImplementedError when called.

    shim.install(patch_python=True)

additionally replaces the reference's pure-torch callers and callees of the operator surface with the
fused HIP versions of this package, so that `MeshRenderer(MeshRasterizer, SoftPhongShader)` built from the UNMODIFIED
reference classes runs on them end to end:

    renderer.mesh.rasterizer.MeshRasterizer.forward           -> camera transform on the PACKED vertices in one launch
                                                                 (csrc/transform.hip) + the fused rasterize_meshes below
    renderer.mesh.rasterize_meshes.rasterize_meshes           -> fused gather + rasterizer (+ HIP clipping), one autograd node
    renderer.mesh.clip.clip_faces / convert_clipped_...       -> csrc/clip.hip
    renderer.mesh.shader.SoftPhongShader.forward              -> csrc/soft_phong.hip: shading + softmax blend in one kernel
                                                                 each way (K in 1, 2, 4, 8, 16), TexturesVertex fused in
    renderer.blending.softmax_rgb_blend / hard_rgb_blend      -> csrc/blend.hip
    renderer.mesh.shading.phong_shading / flat_shading / gouraud_shading -> csrc/shade.hip (+ interp.hip)
    renderer.mesh.textures.TexturesUV / TexturesAtlas .sample_textures   -> csrc/texture*.hip, atlas.hip
    renderer.mesh.shader.Hard{Phong,Gouraud,Flat}Shader.forward -> the shader sees the nearest slot only (hard_rgb_blend keeps
                                                                 nothing else): 1 / K of the shading and texture work
    renderer.mesh.shader.SoftSilhouetteShader.forward         -> no ones_like(bary_coords) (1.6 GB at the bench batch)
    structures.meshes.Meshes.offset_verts / offset_verts_     -> one add on the packed vertices, topology shared, no host sync
                                                                 (the reference re-runs Meshes.__init__: ~70 syncs per call)

Every replacement falls back to the reference's own function for inputs the fused kernels do not cover (CPU tensors,
colour widths other than 3, light classes other than Point / Directional / Ambient, padding modes grid_sample has and
the kernels do not).  The names are replaced in every loaded `pytorch3d.*` module that imported them (`from .x import
f` copies), `uninstall_python_patches()` restores them.
"""
import sys
import types

from . import _C as _ours


def make_module(flavour="ctypes"):
    """flavour: "ctypes" -- the operators of pytorch3d_amd/_C.py (ctypes over libp3d_amd.so; row-cover recall, short workspaces,
    CUDA tie order; no compiler needed) -- or "pybind": the same operator set compiled as a torch extension over the same C ABI
    (pytorch3d_amd/csrc/bind.cpp, built by pytorch3d_amd/build_bind.py: INTEGRATION.md section B, the reference's allocation
    pattern and nothing besides)."""
    if flavour not in ("ctypes", "pybind"):
        raise ValueError("flavour must be 'ctypes' or 'pybind'")
    mod = types.ModuleType("pytorch3d._C")
    mod.__doc__ = "pytorch3d._C provided by pytorch3d_amd (MI355X rasterization hot path), %s flavour" % flavour
    mod.__p3d_amd_flavour__ = flavour
    src = _ours
    if flavour == "pybind":
        from . import build_bind

        src = build_bind.load()
    for name in _ours.HOT_PATH_EXPORTS:
        setattr(mod, name, getattr(src, name))
    for name in ("EPS", "MAX_FLOAT", "MAX_INT", "MAX_UINT", "MAX_USHORT", "PULSAR_MAX_GRAD_SPHERES"):
        setattr(mod, name, getattr(_ours, name))
    # four small operators the reference's mesh classes call on the way to the renderer (face normals / areas, packed <->
    # padded): torch formulations, not part of the hot path (pytorch3d_amd/_aux_ops.py)
    from . import _aux_ops

    for name in ("face_areas_normals_forward", "face_areas_normals_backward", "packed_to_padded", "padded_to_packed"):
        setattr(mod, name, getattr(_aux_ops, name))

    def __getattr__(name):  # PEP 562: anything else is outside the hot path
        if name.startswith("__"):
            raise AttributeError(name)

        def _missing(*args, **kwargs):
            raise NotImplementedError(f"pytorch3d._C.{name} is outside the rasterization hot path that "
                                      "pytorch3d_amd implements")

        return _missing

    mod.__getattr__ = __getattr__
    return mod


def install(reference_root=None, patch_python=False, flavour="ctypes"):
    """Register the shim (idempotent).  Returns the module object.  patch_python: see the module docstring; flavour: make_module."""
    if reference_root is not None and reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    existing = sys.modules.get("pytorch3d._C")
    if existing is not None and getattr(existing, "__p3d_amd__", False) and getattr(existing, "__p3d_amd_flavour__", "ctypes") == flavour:
        mod = existing
    else:
        mod = make_module(flavour)
        mod.__p3d_amd__ = True
        sys.modules["pytorch3d._C"] = mod
        pkg = sys.modules.get("pytorch3d")
        if pkg is not None:
            setattr(pkg, "_C", mod)
    if patch_python:
        patch_reference_python()
    return mod


# ---------------------------------------------------------------------------------------------------------------------
# patch_python: the reference's torch formulations around the operator surface -> the fused kernels
# ---------------------------------------------------------------------------------------------------------------------
# patched MeshRasterizer.forward: rasterize un-clipped first and look for vertices behind the near plane afterwards (see there)
SPECULATE_NO_CLIPPING = True

_PATCHED = []  # (owner object, attribute name, original, replacement)
PATCH_CALLS = {}  # name -> [fused calls, fallback calls]: what actually ran (read by tests/run_reference_suite.py)


def _count(name, fused):
    c = PATCH_CALLS.setdefault(name, [0, 0])
    c[0 if fused else 1] += 1


def _replace_everywhere(orig, new):
    """Rebind every `pytorch3d.*` module attribute that IS `orig` (the defining module and all `from x import f` copies)."""
    for mname, mod in list(sys.modules.items()):
        if mod is None or not (mname == "pytorch3d" or mname.startswith("pytorch3d.")):
            continue
        for attr, val in list(vars(mod).items()):
            if val is orig:
                setattr(mod, attr, new)
                _PATCHED.append((mod, attr, orig, new))


def _is_hip_f32(t):
    import torch

    return torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32


def _fragments_ok(fragments):
    import torch

    p2f = getattr(fragments, "pix_to_face", None)
    return (torch.is_tensor(p2f) and p2f.is_cuda and p2f.dtype == torch.int64 and p2f.dim() == 4
            and _is_hip_f32(getattr(fragments, "bary_coords", None)))


_FUSED_LIGHTS = ("PointLights", "DirectionalLights", "AmbientLights")


def _shading_ok(meshes, fragments, lights, cameras, materials, colors=None):
    if not _fragments_ok(fragments) or type(lights).__name__ not in _FUSED_LIGHTS:
        return False
    if colors is not None and not (_is_hip_f32(colors) and colors.shape[-1] == 3 and colors.dim() == 5):
        return False
    if not all(hasattr(materials, a) for a in ("ambient_color", "diffuse_color", "specular_color", "shininess")):
        return False
    if not hasattr(cameras, "get_camera_center"):
        return False
    v = meshes.verts_packed()
    return _is_hip_f32(v) and v.device == fragments.pix_to_face.device


def patch_reference_python():
    """Idempotent.  The reference package must be importable (install(reference_root) or an installed pytorch3d)."""
    if _PATCHED:
        return
    import importlib

    # (importlib: the package re-exports FUNCTIONS named rasterize_meshes etc. over the sub-modules of the same name)
    our_blend = importlib.import_module(__package__ + ".blending")
    our_clip = importlib.import_module(__package__ + ".clip")
    our_rm = importlib.import_module(__package__ + ".rasterize_meshes")
    our_shade = importlib.import_module(__package__ + ".shading")
    our_tex = importlib.import_module(__package__ + ".textures")

    rm = importlib.import_module("pytorch3d.renderer.mesh.rasterize_meshes")
    clip = importlib.import_module("pytorch3d.renderer.mesh.clip")
    blend = importlib.import_module("pytorch3d.renderer.blending")
    shading = importlib.import_module("pytorch3d.renderer.mesh.shading")
    textures = importlib.import_module("pytorch3d.renderer.mesh.textures")
    importlib.import_module("pytorch3d.renderer")  # every module that copied the names must be loaded before rebinding

    def wrap(name, orig, ours, usable):
        def patched(*args, **kwargs):
            ok = False
            try:
                ok = bool(usable(*args, **kwargs))
            except Exception:  # an argument form the probe does not understand: the reference's own code decides
                ok = False
            _count(name, ok)
            return ours(*args, **kwargs) if ok else orig(*args, **kwargs)

        patched.__name__ = getattr(orig, "__name__", name)
        patched.__doc__ = getattr(orig, "__doc__", None)
        patched.__wrapped__ = orig
        patched.__p3d_amd__ = True
        return patched

    # rasterize_meshes(meshes, image_size, blur_radius, faces_per_pixel, bin_size, max_faces_per_bin, perspective_correct,
    #                  clip_barycentric_coords, cull_backfaces, z_clip_value, cull_to_frustum): rasterize_meshes.py:32-250
    def rm_ok(meshes, *a, **k):
        return _is_hip_f32(meshes.verts_packed()) and meshes.faces_packed().is_cuda

    _replace_everywhere(rm.rasterize_meshes, wrap("rasterize_meshes", rm.rasterize_meshes, our_rm.rasterize_meshes, rm_ok))

    # clip.py:324-734
    def clip_ok(face_verts_unclipped, *a, **k):
        return _is_hip_f32(face_verts_unclipped)

    _replace_everywhere(clip.clip_faces, wrap("clip_faces", clip.clip_faces, our_clip.clip_faces, clip_ok))

    def conv_ok(pix_to_face_clipped, bary_coords_clipped, clipped_faces):
        return pix_to_face_clipped.is_cuda and _is_hip_f32(bary_coords_clipped)

    _replace_everywhere(clip.convert_clipped_rasterization_to_original_faces,
                        wrap("convert_clipped_rasterization_to_original_faces", clip.convert_clipped_rasterization_to_original_faces,
                             our_clip.convert_clipped_rasterization_to_original_faces, conv_ok))

    # blending.py:54-88, 147-244
    def blend_ok(colors, fragments, blend_params, **k):
        import torch

        p2f = fragments.pix_to_face
        return (_is_hip_f32(colors) and colors.dim() == 5 and colors.shape[-1] == 3 and p2f.is_cuda
                and p2f.dtype == torch.int64 and tuple(colors.shape[:4]) == tuple(p2f.shape) and p2f.shape[3] >= 1)

    def soft_ok(colors, fragments, blend_params, znear=1.0, zfar=100):
        return blend_ok(colors, fragments, blend_params) and _is_hip_f32(fragments.dists) and _is_hip_f32(fragments.zbuf)

    _replace_everywhere(blend.softmax_rgb_blend, wrap("softmax_rgb_blend", blend.softmax_rgb_blend, our_blend.softmax_rgb_blend, soft_ok))
    _replace_everywhere(blend.hard_rgb_blend, wrap("hard_rgb_blend", blend.hard_rgb_blend, our_blend.hard_rgb_blend, blend_ok))

    # shading.py:100-225
    def phong_ok(meshes, fragments, lights, cameras, materials, texels):
        return _shading_ok(meshes, fragments, lights, cameras, materials, texels)

    def gouraud_ok(meshes, fragments, lights, cameras, materials):
        tex = getattr(meshes, "textures", None)
        return (type(tex).__name__ == "TexturesVertex" and _shading_ok(meshes, fragments, lights, cameras, materials)
                and tex.verts_features_packed().shape[-1] == 3)

    _replace_everywhere(shading.phong_shading, wrap("phong_shading", shading.phong_shading, our_shade.phong_shading, phong_ok))
    _replace_everywhere(shading.flat_shading, wrap("flat_shading", shading.flat_shading, our_shade.flat_shading, phong_ok))
    _replace_everywhere(shading.gouraud_shading, wrap("gouraud_shading", shading.gouraud_shading, our_shade.gouraud_shading, gouraud_ok))

    # TexturesUV.sample_textures (textures.py:1190-1313), TexturesAtlas.sample_textures (textures.py:565-612): methods
    import torch

    uv_orig = textures.TexturesUV.sample_textures

    def uv_sample(self, fragments, **kwargs):
        ok = (_fragments_ok(fragments) and not self.isempty() and self.padding_mode in our_tex._PAD
              and self.sampling_mode in our_tex._MODE and _is_hip_f32(self.maps_padded())
              and self.maps_padded().device == fragments.pix_to_face.device)
        ids = self.maps_ids_padded() if ok and hasattr(self, "maps_ids_padded") else None
        if ok and ids is not None and self.maps_padded().shape[1] < 2:
            ok = False
        _count("TexturesUV.sample_textures", ok)
        if not ok:
            return uv_orig(self, fragments, **kwargs)
        faces_verts_uvs = torch.cat([i[j] for i, j in zip(self.verts_uvs_list(), self.faces_uvs_list())])  # textures.py:1218-1221
        return our_tex.sample_textures_uv(fragments, faces_verts_uvs.to(torch.float32), self.maps_padded(), align_corners=self.align_corners,
                                          padding_mode=self.padding_mode, sampling_mode=self.sampling_mode, maps_ids=ids)

    uv_sample.__wrapped__ = uv_orig
    textures.TexturesUV.sample_textures = uv_sample
    _PATCHED.append((textures.TexturesUV, "sample_textures", uv_orig, uv_sample))

    atlas_orig = textures.TexturesAtlas.sample_textures

    def atlas_sample(self, fragments, **kwargs):
        ok = _fragments_ok(fragments) and not self.isempty()
        at = self.atlas_packed() if ok else None
        ok = ok and _is_hip_f32(at) and at.device == fragments.pix_to_face.device and at.shape[1] >= 1 and at.shape[3] >= 1
        _count("TexturesAtlas.sample_textures", ok)
        if not ok:
            return atlas_orig(self, fragments, **kwargs)
        return our_tex.sample_textures_atlas(fragments, at)

    atlas_sample.__wrapped__ = atlas_orig
    textures.TexturesAtlas.sample_textures = atlas_sample
    _PATCHED.append((textures.TexturesAtlas, "sample_textures", atlas_orig, atlas_sample))

    # compositing.py:68-96, 148-175, 227-247: one autograd node for the three modes; no .clone() of features / alphas / indices
    # (the kernels never write them) and the renderer's permuted views go to the C ABI with their strides
    our_comp = importlib.import_module(__package__ + ".compositing")
    comp = importlib.import_module("pytorch3d.renderer.compositing")

    def comp_ok(pointsidx, alphas, pt_clds):
        return (_is_hip_f32(alphas) and _is_hip_f32(pt_clds) and pointsidx.is_cuda and pointsidx.dtype == torch.int64
                and pointsidx.dim() == 4 and tuple(pointsidx.shape) == tuple(alphas.shape) and pt_clds.dim() == 2)

    for fname in ("alpha_composite", "norm_weighted_sum", "weighted_sum"):
        _replace_everywhere(getattr(comp, fname), wrap(fname, getattr(comp, fname), getattr(our_comp, fname), comp_ok))

    _patch_mesh_rasterizer(our_rm)
    _patch_points_rasterizer(our_rm)
    _patch_soft_phong_shader(our_shade)
    _patch_meshes_offset_verts()
    _patch_hard_and_silhouette_shaders()


def camera_matrices(cameras, kwargs):
    """(world -> view, view -> NDC, is_perspective, min znear or None) of `cameras`, or None when they have no matrix form.
    Building them is ~25 small torch launches plus a host sync for znear (`.min().item()`): cached on the camera object
    under a fingerprint of its parameters -- (name, object, in-place version counter) of every tensor attribute, the value of
    every plain one -- whenever the call overrides none of them (kwargs holds nothing but the two objects themselves).
    A new tensor assigned to an attribute, an in-place edit of one, or a different scalar all change the fingerprint."""
    import importlib

    import torch

    cam_utils = importlib.import_module("pytorch3d.renderer.cameras")
    cacheable = all(k in ("cameras", "raster_settings") for k in kwargs)
    fp = None
    if cacheable:
        # every tensor the camera holds: plain attributes AND the nn.Module registries (`_parameters`, `_buffers`: a camera whose
        # R / T were registered as nn.Parameter or buffer keeps them there, not in vars()); a camera that holds a Parameter or a
        # tensor that requires grad is being optimised -- its matrices are rebuilt on every call
        items = [(k, v) for k, v in vars(cameras).items() if not k.startswith("_p3d_amd") and k not in ("_parameters", "_buffers")]
        for reg in ("_parameters", "_buffers"):
            items += [(reg + "." + k, v) for k, v in (vars(cameras).get(reg) or {}).items()]
        if any(torch.is_tensor(v) and (isinstance(v, torch.nn.Parameter) or v.requires_grad) for _, v in items):
            cacheable = False
    if cacheable:
        # (the fingerprint holds the tensor OBJECTS, compared by identity: an id() alone could be reused by a new tensor.  Edits
        # that bypass the version counter -- `cameras.T.data.add_(...)`, an external kernel writing through data_ptr() -- are
        # not seen: INTEGRATION.md lists them as unsupported with the cache; `del cameras._p3d_amd_matrices` drops it)
        fp = tuple((k, v, v._version) if torch.is_tensor(v) else (k, v, None) for k, v in items
                   if torch.is_tensor(v) or isinstance(v, (bool, int, float, str, tuple, type(None))))
        hit = cameras.__dict__.get("_p3d_amd_matrices")
        if hit is not None and len(hit[0]) == len(fp) and all(
                a[0] == b[0] and a[2] == b[2] and (a[1] is b[1] if torch.is_tensor(a[1]) or torch.is_tensor(b[1]) else a[1] == b[1])
                for a, b in zip(hit[0], fp)):
            return hit[1]
    w2v = cameras.get_world_to_view_transform(**kwargs).get_matrix()
    proj = cam_utils.try_get_projection_transform(cameras, kwargs)
    out = None
    if proj is not None:
        v2n = proj.compose(cameras.get_ndc_camera_transform(**kwargs)).get_matrix()
        znear = cameras.get_znear()
        if isinstance(znear, torch.Tensor):
            znear = znear.min().item()
        out = (w2v, v2n, bool(cameras.is_perspective()), znear)
    if cacheable and not (out is not None and (out[0].requires_grad or out[1].requires_grad)):
        cameras.__dict__["_p3d_amd_matrices"] = (fp, out)
    return out



def _patch_mesh_rasterizer(our_rm):
    """MeshRasterizer.forward (rasterizer.py:219-276).  The reference transforms the PADDED vertices with two batched
    4x4 `transform_points` (homogeneous divides, `cat`, `update_padded` -> a new Meshes): ~4 ms of small launches and
    Python per call on the bench batch, more than the rasterization itself.  Here: both matrices from the cameras, one
    kernel on the packed vertices, then the fused rasterize_meshes (its own z-clipping / culling) on a view of the mesh's
    topology.  Falls back to the reference's forward when the cameras have no matrix form, need an `eps`, or their
    matrices require grad (camera optimisation: torch autograd through transform_points)."""
    import importlib

    import torch

    rz = importlib.import_module("pytorch3d.renderer.mesh.rasterizer")
    cam_utils = importlib.import_module("pytorch3d.renderer.cameras")
    orig = rz.MeshRasterizer.forward

    def forward(self, meshes_world, **kwargs):
        cameras = kwargs.get("cameras", self.cameras)
        ok = cameras is not None and kwargs.get("eps", None) is None
        if ok:
            try:
                verts = meshes_world.verts_packed()
                ok = _is_hip_f32(verts) and meshes_world.faces_packed().is_cuda and len(cameras) in (1, len(meshes_world))
                if ok:
                    cm = camera_matrices(cameras, kwargs)
                    ok = cm is not None
                    if ok:
                        w2v, v2n, cam_persp, znear = cm
                        ok = not (w2v.requires_grad or v2n.requires_grad) and w2v.device == verts.device
            except Exception:
                ok = False
        _count("MeshRasterizer.forward", ok)
        if not ok:
            return orig(self, meshes_world, **kwargs)
        rs = kwargs.get("raster_settings", self.raster_settings)
        clip_bary = rs.clip_barycentric_coords
        if clip_bary is None:
            clip_bary = rs.blur_radius > 0.0
        persp = rs.perspective_correct if rs.perspective_correct is not None else cam_persp
        if rs.z_clip_value is not None:
            z_clip = rs.z_clip_value
        else:
            z_clip = None if not persp or znear is None else znear / 2
        ndc = our_rm.transform_verts_to_ndc(meshes_world, w2v, v2n)
        view = our_rm._PackedVertsView(meshes_world, ndc)
        common = dict(image_size=rs.image_size, blur_radius=rs.blur_radius, faces_per_pixel=rs.faces_per_pixel, bin_size=rs.bin_size,
                      max_faces_per_bin=rs.max_faces_per_bin, clip_barycentric_coords=clip_bary, perspective_correct=persp,
                      cull_backfaces=rs.cull_backfaces)
        if z_clip is not None and not rs.cull_to_frustum and SPECULATE_NO_CLIPPING:
            # Near-plane clipping alone (the default for perspective cameras, rasterizer.py:244-251): a face is cut or dropped iff
            # one of its vertices has z < z_clip (clip.py:381-388, strict).  clip_faces has to tell the host how many faces
            # come out of it -- a sync in the MIDDLE of the forward, behind which the host cannot run ahead (measured: 0.4 ms of
            # idle GPU per step on the bench batch).  Nearly always nothing is behind the plane, so: queue the un-clipped, fully
            # fused rasterization (gather, rasterizer and backward-to-vertices in one node: no face gather, no clip plan, no
            # scatter launch) and ask the device afterwards whether ANY vertex was behind the plane -- one sync at the END of the
            # forward, with the whole forward already running.  If one was (rare): that result is dropped and the clipping
            # path below runs.  (A vertex no face uses can only send a mesh down the slow path, never the other way.)
            behind = (ndc[:, 2] < z_clip).any()
            out = our_rm.rasterize_meshes(view, z_clip_value=None, cull_to_frustum=False, **common)
            if not bool(behind):  # the host sync
                _count("MeshRasterizer.forward: no vertex behind z_clip, un-clipped fused path kept", True)
                return rz.Fragments(pix_to_face=out[0], zbuf=out[1], bary_coords=out[2], dists=out[3])
            _count("MeshRasterizer.forward: no vertex behind z_clip, un-clipped fused path kept", False)
            del out
        p2f, zbuf, bary, dists = our_rm.rasterize_meshes(view, z_clip_value=z_clip, cull_to_frustum=rs.cull_to_frustum, **common)
        return rz.Fragments(pix_to_face=p2f, zbuf=zbuf, bary_coords=bary, dists=dists)

    forward.__wrapped__ = orig
    rz.MeshRasterizer.forward = forward
    _PATCHED.append((rz.MeshRasterizer, "forward", orig, forward))


def _patch_points_rasterizer(our_rm):
    """PointsRasterizer.forward (renderer/points/rasterizer.py:118-168).  The reference transforms the PADDED points with two
    batched 4x4 `transform_points` (four hipBLASLt GEMMs, cat, homogeneous divides, `update_padded` -> a new Pointclouds:
    ~1 ms of GPU time and ~50 small copies per call at 1M points) before `rasterize_points`.  Here: both matrices from the
    cameras, ONE kernel on the packed points (csrc/transform.hip: p3d_transform_verts_forward / _backward, the kernel
    MeshRasterizer.forward uses for vertices: x, y to NDC, z = view depth, as rasterizer.py:140-141 keeps it) and the
    rasterizer's own autograd node.  Falls back to the reference's forward when the cameras have no matrix form, need an
    `eps`, or their matrices require grad.

    PointsRenderer.forward (renderer/points/renderer.py:56-76) with the plain PointsRasterizer and AlphaCompositor or NormWeightedCompositor, float32
    (P, C <= 4) features, one scalar radius and K <= 16: the whole chain is pytorch3d_amd.render_points' fused node -- the image is
    formed in the fine kernel's epilogue, the backward is one kernel.  The packed tensors are taken from the cloud's lists when it has
    not packed them yet: `Pointclouds.points_packed()` builds a cloud-index entry per point with arange + bucketize, twice (points and
    features, structures/utils.py:119-154) -- ~30 launches per freshly built cloud for tensors this chain never reads."""
    import importlib

    import torch

    pr = importlib.import_module("pytorch3d.renderer.points.rasterizer")
    rr = importlib.import_module("pytorch3d.renderer.points.renderer")
    pc = importlib.import_module("pytorch3d.renderer.points.compositor")
    structures = importlib.import_module("pytorch3d.structures")
    our_rp = importlib.import_module(__package__ + ".rasterize_points")
    our_rd = importlib.import_module(__package__ + ".render_points")
    orig = pr.PointsRasterizer.forward
    orig_render = rr.PointsRenderer.forward

    def packed_of(clouds, want_features):
        """(points_packed, features_packed or None, first, count) without `_compute_packed` where the cloud still holds its lists."""
        if type(clouds) is structures.Pointclouds and clouds._points_packed is None and clouds._points_list is not None:
            pl = clouds._points_list
            fl = getattr(clouds, "_features_list", None) if want_features else None
            if len(pl) > 0 and all(torch.is_tensor(t) and t.dim() == 2 and t.shape[1] == 3 for t in pl) and (
                    not want_features or (fl is not None and len(fl) == len(pl) and all(torch.is_tensor(t) and t.dim() == 2 for t in fl))):
                count = clouds.num_points_per_cloud()  # built by the constructor
                if len(pl) == 1:
                    first = _small_zeros(pl[0].device)
                    return pl[0], (fl[0] if want_features else None), first, count
                first = torch.cumsum(count, 0) - count
                return torch.cat(pl, 0), (torch.cat(fl, 0) if want_features else None), first, count
        return (clouds.points_packed(), clouds.features_packed() if want_features else None, clouds.cloud_to_packed_first_idx(),
                clouds.num_points_per_cloud())

    def prepare(self, point_clouds, kwargs, want_features=False):
        """(ndc points, features, first, count) when this rasterizer call can run on the package's kernels, else None."""
        cameras = kwargs.get("cameras", self.cameras)
        if cameras is None or kwargs.get("eps", None) is not None:
            return None
        try:
            pts, feats, first, count = packed_of(point_clouds, want_features)
            if not (_is_hip_f32(pts) and pts.dim() == 2 and pts.shape[1] == 3 and len(cameras) in (1, len(point_clouds))):
                return None
            cm = camera_matrices(cameras, kwargs)
            if cm is None:
                return None
            w2v, v2n = cm[0], cm[1]
            if w2v.requires_grad or v2n.requires_grad or w2v.device != pts.device:
                return None
        except Exception:
            return None
        mats = _packed_matrices(our_rm, cameras, w2v, v2n, len(point_clouds), pts.device)
        return our_rm._TransformVerts.apply(pts, first.contiguous(), mats), feats, first, count

    def forward(self, point_clouds, **kwargs):
        got = prepare(self, point_clouds, kwargs)
        _count("PointsRasterizer.forward", got is not None)
        if got is None:
            return orig(self, point_clouds, **kwargs)
        ndc, _, first, count = got
        rs = kwargs.get("raster_settings", self.raster_settings)
        idx, zbuf, dists2 = our_rp.rasterize_points(_PackedPointsView(point_clouds, ndc, first, count), image_size=rs.image_size,
                                                    radius=rs.radius, points_per_pixel=rs.points_per_pixel, bin_size=rs.bin_size,
                                                    max_points_per_bin=rs.max_points_per_bin)
        return pr.PointFragments(idx=idx, zbuf=zbuf, dists=dists2)

    def render(self, point_clouds, **kwargs):
        rz = self.rasterizer
        mode = {pc.AlphaCompositor: "alpha", pc.NormWeightedCompositor: "norm"}.get(type(self.compositor))
        ok = FUSE_POINTS_RENDERER and mode is not None and type(rz).forward is forward
        got = None
        if ok:
            rs = kwargs.get("raster_settings", rz.raster_settings)
            r_w = rz.raster_settings.radius  # renderer.py:62: the weights' radius is the rasterizer's OWN setting
            ok = (isinstance(rs.radius, (float, int)) and not isinstance(rs.radius, bool) and isinstance(r_w, (float, int))
                  and not isinstance(r_w, bool) and r_w > 0 and 0 < int(rs.points_per_pixel) <= our_rd.MAX_FUSED_K)
        if ok:
            got = prepare(rz, point_clouds, kwargs, want_features=True)
            ok = got is not None and our_rd.fusable(got[1], r_w, rs.points_per_pixel)
        _count("PointsRenderer.forward", ok)
        if not ok:
            return orig_render(self, point_clouds, **kwargs)
        ndc, feats, first, count = got
        view = _PackedPointsView(point_clouds, ndc, first, count)
        images, idx, _, _ = our_rd.render_points_alpha(view, feats, image_size=rs.image_size, radius=rs.radius,
                                                       points_per_pixel=rs.points_per_pixel, bin_size=rs.bin_size,
                                                       max_points_per_bin=rs.max_points_per_bin, weight_radius=r_w,
                                                       radius_per_point=_scalar_radius(rs.radius, ndc), compositor=mode)
        background_color = kwargs.get("background_color", self.compositor.background_color)
        if background_color is not None:  # compositor.py:41-46, on (N, C, H, W) views
            images = pc._add_background_color_to_images(idx.long().permute(0, 3, 1, 2), images.permute(0, 3, 1, 2),
                                                        background_color).permute(0, 2, 3, 1)
        return images

    forward.__wrapped__ = orig
    pr.PointsRasterizer.forward = forward
    _PATCHED.append((pr.PointsRasterizer, "forward", orig, forward))
    render.__wrapped__ = orig_render
    rr.PointsRenderer.forward = render
    _PATCHED.append((rr.PointsRenderer, "forward", orig_render, render))


def _packed_matrices(our_rm, cameras, w2v, v2n, n, device):
    """rasterize_meshes._pack_matrices(w2v, v2n) -- one stack + copy per call -- kept on the camera object for as long as
    camera_matrices hands out the SAME two matrix tensors (its cache: the camera's parameters have not changed)."""
    hit = cameras.__dict__.get("_p3d_amd_packed")
    if hit is not None and hit[0] is w2v and hit[1] is v2n and hit[2] == (n, str(device)):
        return hit[3]
    mats = our_rm._pack_matrices(w2v, v2n, n, device)
    if cameras.__dict__.get("_p3d_amd_matrices") is not None and not mats.requires_grad:
        cameras.__dict__["_p3d_amd_packed"] = (w2v, v2n, (n, str(device)), mats)
    return mats


FUSE_POINTS_RENDERER = True  # (tests and profiles switch the fused PointsRenderer off to time / compare the operator chain)
_SMALL = {}


def _small_zeros(device):
    """A cached int64 zeros(1) per device: cloud_to_packed_first_idx of a single cloud (read-only here)."""
    import torch

    key = ("zeros1", str(device))
    if key not in _SMALL:
        _SMALL[key] = torch.zeros(1, dtype=torch.int64, device=device)
    return _SMALL[key]


def _scalar_radius(radius, points):
    """The (P,) radius tensor of a scalar radius; the last one is kept (a renderer is called with the same cloud size over and over,
    and the tensor is read-only inside the package)."""
    import torch

    key = ("radius", str(points.device))
    hit = _SMALL.get(key)
    if hit is not None and hit[0] == (float(radius), points.shape[0]):
        return hit[1]
    t = torch.full((points.shape[0],), float(radius), dtype=torch.float32, device=points.device)
    _SMALL[key] = ((float(radius), points.shape[0]), t)
    return t


class _PackedPointsView:
    """The accessors pytorch3d_amd.rasterize_points reads, with the packed points replaced (everything else is the cloud's)."""

    def __init__(self, clouds, points_packed, first=None, count=None):
        self._clouds, self._points, self._first, self._count = clouds, points_packed, first, count

    def __len__(self):
        return len(self._clouds)

    def points_packed(self):
        return self._points

    def cloud_to_packed_first_idx(self):
        return self._first if self._first is not None else self._clouds.cloud_to_packed_first_idx()

    def num_points_per_cloud(self):
        return self._count if self._count is not None else self._clouds.num_points_per_cloud()

    def padded_to_packed_idx(self):
        return self._clouds.padded_to_packed_idx()

    @property
    def _P(self):
        return self._clouds._P


def _patch_soft_phong_shader(our_shade):
    """SoftPhongShader.forward (shader.py:113-147): texels = meshes.sample_textures(fragments); phong_shading;
    softmax_rgb_blend -> pytorch3d_amd.shading.soft_phong_shading (the colours never reach memory).  With TexturesVertex of
    three channels the texel interpolation is fused in as well.  Falls back to the reference's forward (whose pieces are
    themselves patched) whenever the fused kernels do not cover the call."""
    import importlib

    shader = importlib.import_module("pytorch3d.renderer.mesh.shader")
    shading_mod = importlib.import_module("pytorch3d.renderer.mesh.shading")
    blend_mod = importlib.import_module("pytorch3d.renderer.blending")
    orig = shader.SoftPhongShader.forward

    def forward(self, fragments, meshes, **kwargs):
        try:
            cameras = kwargs.get("cameras", self.cameras)
            lights = kwargs.get("lights", self.lights)
            materials = kwargs.get("materials", self.materials)
            blend_params = kwargs.get("blend_params", self.blend_params)
            ok = (cameras is not None and _shading_ok(meshes, fragments, lights, cameras, materials)
                  and our_shade.soft_phong_supported(fragments) and _is_hip_f32(fragments.dists) and _is_hip_f32(fragments.zbuf)
                  and not getattr(blend_params.background_color, "requires_grad", False))
        except Exception:
            ok = False
        vcol = texels = None
        if ok:
            tex = getattr(meshes, "textures", None)
            if type(tex).__name__ == "TexturesVertex" and tex.verts_features_packed().shape[-1] == 3:
                vcol = tex.verts_features_packed()
                ok = _is_hip_f32(vcol)
            else:
                texels = meshes.sample_textures(fragments)
                ok = _is_hip_f32(texels) and texels.dim() == 5 and texels.shape[-1] == 3
        _count("SoftPhongShader.forward", ok)
        if not ok:
            if texels is not None:
                # the textures were sampled for the probe and turned out not to fit the fused kernel (channels, dtype): finish the
                # reference's own forward with them (shader.py:131-146) instead of letting it sample them a second time
                colors = shading_mod.phong_shading(meshes=meshes, fragments=fragments, texels=texels, lights=lights, cameras=cameras,
                                                   materials=materials)
                return blend_mod.softmax_rgb_blend(colors, fragments, blend_params, znear=kwargs.get("znear", getattr(cameras, "znear", 1.0)),
                                                   zfar=kwargs.get("zfar", getattr(cameras, "zfar", 100.0)))
            return orig(self, fragments, meshes, **kwargs)
        znear = kwargs.get("znear", getattr(cameras, "znear", 1.0))
        zfar = kwargs.get("zfar", getattr(cameras, "zfar", 100.0))
        return our_shade.soft_phong_shading(meshes, fragments, lights, cameras, materials, texels, blend_params, znear=znear, zfar=zfar,
                                            verts_colors_packed=vcol)

    forward.__wrapped__ = orig
    shader.SoftPhongShader.forward = forward
    _PATCHED.append((shader.SoftPhongShader, "forward", orig, forward))


def _patch_hard_and_silhouette_shaders():
    """HardPhongShader / HardGouraudShader / HardFlatShader (shader.py:81-112, 150-185, 245-275) shade and texture ALL K slots of
    every pixel and then `hard_rgb_blend` keeps slot 0 (blending.py:54-88: `pix_to_face[..., 0]`, `colors[..., 0, :]`): K - 1 of K
    samples are computed, stored and differentiated for nothing.  Here the shader sees the fragments cut to their nearest slot
    (contiguous copies of 1 / K of the data; autograd routes the gradient back into slot 0 of the originals, the other slots
    get the zeros the reference's blend gives them) -- same image, same gradients, 1 / K of the shading and texture work.
    SoftSilhouetteShader (shader.py:277-300) materialises `torch.ones_like(bary_coords)` -- 1.6 GB at the bench batch -- to
    read one constant pixel of it: a stride-0 view of a single 1 serves the reference's `sigmoid_alpha_blend` as well."""
    import importlib

    import torch

    shader = importlib.import_module("pytorch3d.renderer.mesh.shader")
    rz = importlib.import_module("pytorch3d.renderer.mesh.rasterizer")
    blend = importlib.import_module("pytorch3d.renderer.blending")

    def nearest_slot(orig, name):
        def forward(self, fragments, meshes, **kwargs):
            ok = False
            try:
                p2f = fragments.pix_to_face
                ok = torch.is_tensor(p2f) and p2f.dim() == 4 and p2f.shape[3] > 1 and torch.is_tensor(fragments.bary_coords)
            except Exception:
                ok = False
            first = None
            if ok:
                try:  # fragments of another shape than the rasterizer's (a subclass, missing fields): the reference's own path
                    first = rz.Fragments(pix_to_face=fragments.pix_to_face[..., :1].contiguous(), zbuf=fragments.zbuf[..., :1].contiguous(),
                                         bary_coords=fragments.bary_coords[..., :1, :].contiguous(),
                                         dists=fragments.dists[..., :1].contiguous())
                except Exception:
                    first = None
            _count(name + ".forward", first is not None)
            if first is None:
                return orig(self, fragments, meshes, **kwargs)
            return orig(self, first, meshes, **kwargs)

        forward.__wrapped__ = orig
        return forward

    for cls_name in ("HardPhongShader", "HardGouraudShader", "HardFlatShader"):
        cls = getattr(shader, cls_name, None)
        if cls is None:
            continue
        orig = cls.forward
        new = nearest_slot(orig, cls_name)
        cls.forward = new
        _PATCHED.append((cls, "forward", orig, new))

    sil = getattr(shader, "SoftSilhouetteShader", None)
    if sil is not None:
        sil_orig = sil.forward

        def sil_forward(self, fragments, meshes, **kwargs):
            ok = False
            try:
                ok = torch.is_tensor(fragments.bary_coords) and fragments.bary_coords.dim() == 5
            except Exception:
                ok = False
            _count("SoftSilhouetteShader.forward", ok)
            if not ok:
                return sil_orig(self, fragments, meshes, **kwargs)
            b = fragments.bary_coords
            colors = torch.ones((1, 1, 1, 1, 1), dtype=b.dtype, device=b.device).expand(b.shape)  # shader.py:154 without the 12 K bytes per pixel
            blend_params = kwargs.get("blend_params", self.blend_params)
            return blend.sigmoid_alpha_blend(colors, fragments, blend_params)

        sil_forward.__wrapped__ = sil_orig
        sil.forward = sil_forward
        _PATCHED.append((sil, "forward", sil_orig, sil_forward))


def _patch_meshes_offset_verts():
    """Meshes.offset_verts / offset_verts_ (structures/meshes.py:1295-1360), what every mesh-fitting loop of the reference's
    tutorials calls once per step.  The reference clones the whole object -- `clone()` re-runs `Meshes.__init__` on lists
    (per mesh a boolean-mask index of the faces = one host sync each, `int(max())` and `unique()` syncs) and clones ~20
    internal tensors -- and then reads `num_verts_per_mesh().tolist()` from the device: ~4.5 ms of CPU and ~70 syncs per
    call on the 64-mesh bench batch, twice the rasterizer's forward + backward, and the syncs keep the CPU from ever running
    ahead of the GPU (profiles/r03/dropin_breakdown.py).  Here: one elementwise add on the packed vertices; the new object
    SHARES every topology tensor and cache of the old one (faces, packed / padded indices, edges, Laplacian -- none of them
    depends on the vertex positions and the Meshes API has no in-place edit of them; the reference would hand out clones),
    its vertex list is 64 views of the new packed tensor split by sizes kept on the host, the padded vertices are dropped
    (recomputed lazily by `verts_padded()`), face areas / normals and vertex normals are recomputed if and only if the
    original had them, as the reference does, and textures are cloned as `clone()` clones them.  Falls back to the
    reference's method for empty meshes and offsets of another dtype / device / shape."""
    import importlib

    import torch

    meshes_mod = importlib.import_module("pytorch3d.structures.meshes")
    Meshes = meshes_mod.Meshes
    orig = Meshes.offset_verts
    orig_ = Meshes.offset_verts_

    def usable(self, off):
        if type(self) is not Meshes:  # a subclass may keep state of its own that clone() copies and a shared __dict__ would alias
            return False
        v = self.verts_packed()
        # `isempty()` reads `valid.eq(False).all()` from the device: a host sync per call (measured in round 5: 0.95 ms of a 3.0 ms
        # step -- the host waited there for the previous step's backward).  Emptiness is a property of the topology: asked once,
        # kept on the object and inherited by the copies made below, like the vertex counts.
        empty = self.__dict__.get("_p3d_amd_isempty")
        if empty is None:
            empty = bool(self.isempty())
            self.__dict__["_p3d_amd_isempty"] = empty
        return (torch.is_tensor(off) and self._N > 0 and not empty and v.dtype == torch.float32 and off.device == v.device
                and off.dtype == torch.float32 and (off.shape == v.shape or tuple(off.shape) == (3,)))

    def host_sizes(self):
        sizes = self.__dict__.get("_p3d_amd_num_verts")
        if sizes is None:
            sizes = self.num_verts_per_mesh().tolist()  # once per topology: the copies made below inherit it
            self.__dict__["_p3d_amd_num_verts"] = sizes
        return sizes

    def apply(dst, src, off):
        update_normals = tuple(off.shape) != (3,)
        dst._verts_packed = src.verts_packed() + off
        dst._verts_list = list(dst._verts_packed.split(host_sizes(src), 0))
        dst._verts_padded = None
        if update_normals and (src._faces_areas_packed is not None or src._faces_normals_packed is not None):
            dst._compute_face_areas_normals(refresh=True)
        if update_normals and src._verts_normals_packed is not None:
            dst._compute_vertex_normals(refresh=True)
        return dst

    def offset_verts(self, vert_offsets_packed):
        ok = False
        try:
            ok = usable(self, vert_offsets_packed)
        except Exception:
            ok = False
        _count("Meshes.offset_verts", ok)
        if not ok:
            return orig(self, vert_offsets_packed)
        self.verts_list(), self.faces_list()  # the lists exist before they are shared (meshes built from padded tensors)
        new = object.__new__(type(self))
        new.__dict__.update(self.__dict__)
        # the list OBJECTS are the copy's own (appending to / replacing an element of new.faces_list() must not show in the original);
        # the tensors in them are shared with the original, where the reference's clone() hands out copies: INTEGRATION.md
        for name in ("_faces_list", "_verts_list"):
            if isinstance(new.__dict__.get(name), list):
                new.__dict__[name] = list(new.__dict__[name])
        if self.textures is not None:
            new.textures = self.textures.clone()
        return apply(new, self, vert_offsets_packed)

    def offset_verts_(self, vert_offsets_packed):
        ok = False
        try:
            ok = usable(self, vert_offsets_packed)
        except Exception:
            ok = False
        _count("Meshes.offset_verts_", ok)
        if not ok:
            return orig_(self, vert_offsets_packed)
        return apply(self, self, vert_offsets_packed)

    offset_verts.__wrapped__ = orig
    offset_verts_.__wrapped__ = orig_
    Meshes.offset_verts = offset_verts
    Meshes.offset_verts_ = offset_verts_
    _PATCHED.append((Meshes, "offset_verts", orig, offset_verts))
    _PATCHED.append((Meshes, "offset_verts_", orig_, offset_verts_))


def uninstall_python_patches():
    """Put the reference's own functions back (the `_C` module stays installed)."""
    while _PATCHED:
        owner, attr, orig, _new = _PATCHED.pop()
        setattr(owner, attr, orig)
