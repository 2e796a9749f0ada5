// bind.cpp -- the pybind flavour of the drop-in boundary: `pytorch3d._C`'s hot-path operators (pytorch3d/csrc/ext.cpp:38-73)
// as a compiled torch extension over the C ABI of include/p3d_amd.h.
//
// This is INTEGRATION.md section B for real: what a maintainer of the reference would compile into its own extension on a
// ROCm build -- validation, allocation, device guard, current stream (the reference does the same in
// rasterize_meshes.cu:375-402 etc.), then ONE call into libp3d_amd.so.  No torch type crosses the ABI.  The default flavour of
// this package stays the ctypes module pytorch3d_amd/_C.py (no compiler needed where the library is used); this one is built by
// pytorch3d_amd/build_bind.py, selected with `pytorch3d_amd.shim.install(flavour="pybind")`, and tested against the ctypes one
// bit for bit (tests/test_gpu_pybind_boundary.py).  It carries none of _C.py's extras: no row-cover recall, no short
// workspaces, no CUDA tie order -- the reference's operator set with the reference's allocation pattern (one at::empty per call,
// worst-case workspace).
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <algorithm>
#include <tuple>

#include "p3d_amd.h"

namespace {

using at::Tensor;
// PyTorch-ROCm tensors carry the device type "cuda": the guard and the stream are the ones that know (plain c10::hip::HIPGuard
// refuses them: "HIPGuardImpl initialized with non-HIP DeviceType")
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;

void check_gpu(std::initializer_list<std::pair<const Tensor*, const char*>> ts) {
  const Tensor* first = nullptr;
  for (auto& t : ts) {
    TORCH_CHECK(t.first->is_cuda(), t.second, " must be a GPU tensor (pytorch3d_amd has no CPU path: the reference's CPU operators are its own)");
    if (first == nullptr) first = t.first;
    TORCH_CHECK(t.first->device() == first->device(), t.second, " is on another device than ", ts.begin()->second);
  }
}

void ok(int rc, const char* what) { TORCH_CHECK(rc == P3D_OK, what, ": ", p3d_error_string(rc)); }

p3d_stream_t stream_of(const Tensor& t) { return (p3d_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

Tensor workspace(size_t bytes, const Tensor& like) {
  return at::empty({(int64_t)std::max<size_t>(bytes, 256)}, like.options().dtype(at::kByte));
}

void check_bins(int H, int W, int bin_size) {  // rasterize_coarse.cu:244-249
  TORCH_CHECK(bin_size > 0, "bin_size must be positive");
  const int bins = 1 + (std::max(H, W) - 1) / bin_size;
  TORCH_CHECK(bins < 22, "In RasterizeCoarseCuda got num_bins_y: ", 1 + (H - 1) / bin_size, ", num_bins_x: ", 1 + (W - 1) / bin_size,
              ", ; that's too many!");
}

void check_face_verts(const Tensor& fv) {
  TORCH_CHECK(fv.dim() == 3 && fv.size(1) == 3 && fv.size(2) == 3, "face_verts must have dimensions (num_faces, 3, 3)");
}

// ---- meshes -----------------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor> rasterize_meshes(const Tensor& face_verts, const Tensor& mesh_to_face_first_idx,
                                                            const Tensor& num_faces_per_mesh, const Tensor& clipped_faces_neighbor_idx,
                                                            std::tuple<int, int> image_size, double blur_radius, int faces_per_pixel,
                                                            int bin_size, int max_faces_per_bin, bool perspective_correct,
                                                            bool clip_barycentric_coords, bool cull_backfaces) {
  check_gpu({{&face_verts, "face_verts"}, {&mesh_to_face_first_idx, "mesh_to_face_first_idx"}, {&num_faces_per_mesh, "num_faces_per_mesh"},
             {&clipped_faces_neighbor_idx, "clipped_faces_neighbor_idx"}});
  check_face_verts(face_verts);
  TORCH_CHECK(num_faces_per_mesh.size(0) == mesh_to_face_first_idx.size(0), "num_faces_per_mesh must have save size first dimension as mesh_to_face_first_idx");
  TORCH_CHECK(clipped_faces_neighbor_idx.size(0) == face_verts.size(0), "clipped_faces_neighbor_idx must have save size first dimension as face_verts");
  TORCH_CHECK(faces_per_pixel <= P3D_MAX_K, "Must have points_per_pixel <= 150");  // rasterize_meshes.cu:361-365
  const int H = std::get<0>(image_size), W = std::get<1>(image_size), K = faces_per_pixel;
  const bool binned = bin_size > 0 && max_faces_per_bin > 0;
  if (binned) check_bins(H, W, bin_size);
  DeviceGuard guard(face_verts.device());
  auto fv = face_verts.contiguous().to(at::kFloat);
  auto first = mesh_to_face_first_idx.contiguous().to(at::kLong), count = num_faces_per_mesh.contiguous().to(at::kLong);
  auto nbr = clipped_faces_neighbor_idx.contiguous().to(at::kLong);
  const int N = (int)count.size(0);
  const int64_t F = fv.size(0);
  auto lopts = fv.options().dtype(at::kLong), fopts = fv.options().dtype(at::kFloat);
  // uninitialised: the kernels write every element, -1 padding included (no at::full pre-fill, rasterize_meshes.cu:788-791)
  auto p2f = at::empty({N, H, W, K}, lopts), zbuf = at::empty({N, H, W, K}, fopts);
  auto bary = at::empty({N, H, W, K, 3}, fopts), dists = at::empty({N, H, W, K}, fopts);
  if (p2f.numel() == 0) return {p2f, zbuf, bary, dists};
  auto ws = workspace(binned ? p3d_rasterize_meshes_workspace_bytes(F, N, H, W, bin_size, max_faces_per_bin) : 0, fv);
  ok(p3d_rasterize_meshes(fv.data_ptr<float>(), first.data_ptr<int64_t>(), count.data_ptr<int64_t>(), nbr.data_ptr<int64_t>(), F, N, H, W,
                          (float)blur_radius, K, binned ? bin_size : 0, binned ? max_faces_per_bin : 0, perspective_correct,
                          clip_barycentric_coords, cull_backfaces, p2f.data_ptr<int64_t>(), zbuf.data_ptr<float>(), bary.data_ptr<float>(),
                          dists.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(), stream_of(fv)),
     "rasterize_meshes");
  return {p2f, zbuf, bary, dists};
}

std::tuple<Tensor, Tensor, Tensor, Tensor> rasterize_meshes_naive(const Tensor& face_verts, const Tensor& first, const Tensor& count,
                                                                  const Tensor& nbr, std::tuple<int, int> image_size, double blur_radius,
                                                                  int faces_per_pixel, bool perspective_correct,
                                                                  bool clip_barycentric_coords, bool cull_backfaces) {
  return rasterize_meshes(face_verts, first, count, nbr, image_size, blur_radius, faces_per_pixel, 0, 0, perspective_correct,
                          clip_barycentric_coords, cull_backfaces);
}

Tensor rasterize_meshes_coarse(const Tensor& face_verts, const Tensor& mesh_to_face_first_idx, const Tensor& num_faces_per_mesh,
                               std::tuple<int, int> image_size, double blur_radius, int bin_size, int max_faces_per_bin) {
  check_gpu({{&face_verts, "face_verts"}, {&mesh_to_face_first_idx, "mesh_to_face_first_idx"}, {&num_faces_per_mesh, "num_faces_per_mesh"}});
  check_face_verts(face_verts);
  const int H = std::get<0>(image_size), W = std::get<1>(image_size);
  check_bins(H, W, bin_size);
  DeviceGuard guard(face_verts.device());
  auto fv = face_verts.contiguous().to(at::kFloat);
  auto first = mesh_to_face_first_idx.contiguous().to(at::kLong), count = num_faces_per_mesh.contiguous().to(at::kLong);
  const int N = (int)count.size(0), BH = 1 + (H - 1) / bin_size, BW = 1 + (W - 1) / bin_size;
  const int64_t F = fv.size(0);
  auto out = at::empty({N, BH, BW, max_faces_per_bin}, fv.options().dtype(at::kInt));
  if (out.numel() == 0) return out;
  auto ws = workspace(p3d_rasterize_meshes_workspace_bytes(F, N, H, W, bin_size, max_faces_per_bin), fv);
  ok(p3d_rasterize_meshes_coarse(fv.data_ptr<float>(), first.data_ptr<int64_t>(), count.data_ptr<int64_t>(), F, N, H, W, (float)blur_radius,
                                 bin_size, max_faces_per_bin, out.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(), stream_of(fv)),
     "_rasterize_meshes_coarse");
  return out;
}

std::tuple<Tensor, Tensor, Tensor, Tensor> rasterize_meshes_fine(const Tensor& face_verts, const Tensor& bin_faces,
                                                                 const Tensor& clipped_faces_neighbor_idx, std::tuple<int, int> image_size,
                                                                 double blur_radius, int bin_size, int faces_per_pixel,
                                                                 bool perspective_correct, bool clip_barycentric_coords, bool cull_backfaces) {
  check_gpu({{&face_verts, "face_verts"}, {&bin_faces, "bin_faces"}, {&clipped_faces_neighbor_idx, "clipped_faces_neighbor_idx"}});
  check_face_verts(face_verts);
  TORCH_CHECK(bin_faces.dim() == 4, "bin_faces must have 4 dimensions");
  TORCH_CHECK(clipped_faces_neighbor_idx.size(0) == face_verts.size(0), "clipped_faces_neighbor_idx must have the same first dimension as face_verts");
  TORCH_CHECK(faces_per_pixel <= P3D_MAX_K, "Must have num_closest <= 150");
  const int H = std::get<0>(image_size), W = std::get<1>(image_size), K = faces_per_pixel;
  DeviceGuard guard(face_verts.device());
  auto fv = face_verts.contiguous().to(at::kFloat);
  auto bf = bin_faces.contiguous().to(at::kInt);
  auto nbr = clipped_faces_neighbor_idx.contiguous().to(at::kLong);
  const int N = (int)bf.size(0), BH = (int)bf.size(1), BW = (int)bf.size(2), M = (int)bf.size(3);
  auto lopts = fv.options().dtype(at::kLong), fopts = fv.options().dtype(at::kFloat);
  auto p2f = at::empty({N, H, W, K}, lopts), zbuf = at::empty({N, H, W, K}, fopts);
  auto bary = at::empty({N, H, W, K, 3}, fopts), dists = at::empty({N, H, W, K}, fopts);
  if (p2f.numel() == 0) return {p2f, zbuf, bary, dists};
  auto ws = workspace(p3d_rasterize_fine_workspace_bytes(N, BH, BW, M), fv);
  ok(p3d_rasterize_meshes_fine(fv.data_ptr<float>(), bf.data_ptr<int32_t>(), nbr.data_ptr<int64_t>(), fv.size(0), N, BH, BW, M, H, W,
                               (float)blur_radius, bin_size, K, perspective_correct, clip_barycentric_coords, cull_backfaces,
                               p2f.data_ptr<int64_t>(), zbuf.data_ptr<float>(), bary.data_ptr<float>(), dists.data_ptr<float>(), ws.data_ptr(),
                               (size_t)ws.numel(), stream_of(fv)),
     "_rasterize_meshes_fine");
  return {p2f, zbuf, bary, dists};
}

Tensor rasterize_meshes_backward(const Tensor& face_verts, const Tensor& pix_to_face, const Tensor& grad_zbuf, const Tensor& grad_bary,
                                 const Tensor& grad_dists, bool perspective_correct, bool clip_barycentric_coords) {
  check_gpu({{&face_verts, "face_verts"}, {&pix_to_face, "pix_to_face"}, {&grad_zbuf, "grad_zbuf"}, {&grad_bary, "grad_bary"},
             {&grad_dists, "grad_dists"}});
  // float atomics: the accumulation order is not deterministic (rasterize_meshes.cu:587)
  at::globalContext().alertNotDeterministic("RasterizeMeshesBackwardCuda");
  DeviceGuard guard(face_verts.device());
  auto fv = face_verts.contiguous().to(at::kFloat);
  auto p2f = pix_to_face.contiguous().to(at::kLong);
  auto gz = grad_zbuf.contiguous().to(at::kFloat), gb = grad_bary.contiguous().to(at::kFloat), gd = grad_dists.contiguous().to(at::kFloat);
  const int N = (int)p2f.size(0), H = (int)p2f.size(1), W = (int)p2f.size(2), K = (int)p2f.size(3);
  const int64_t F = fv.size(0);
  auto out = at::empty({F, 3, 3}, fv.options());
  if (F == 0) return out;
  ok(p3d_rasterize_meshes_backward(fv.data_ptr<float>(), p2f.data_ptr<int64_t>(), gz.data_ptr<float>(), gb.data_ptr<float>(),
                                   gd.data_ptr<float>(), F, N, H, W, K, perspective_correct, clip_barycentric_coords, out.data_ptr<float>(),
                                   stream_of(fv)),
     "rasterize_meshes_backward");
  return out;
}

// ---- points -----------------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> rasterize_points(const Tensor& points, const Tensor& cloud_to_packed_first_idx,
                                                    const Tensor& num_points_per_cloud, std::tuple<int, int> image_size, const Tensor& radius,
                                                    int points_per_pixel, int bin_size, int max_points_per_bin) {
  check_gpu({{&points, "points"}, {&cloud_to_packed_first_idx, "cloud_to_packed_first_idx"}, {&num_points_per_cloud, "num_points_per_cloud"},
             {&radius, "radius"}});
  TORCH_CHECK(points.dim() == 2 && points.size(1) == 3, "points must have dimensions (num_points, 3)");
  TORCH_CHECK(radius.dim() == 1 && radius.size(0) == points.size(0), "radius must be of shape (P,)");
  TORCH_CHECK(points_per_pixel <= P3D_MAX_K, "Must have points_per_pixel <= 150");
  const int H = std::get<0>(image_size), W = std::get<1>(image_size), K = points_per_pixel;
  const bool binned = bin_size > 0 && max_points_per_bin > 0;
  if (binned) check_bins(H, W, bin_size);
  DeviceGuard guard(points.device());
  auto pts = points.contiguous().to(at::kFloat), rad = radius.contiguous().to(at::kFloat);
  auto first = cloud_to_packed_first_idx.contiguous().to(at::kLong), count = num_points_per_cloud.contiguous().to(at::kLong);
  const int N = (int)count.size(0);
  const int64_t P = pts.size(0);
  auto idx = at::empty({N, H, W, K}, pts.options().dtype(at::kInt));
  auto zbuf = at::empty({N, H, W, K}, pts.options()), dists = at::empty({N, H, W, K}, pts.options());
  if (idx.numel() == 0) return {idx, zbuf, dists};
  auto ws = workspace(binned ? p3d_rasterize_points_workspace_bytes(P, N, H, W, bin_size, max_points_per_bin) : 0, pts);
  ok(p3d_rasterize_points(pts.data_ptr<float>(), first.data_ptr<int64_t>(), count.data_ptr<int64_t>(), rad.data_ptr<float>(), P, N, H, W, K,
                          binned ? bin_size : 0, binned ? max_points_per_bin : 0, idx.data_ptr<int32_t>(), zbuf.data_ptr<float>(),
                          dists.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(), stream_of(pts)),
     "rasterize_points");
  return {idx, zbuf, dists};
}

std::tuple<Tensor, Tensor, Tensor> rasterize_points_naive(const Tensor& points, const Tensor& first, const Tensor& count,
                                                          std::tuple<int, int> image_size, const Tensor& radius, int points_per_pixel) {
  return rasterize_points(points, first, count, image_size, radius, points_per_pixel, 0, 0);
}

Tensor rasterize_points_coarse(const Tensor& points, const Tensor& cloud_to_packed_first_idx, const Tensor& num_points_per_cloud,
                               std::tuple<int, int> image_size, const Tensor& radius, int bin_size, int max_points_per_bin) {
  check_gpu({{&points, "points"}, {&cloud_to_packed_first_idx, "cloud_to_packed_first_idx"}, {&num_points_per_cloud, "num_points_per_cloud"},
             {&radius, "radius"}});
  TORCH_CHECK(points.dim() == 2 && points.size(1) == 3, "points must have dimensions (num_points, 3)");
  const int H = std::get<0>(image_size), W = std::get<1>(image_size);
  check_bins(H, W, bin_size);
  DeviceGuard guard(points.device());
  auto pts = points.contiguous().to(at::kFloat), rad = radius.contiguous().to(at::kFloat);
  auto first = cloud_to_packed_first_idx.contiguous().to(at::kLong), count = num_points_per_cloud.contiguous().to(at::kLong);
  const int N = (int)count.size(0), BH = 1 + (H - 1) / bin_size, BW = 1 + (W - 1) / bin_size;
  const int64_t P = pts.size(0);
  auto out = at::empty({N, BH, BW, max_points_per_bin}, pts.options().dtype(at::kInt));
  if (out.numel() == 0) return out;
  auto ws = workspace(p3d_rasterize_points_workspace_bytes(P, N, H, W, bin_size, max_points_per_bin), pts);
  ok(p3d_rasterize_points_coarse(pts.data_ptr<float>(), first.data_ptr<int64_t>(), count.data_ptr<int64_t>(), rad.data_ptr<float>(), P, N, H, W,
                                 bin_size, max_points_per_bin, out.data_ptr<int32_t>(), ws.data_ptr(), (size_t)ws.numel(), stream_of(pts)),
     "_rasterize_points_coarse");
  return out;
}

std::tuple<Tensor, Tensor, Tensor> rasterize_points_fine(const Tensor& points, const Tensor& bin_points, std::tuple<int, int> image_size,
                                                         const Tensor& radius, int bin_size, int points_per_pixel) {
  check_gpu({{&points, "points"}, {&bin_points, "bin_points"}, {&radius, "radius"}});
  TORCH_CHECK(points.dim() == 2 && points.size(1) == 3, "points must have dimensions (num_points, 3)");
  TORCH_CHECK(points_per_pixel <= P3D_MAX_K, "Must have num_closest <= 150");
  const int H = std::get<0>(image_size), W = std::get<1>(image_size), K = points_per_pixel;
  DeviceGuard guard(points.device());
  auto pts = points.contiguous().to(at::kFloat), rad = radius.contiguous().to(at::kFloat);
  auto bp = bin_points.contiguous().to(at::kInt);
  const int N = (int)bp.size(0), BH = (int)bp.size(1), BW = (int)bp.size(2), M = (int)bp.size(3);
  auto idx = at::empty({N, H, W, K}, pts.options().dtype(at::kInt));
  auto zbuf = at::empty({N, H, W, K}, pts.options()), dists = at::empty({N, H, W, K}, pts.options());
  if (idx.numel() == 0) return {idx, zbuf, dists};
  auto ws = workspace(p3d_rasterize_fine_workspace_bytes(N, BH, BW, M), pts);
  ok(p3d_rasterize_points_fine(pts.data_ptr<float>(), bp.data_ptr<int32_t>(), rad.data_ptr<float>(), pts.size(0), N, BH, BW, M, H, W, bin_size, K,
                               idx.data_ptr<int32_t>(), zbuf.data_ptr<float>(), dists.data_ptr<float>(), ws.data_ptr(), (size_t)ws.numel(),
                               stream_of(pts)),
     "_rasterize_points_fine");
  return {idx, zbuf, dists};
}

Tensor rasterize_points_backward(const Tensor& points, const Tensor& idxs, const Tensor& grad_zbuf, const Tensor& grad_dists) {
  check_gpu({{&points, "points"}, {&idxs, "idxs"}, {&grad_zbuf, "grad_zbuf"}, {&grad_dists, "grad_dists"}});
  at::globalContext().alertNotDeterministic("RasterizePointsBackwardCuda");
  DeviceGuard guard(points.device());
  auto pts = points.contiguous().to(at::kFloat);
  auto ix = idxs.contiguous().to(at::kInt);
  auto gz = grad_zbuf.contiguous().to(at::kFloat), gd = grad_dists.contiguous().to(at::kFloat);
  const int N = (int)ix.size(0), H = (int)ix.size(1), W = (int)ix.size(2), K = (int)ix.size(3);
  const int64_t P = pts.size(0);
  auto out = at::empty({P, 3}, pts.options());
  if (P == 0) return out;
  ok(p3d_rasterize_points_backward(pts.data_ptr<float>(), ix.data_ptr<int32_t>(), gz.data_ptr<float>(), gd.data_ptr<float>(), P, N, H, W, K,
                                   out.data_ptr<float>(), stream_of(pts)),
     "rasterize_points_backward");
  return out;
}

// ---- compositors (element strides cross the ABI: the renderer's permuted (N, H, W, K) views need no copy) ------------------------
void composite_check(const Tensor& features, const Tensor& alphas, const Tensor& points_idx) {
  check_gpu({{&features, "features"}, {&alphas, "alphas"}, {&points_idx, "points_idx"}});
  TORCH_CHECK(features.dim() == 2, "features must have 2 dimensions (C, P)");
  TORCH_CHECK(alphas.dim() == 4 && points_idx.dim() == 4 && alphas.sizes() == points_idx.sizes(),
              "alphas and points_idx must both have shape (N, K, H, W)");
  TORCH_CHECK(features.scalar_type() == at::kFloat && alphas.scalar_type() == at::kFloat && points_idx.scalar_type() == at::kLong,
              "features/alphas must be float32 and points_idx int64");
}

// the two layouts the kernels read as they lie: the contiguous (C, P) tensor, or the transposed view of a contiguous (P, C) one
Tensor feature_layout(const Tensor& features, int64_t st[2], bool* interleaved) {
  const int64_t C = features.size(0), P = features.size(1);
  if (C > 1 && P > 1 && features.stride(0) == 1 && features.stride(1) == C) {
    st[0] = 1, st[1] = C, *interleaved = true;
    return features;
  }
  st[0] = P, st[1] = 1, *interleaved = false;
  return features.contiguous();
}

Tensor composite_forward(int mode, const char* name, const Tensor& features, const Tensor& alphas, const Tensor& points_idx) {
  composite_check(features, alphas, points_idx);
  DeviceGuard guard(features.device());
  int64_t fst[2];
  bool il;
  auto feats = feature_layout(features, fst, &il);
  const int N = (int)alphas.size(0), K = (int)alphas.size(1), H = (int)alphas.size(2), W = (int)alphas.size(3), C = (int)feats.size(0);
  const int64_t P = feats.size(1);
  auto out = at::empty({N, C, H, W}, feats.options());
  if (out.numel() == 0) return out;
  const int64_t as[4] = {alphas.stride(0), alphas.stride(1), alphas.stride(2), alphas.stride(3)};
  const int64_t is[4] = {points_idx.stride(0), points_idx.stride(1), points_idx.stride(2), points_idx.stride(3)};
  ok(p3d_composite_forward_strided(mode, feats.data_ptr<float>(), fst, alphas.data_ptr<float>(), points_idx.data_ptr<int64_t>(), N, C, P, K, H, W,
                                   as, is, out.data_ptr<float>(), stream_of(feats)),
     name);
  return out;
}

std::tuple<Tensor, Tensor> composite_backward(int mode, const char* name, const Tensor& grad_outputs, const Tensor& features,
                                              const Tensor& alphas, const Tensor& points_idx) {
  composite_check(features, alphas, points_idx);
  check_gpu({{&grad_outputs, "grad_outputs"}});
  DeviceGuard guard(features.device());
  int64_t fst[2];
  bool il;
  auto feats = feature_layout(features, fst, &il);
  auto go = grad_outputs.contiguous().to(at::kFloat);
  const int N = (int)alphas.size(0), K = (int)alphas.size(1), H = (int)alphas.size(2), W = (int)alphas.size(3), C = (int)feats.size(0);
  const int64_t P = feats.size(1);
  auto gf = il ? at::empty({P, C}, feats.options()).t() : at::empty({C, P}, feats.options());
  auto ga = at::empty({N, K, H, W}, feats.options());
  const int64_t as[4] = {alphas.stride(0), alphas.stride(1), alphas.stride(2), alphas.stride(3)};
  const int64_t is[4] = {points_idx.stride(0), points_idx.stride(1), points_idx.stride(2), points_idx.stride(3)};
  ok(p3d_composite_backward_strided(mode, go.data_ptr<float>(), feats.data_ptr<float>(), fst, alphas.data_ptr<float>(),
                                    points_idx.data_ptr<int64_t>(), N, C, P, K, H, W, as, is, gf.data_ptr<float>(), fst, ga.data_ptr<float>(),
                                    stream_of(feats)),
     name);
  return {gf, ga};
}

// ---- interpolate_face_attributes ---------------------------------------------------------------------------------------------
int dtype_code(const Tensor& t) {
  TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kDouble, "barycentric_coords and face_attributes must have the same floating dtype");
  return t.scalar_type() == at::kFloat ? 0 : 1;
}

Tensor interp_face_attrs_forward(const Tensor& pix_to_face, const Tensor& barycentric_coords, const Tensor& face_attrs) {
  check_gpu({{&pix_to_face, "pix_to_face"}, {&barycentric_coords, "barycentric_coords"}, {&face_attrs, "face_attributes"}});
  TORCH_CHECK(barycentric_coords.scalar_type() == face_attrs.scalar_type(), "barycentric_coords and face_attributes must have the same floating dtype");
  const int dt = dtype_code(face_attrs);
  const int64_t P = pix_to_face.size(0);
  TORCH_CHECK(barycentric_coords.dim() == 2 && barycentric_coords.size(0) == P && barycentric_coords.size(1) == 3, "barycentric_coords must have size (P, 3)");
  TORCH_CHECK(face_attrs.dim() == 3 && face_attrs.size(1) == 3, "face_attrs must have size (F, 3, D)");
  DeviceGuard guard(face_attrs.device());
  auto p2f = pix_to_face.contiguous().to(at::kLong);
  auto bary = barycentric_coords.contiguous(), attrs = face_attrs.contiguous();
  const int64_t F = attrs.size(0), D = attrs.size(2);
  auto out = at::empty({P, D}, attrs.options());
  if (out.numel() == 0) return out;
  ok(p3d_interp_face_attrs_forward(dt, p2f.data_ptr<int64_t>(), bary.data_ptr(), attrs.data_ptr(), P, F, D, out.data_ptr(), stream_of(attrs)),
     "interp_face_attrs_forward");
  return out;
}

std::tuple<Tensor, Tensor> interp_face_attrs_backward(const Tensor& pix_to_face, const Tensor& barycentric_coords, const Tensor& face_attrs,
                                                      const Tensor& grad_pix_attrs) {
  check_gpu({{&pix_to_face, "pix_to_face"}, {&barycentric_coords, "barycentric_coords"}, {&face_attrs, "face_attributes"},
             {&grad_pix_attrs, "pix_attrs"}});
  TORCH_CHECK(barycentric_coords.scalar_type() == face_attrs.scalar_type() && grad_pix_attrs.scalar_type() == face_attrs.scalar_type(),
              "barycentric_coords, face_attributes and pix_attrs must have the same floating dtype");
  const int dt = dtype_code(face_attrs);
  at::globalContext().alertNotDeterministic("InterpFaceAttrsBackwardCuda");
  const int64_t P = pix_to_face.size(0);
  TORCH_CHECK(barycentric_coords.dim() == 2 && barycentric_coords.size(0) == P && barycentric_coords.size(1) == 3, "barycentric_coords must have size (P, 3)");
  TORCH_CHECK(face_attrs.dim() == 3 && face_attrs.size(1) == 3, "face_attrs must have size (F, 3, D)");
  const int64_t F = face_attrs.size(0), D = face_attrs.size(2);
  TORCH_CHECK(grad_pix_attrs.dim() == 2 && grad_pix_attrs.size(0) == P && grad_pix_attrs.size(1) == D, "grad_pix_attrs must have size (P, D)");
  DeviceGuard guard(face_attrs.device());
  auto p2f = pix_to_face.contiguous().to(at::kLong);
  auto bary = barycentric_coords.contiguous(), attrs = face_attrs.contiguous(), g = grad_pix_attrs.contiguous();
  auto gb = at::empty({P, 3}, attrs.options()), gf = at::empty({F, 3, D}, attrs.options());
  ok(p3d_interp_face_attrs_backward(dt, p2f.data_ptr<int64_t>(), bary.data_ptr(), attrs.data_ptr(), g.data_ptr(), P, F, D, gb.data_ptr(),
                                    gf.data_ptr(), stream_of(attrs)),
     "interp_face_attrs_backward");
  return {gb, gf};
}

// ---- sigmoid alpha blend ----------------------------------------------------------------------------------------------------
Tensor sigmoid_alpha_blend(const Tensor& distances, const Tensor& pix_to_face, double sigma) {
  check_gpu({{&distances, "distances"}, {&pix_to_face, "pix_to_face"}});
  TORCH_CHECK(distances.dim() == 4 && pix_to_face.sizes() == distances.sizes(), "distances and pix_to_face must both have shape (N, H, W, K)");
  DeviceGuard guard(distances.device());
  auto d = distances.contiguous().to(at::kFloat);
  auto p2f = pix_to_face.contiguous().to(at::kLong);
  const int64_t N = d.size(0), H = d.size(1), W = d.size(2);
  const int K = (int)d.size(3);
  auto out = at::empty({N, H, W}, d.options());
  if (out.numel() == 0) return out;
  ok(p3d_sigmoid_alpha_blend_forward(d.data_ptr<float>(), p2f.data_ptr<int64_t>(), (float)sigma, N * H * W, K, out.data_ptr<float>(), stream_of(d)),
     "sigmoid_alpha_blend");
  return out;
}

Tensor sigmoid_alpha_blend_backward(const Tensor& grad_alphas, const Tensor& alphas, const Tensor& distances, const Tensor& pix_to_face,
                                    double sigma) {
  check_gpu({{&distances, "distances"}, {&pix_to_face, "pix_to_face"}, {&alphas, "alphas"}, {&grad_alphas, "grad_alphas"}});
  DeviceGuard guard(distances.device());
  auto d = distances.contiguous().to(at::kFloat);
  auto p2f = pix_to_face.contiguous().to(at::kLong);
  auto ga = grad_alphas.contiguous().to(at::kFloat), al = alphas.contiguous().to(at::kFloat);
  const int64_t N = d.size(0), H = d.size(1), W = d.size(2);
  const int K = (int)d.size(3);
  auto out = at::empty({N, H, W, K}, d.options());
  if (out.numel() == 0) return out;
  ok(p3d_sigmoid_alpha_blend_backward(ga.data_ptr<float>(), al.data_ptr<float>(), d.data_ptr<float>(), p2f.data_ptr<int64_t>(), (float)sigma,
                                      N * H * W, K, out.data_ptr<float>(), stream_of(d)),
     "sigmoid_alpha_blend_backward");
  return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "pytorch3d._C hot-path operators over libp3d_amd.so (pybind flavour; INTEGRATION.md section B)";
  // the names and argument orders of pytorch3d/csrc/ext.cpp:38-73
  m.def("rasterize_meshes", &rasterize_meshes);
  m.def("_rasterize_meshes_naive", &rasterize_meshes_naive);
  m.def("_rasterize_meshes_coarse", &rasterize_meshes_coarse);
  m.def("_rasterize_meshes_fine", &rasterize_meshes_fine);
  m.def("rasterize_meshes_backward", &rasterize_meshes_backward);
  m.def("rasterize_points", &rasterize_points);
  m.def("_rasterize_points_naive", &rasterize_points_naive);
  m.def("_rasterize_points_coarse", &rasterize_points_coarse);
  m.def("_rasterize_points_fine", &rasterize_points_fine);
  m.def("rasterize_points_backward", &rasterize_points_backward);
  m.def("accum_alphacomposite", [](const Tensor& f, const Tensor& a, const Tensor& i) { return composite_forward(P3D_COMPOSITE_ALPHA, "accum_alphacomposite", f, a, i); });
  m.def("accum_weightedsumnorm", [](const Tensor& f, const Tensor& a, const Tensor& i) { return composite_forward(P3D_COMPOSITE_NORM_SUM, "accum_weightedsumnorm", f, a, i); });
  m.def("accum_weightedsum", [](const Tensor& f, const Tensor& a, const Tensor& i) { return composite_forward(P3D_COMPOSITE_SUM, "accum_weightedsum", f, a, i); });
  m.def("accum_alphacomposite_backward", [](const Tensor& g, const Tensor& f, const Tensor& a, const Tensor& i) { return composite_backward(P3D_COMPOSITE_ALPHA, "accum_alphacomposite_backward", g, f, a, i); });
  m.def("accum_weightedsumnorm_backward", [](const Tensor& g, const Tensor& f, const Tensor& a, const Tensor& i) { return composite_backward(P3D_COMPOSITE_NORM_SUM, "accum_weightedsumnorm_backward", g, f, a, i); });
  m.def("accum_weightedsum_backward", [](const Tensor& g, const Tensor& f, const Tensor& a, const Tensor& i) { return composite_backward(P3D_COMPOSITE_SUM, "accum_weightedsum_backward", g, f, a, i); });
  m.def("interp_face_attrs_forward", &interp_face_attrs_forward);
  m.def("interp_face_attrs_backward", &interp_face_attrs_backward);
  m.def("sigmoid_alpha_blend", &sigmoid_alpha_blend);
  m.def("sigmoid_alpha_blend_backward", &sigmoid_alpha_blend_backward);
  m.attr("EPS") = py::float_(1e-6);  // constants pytorch3d/renderer/points/pulsar/renderer.py reads at import time (ext.cpp:180-185)
  m.attr("MAX_FLOAT") = py::float_(3.4e38);
  m.attr("MAX_INT") = py::int_(2147483647);
  m.attr("MAX_UINT") = py::int_(4294967295u);
  m.attr("MAX_USHORT") = py::int_(65535);
  m.attr("PULSAR_MAX_GRAD_SPHERES") = py::int_(128);
  m.attr("__p3d_amd_flavour__") = "pybind";
}
