// transform.hip -- world -> NDC vertex transform fused into the passes either side of the rasterizer.
//
// The reference's MeshRasterizer.transform (pytorch3d/renderer/mesh/rasterizer.py:171-216) runs, as torch ops on the
// padded (N, Vmax, 3) vertices: verts_view = world_to_view.transform_points(verts_world); verts_ndc = (projection o
// to_ndc).transform_points(verts_view); verts_ndc.z = verts_view.z -- two batched 4x4 products with homogeneous
// divides (transforms/transform3d.py:325-360), a slice assignment, padded <-> packed conversions, and the autograd
// twins of all of them.  SURVEY.md 8(f) row 3: here the transform happens INSIDE the face gather
// (p3d_transform_gather_face_verts: world vertices + faces + two 4x4 matrices per mesh -> NDC face_verts (F,3,3), one
// launch, no NDC vertex tensor, no padded layout), and its backward is one per-vertex kernel applied to the NDC vertex
// gradient that p3d_rasterize_meshes_backward_verts has already reduced per vertex.
//
// Matrices follow the reference's row-vector convention: out_j = sum_i in_i * M[i][j] with in = (x, y, z, 1), then
// xyz / w.  matrices: (N, 2, 4, 4) f32 row-major: [n][0] world -> view, [n][1] view -> NDC (projection composed with
// the NDC conversion).  Arithmetic is plain float (tolerance-gated against the reference's bmm: 1e-5 on NDC).
#include "p3d_common.h"

namespace p3d {
namespace {

struct Mat4 {
  float m[16];
};

__device__ __forceinline__ void load_mats(const float* __restrict__ mats, int n, Mat4* A, Mat4* B) {
  const float4* s = reinterpret_cast<const float4*>(mats + (int64_t)n * 32);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 a = s[i], b = s[4 + i];
    A->m[4 * i] = a.x;
    A->m[4 * i + 1] = a.y;
    A->m[4 * i + 2] = a.z;
    A->m[4 * i + 3] = a.w;
    B->m[4 * i] = b.x;
    B->m[4 * i + 1] = b.y;
    B->m[4 * i + 2] = b.z;
    B->m[4 * i + 3] = b.w;
  }
}

// (x, y, z, 1) @ M
__device__ __forceinline__ void mul4(const Mat4& M, float x, float y, float z, float (&o)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = x * M.m[j] + y * M.m[4 + j] + z * M.m[8 + j] + M.m[12 + j];
}

struct NdcPoint {
  float x, y, z;          // NDC x, y and view-space depth
  float vh[4], nh[4];     // homogeneous view / NDC coordinates (for the backward)
};

__device__ __forceinline__ NdcPoint to_ndc(const Mat4& A, const Mat4& B, float px, float py, float pz) {
  NdcPoint r;
  mul4(A, px, py, pz, r.vh);
  const float vx = r.vh[0] / r.vh[3], vy = r.vh[1] / r.vh[3], vz = r.vh[2] / r.vh[3];
  mul4(B, vx, vy, vz, r.nh);
  r.x = r.nh[0] / r.nh[3];
  r.y = r.nh[1] / r.nh[3];
  r.z = vz;
  return r;
}

// index of the segment that holds `i`: the last n with first[n] <= i (first ascending, first[0] == 0)
__device__ __forceinline__ int segment_of(const int64_t* __restrict__ first, int N, int64_t i) {
  int lo = 0, hi = N - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= i)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void transform_gather_kernel(const float* __restrict__ verts, const int64_t* __restrict__ faces,
                                                               const int64_t* __restrict__ face_first, const float* __restrict__ mats,
                                                               int64_t V, int64_t n_corners, int N, int mats_n,
                                                               float* __restrict__ face_verts) {
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n_corners; c += (int64_t)gridDim.x * 256) {
    const int n = segment_of(face_first, N, c / 3);
    Mat4 A, B;
    load_mats(mats, mats_n == 1 ? 0 : n, &A, &B);
    int64_t v = faces[c];
    if (v < 0) v += V;
    const bool ok = v >= 0 && v < V;
    const float* s = verts + (ok ? v : 0) * 3;
    const NdcPoint p = to_ndc(A, B, s[0], s[1], s[2]);
    const float nan = __int_as_float(0x7fc00000);
    float* d = face_verts + c * 3;
    d[0] = ok ? p.x : nan;
    d[1] = ok ? p.y : nan;
    d[2] = ok ? p.z : nan;
  }
}

__global__ __launch_bounds__(256) void transform_verts_kernel(const float* __restrict__ verts, const int64_t* __restrict__ vert_first,
                                                              const float* __restrict__ mats, int64_t V, int N, int mats_n,
                                                              float* __restrict__ verts_ndc) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    const int n = segment_of(vert_first, N, v);
    Mat4 A, B;
    load_mats(mats, mats_n == 1 ? 0 : n, &A, &B);
    const NdcPoint p = to_ndc(A, B, verts[v * 3], verts[v * 3 + 1], verts[v * 3 + 2]);
    verts_ndc[v * 3] = p.x;
    verts_ndc[v * 3 + 1] = p.y;
    verts_ndc[v * 3 + 2] = p.z;
  }
}

// grad_world = J^T grad_ndc for out = (X/W, Y/W, vz) with (X, Y, ., W) = (view, 1) @ B, view = ((p, 1) @ A).xyz / w
__global__ __launch_bounds__(256) void transform_verts_backward_kernel(const float* __restrict__ verts,
                                                                       const int64_t* __restrict__ vert_first,
                                                                       const float* __restrict__ mats,
                                                                       const float* __restrict__ grad_ndc, int64_t V, int N,
                                                                       int mats_n, float* __restrict__ grad_world) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) {
    const int n = segment_of(vert_first, N, v);
    Mat4 A, B;
    load_mats(mats, mats_n == 1 ? 0 : n, &A, &B);
    const NdcPoint p = to_ndc(A, B, verts[v * 3], verts[v * 3 + 1], verts[v * 3 + 2]);
    const float gx = grad_ndc[v * 3], gy = grad_ndc[v * 3 + 1], gz = grad_ndc[v * 3 + 2];
    const float iw = 1.0f / p.nh[3];
    // gradient wrt the homogeneous NDC coordinates (its z component has no consumer: the depth comes from the view)
    const float gn[4] = {gx * iw, gy * iw, 0.0f, -(gx * p.nh[0] + gy * p.nh[1]) * iw * iw};
    float gv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) gv[i] = gn[0] * B.m[4 * i] + gn[1] * B.m[4 * i + 1] + gn[2] * B.m[4 * i + 2] + gn[3] * B.m[4 * i + 3];
    gv[2] += gz;
    const float ia = 1.0f / p.vh[3];
    const float gh[4] = {gv[0] * ia, gv[1] * ia, gv[2] * ia, -(gv[0] * p.vh[0] + gv[1] * p.vh[1] + gv[2] * p.vh[2]) * ia * ia};
#pragma unroll
    for (int i = 0; i < 3; ++i)
      grad_world[v * 3 + i] = gh[0] * A.m[4 * i] + gh[1] * A.m[4 * i + 1] + gh[2] * A.m[4 * i + 2] + gh[3] * A.m[4 * i + 3];
  }
}

inline unsigned blocks_for(int64_t n) {
  int64_t b = ceil_div(n, 256);
  if (b > 256 * 16) b = 256 * 16;
  return (unsigned)(b > 0 ? b : 1);
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API int p3d_transform_gather_face_verts(const float* verts_world, const int64_t* faces, const int64_t* mesh_to_face_first_idx,
                                            const float* matrices, int64_t V, int64_t F, int N, int num_matrices,
                                            float* face_verts, p3d_stream_t stream) {
  if (V < 0 || F < 0 || N < 0 || (num_matrices != 1 && num_matrices != N)) return P3D_ERR_INVALID_ARG;
  if (F == 0) return P3D_OK;
  if (!verts_world || !faces || !mesh_to_face_first_idx || !matrices || !face_verts || N == 0) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls("transform_gather_face_verts", s);
  transform_gather_kernel<<<blocks_for(F * 3), 256, 0, s>>>(verts_world, faces, mesh_to_face_first_idx, matrices, V, F * 3, N,
                                                          num_matrices, face_verts);
  return launch_status();
}

P3D_API int p3d_transform_verts_forward(const float* verts_world, const int64_t* mesh_to_vert_first_idx, const float* matrices,
                                        int64_t V, int N, int num_matrices, float* verts_ndc, p3d_stream_t stream) {
  if (V < 0 || N < 0 || (num_matrices != 1 && num_matrices != N)) return P3D_ERR_INVALID_ARG;
  if (V == 0) return P3D_OK;
  if (!verts_world || !mesh_to_vert_first_idx || !matrices || !verts_ndc || N == 0) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls("transform_verts", s);
  transform_verts_kernel<<<blocks_for(V), 256, 0, s>>>(verts_world, mesh_to_vert_first_idx, matrices, V, N, num_matrices, verts_ndc);
  return launch_status();
}

P3D_API int p3d_transform_verts_backward(const float* verts_world, const int64_t* mesh_to_vert_first_idx, const float* matrices,
                                         const float* grad_verts_ndc, int64_t V, int N, int num_matrices,
                                         float* grad_verts_world, p3d_stream_t stream) {
  if (V < 0 || N < 0 || (num_matrices != 1 && num_matrices != N)) return P3D_ERR_INVALID_ARG;
  if (V == 0) return P3D_OK;
  if (!verts_world || !mesh_to_vert_first_idx || !matrices || !grad_verts_ndc || !grad_verts_world || N == 0) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls("transform_verts_bwd", s);
  transform_verts_backward_kernel<<<blocks_for(V), 256, 0, s>>>(verts_world, mesh_to_vert_first_idx, matrices, grad_verts_ndc, V, N,
                                                                 num_matrices, grad_verts_world);
  return launch_status();
}
