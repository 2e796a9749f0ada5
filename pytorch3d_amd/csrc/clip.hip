// clip.hip -- frustum culling / z-plane clipping of faces before rasterization, and the mapping of the
// rasterized fragments back to the original faces.
//
// Replaces the pure-torch pre / post step of the reference, pytorch3d/renderer/mesh/clip.py:
//   clip_faces (:324-615)  ~40 small torch kernels (nonzero / cumsum / gather / index_put ...) and 2-3 host syncs
//   convert_clipped_rasterization_to_original_faces (:618-734)  masked gather + bmm + masked scatter
// with
//   p3d_clip_faces_plan      classify every face (case 1 kept / 2 removed / 3 clipped to a triangle / 4 clipped
//                            to a quadrilateral = two triangles), then three device scans: destination index,
//                            rank among case-3 faces, rank among case-4 faces.  Totals land in the first four
//                            int64 of the plan buffer -- the caller reads them with ONE sync (the reference syncs too: the output
//                            sizes depend on the data).
//   p3d_clip_faces_emit      one thread per face writes its 0 / 1 / 2 output faces and every index table.
//   p3d_clip_faces_backward  one thread per face gathers the gradients of its output faces (no atomics).
//   p3d_convert_clipped_*    one thread per (pixel, k) sample; the backward accumulates the gradient of the
//                            3x3 conversion matrices in the wave-private LDS table of wave_table.h.
// Same arithmetic, same output order and the same autograd quirks as the reference: faces keep their
// relative order, a case-4 face becomes two consecutive faces (p4,p2,p5), (p5,p2,p3); barycentric_conversion
// rows are [case-3 rows | first triangles of case 4 | second triangles of case 4]; the interpolation weight w3
// is a constant for autograd (`w3 = w3.detach()`, clip.py:291) while w2 is not.
#include "binning.h"
#include "p3d_common.h"
#include "wave_table.h"

namespace p3d {
namespace {

struct ClipParams {
  float plane[6];  // left, right, top, bottom, znear, zfar
  int plane_mask;  // bit i: plane i is set (and culling is on)
  int has_z_clip;
  float z_clip;
  int persp;
};

__device__ __forceinline__ int classify_face(const float* v, const ClipParams& c) {
  bool culled = false;
#pragma unroll
  for (int pl = 0; pl < 6; ++pl) {
    if (c.plane_mask & (1 << pl)) {
      // clip.py:191-198 indexes `face_verts[:, axis]` on the (F, 3, 3) tensor, i.e. it tests the three
      // COORDINATES of vertex number `axis` (not coordinate `axis` of the three vertices); results must be
      // identical to the reference's, so the same quantity is tested here.
      const int axis = pl >> 1;
      int n = 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float x = v[axis * 3 + j];
        n += (pl & 1) ? (x > c.plane[pl]) : (x < c.plane[pl]);  // "<" for left / top / znear, ">" for the others
      }
      culled |= n == 3;
    }
  }
  int nclip = 0;
  if (c.has_z_clip)
    for (int i = 0; i < 3; ++i) nclip += v[i * 3 + 2] < c.z_clip;
  if (culled || nclip == 3) return 2;
  return nclip == 0 ? 1 : (nclip == 2 ? 3 : 4);
}

__global__ __launch_bounds__(256) void clip_classify_kernel(const float* __restrict__ fv, int64_t F, ClipParams c,
                                                            int* __restrict__ kase, int* __restrict__ delta,
                                                            int* __restrict__ is3, int* __restrict__ is4) {
  for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < F; f += (int64_t)gridDim.x * 256) {
    float v[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = fv[f * 9 + j];
    const int k = classify_face(v, c);
    kase[f] = k;
    delta[f] = 1 + (k == 4) - (k == 2);
    is3[f] = k == 3;
    is4[f] = k == 4;
  }
}

__global__ void clip_totals_kernel(const int64_t* __restrict__ dst, const int64_t* __restrict__ r3,
                                   const int64_t* __restrict__ r4, int64_t F, int64_t* __restrict__ totals) {
  totals[0] = dst[F];  // F_clipped
  totals[1] = r3[F];   // T3
  totals[2] = r4[F];   // T4
  totals[3] = F;
}

// The five points of clip.py:205-321 for one face.  i1 = the vertex alone on its side of the plane.
struct ClipPoints {
  float p1[3], p2[3], p3[3], p4[3], p5[3];
  float w2, w3;
  int i1, i2, i3;
};

__device__ __forceinline__ void interp(const float* pa, const float* pb, float w, bool persp, float c, float* out) {
  const float omw = 1.0f - w;
#pragma unroll
  for (int j = 0; j < 3; ++j) out[j] = pa[j] * omw + pb[j] * w;
  if (persp) {
#pragma unroll
    for (int j = 0; j < 2; ++j) out[j] = ((pa[j] * pa[2]) * omw + (pb[j] * pb[2]) * w) / c;
  }
}

__device__ __forceinline__ ClipPoints clip_points(const float* v, int kase, const ClipParams& c) {
  ClipPoints q;
  // case 3: the one vertex in FRONT of the plane; case 4: the one vertex BEHIND it
  int i1 = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const bool behind = v[i * 3 + 2] < c.z_clip;
    if (behind == (kase == 4)) i1 = i;
  }
  q.i1 = i1;
  q.i2 = (i1 + 1) % 3;
  q.i3 = (i1 + 2) % 3;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    q.p1[j] = v[q.i1 * 3 + j];
    q.p2[j] = v[q.i2 * 3 + j];
    q.p3[j] = v[q.i3 * 3 + j];
  }
  q.w2 = (q.p1[2] - c.z_clip) / (q.p1[2] - q.p2[2]);
  q.w3 = (q.p1[2] - c.z_clip) / (q.p1[2] - q.p3[2]);
  interp(q.p1, q.p2, q.w2, c.persp != 0, c.z_clip, q.p4);
  interp(q.p1, q.p3, q.w3, c.persp != 0, c.z_clip, q.p5);
  return q;
}

struct EmitArgs {
  const float* fv;
  const int* kase;
  const int64_t* dst;
  const int64_t* r3;
  const int64_t* r4;
  int64_t F, T3, T4;
  ClipParams c;
  float* out_fv;        // (Fc,3,3)
  int64_t* c2u;         // (Fc)
  float* conv;          // (T3 + 2*T4, 3, 3) or null
  int64_t* conv_idx;    // (Fc) or null
  int64_t* neighbor;    // (Fc) or null
};

__device__ __forceinline__ void put_face(float* o, const float* a, const float* b, const float* c) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    o[j] = a[j];
    o[3 + j] = b[j];
    o[6 + j] = c[j];
  }
}

// conv[row][j][k] = weight of ORIGINAL vertex j in clipped vertex k (clip.py:505, 557-558: stack(..., 2))
__device__ __forceinline__ void put_conv(float* m, int k, int ja, float wa, int jb, float wb) {
#pragma unroll
  for (int j = 0; j < 3; ++j) m[j * 3 + k] = 0.0f;
  m[ja * 3 + k] = wa;
  if (jb >= 0) m[jb * 3 + k] = wb;
}

__global__ __launch_bounds__(256) void clip_emit_kernel(EmitArgs a) {
  for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < a.F; f += (int64_t)gridDim.x * 256) {
    const int k = a.kase[f];
    if (k == 2) continue;
    const int64_t d = a.dst[f];
    float v[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = a.fv[f * 9 + j];
    if (k == 1) {
#pragma unroll
      for (int j = 0; j < 9; ++j) a.out_fv[d * 9 + j] = v[j];
      a.c2u[d] = f;
      if (a.conv_idx) a.conv_idx[d] = -1;
      if (a.neighbor) a.neighbor[d] = -1;
      continue;
    }
    const ClipPoints q = clip_points(v, k, a.c);
    if (k == 3) {
      put_face(a.out_fv + d * 9, q.p4, q.p5, q.p1);
      a.c2u[d] = f;
      const int64_t row = a.r3[f];
      float* m = a.conv + row * 9;
      put_conv(m, 0, q.i1, 1.0f - q.w2, q.i2, q.w2);  // p4
      put_conv(m, 1, q.i1, 1.0f - q.w3, q.i3, q.w3);  // p5
      put_conv(m, 2, q.i1, 1.0f, -1, 0.0f);           // p1
      a.conv_idx[d] = row;
      a.neighbor[d] = -1;
    } else {
      put_face(a.out_fv + d * 9, q.p4, q.p2, q.p5);        // t1
      put_face(a.out_fv + (d + 1) * 9, q.p5, q.p2, q.p3);  // t2
      a.c2u[d] = f;
      a.c2u[d + 1] = f;
      const int64_t row1 = a.T3 + a.r4[f], row2 = a.T3 + a.T4 + a.r4[f];
      float* m1 = a.conv + row1 * 9;
      put_conv(m1, 0, q.i1, 1.0f - q.w2, q.i2, q.w2);  // p4
      put_conv(m1, 1, q.i2, 1.0f, -1, 0.0f);           // p2
      put_conv(m1, 2, q.i1, 1.0f - q.w3, q.i3, q.w3);  // p5
      float* m2 = a.conv + row2 * 9;
      put_conv(m2, 0, q.i1, 1.0f - q.w3, q.i3, q.w3);  // p5
      put_conv(m2, 1, q.i2, 1.0f, -1, 0.0f);           // p2
      put_conv(m2, 2, q.i3, 1.0f, -1, 0.0f);           // p3
      a.conv_idx[d] = row1;
      a.conv_idx[d + 1] = row2;
      a.neighbor[d] = d + 1;
      a.neighbor[d + 1] = d;
    }
  }
}

__global__ __launch_bounds__(256) void clip_mesh_index_kernel(const int64_t* __restrict__ dst,
                                                              const int64_t* __restrict__ mesh_first, int N, int64_t F,
                                                              int64_t Fc, int64_t* __restrict__ first_c,
                                                              int64_t* __restrict__ count_c) {
  // first_clipped[n] = dst[first[n]]; count_clipped[n] = first_clipped[n + 1] - first_clipped[n] (clip.py:446-451)
  for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
    int64_t a = mesh_first[n];
    a = a < 0 ? 0 : (a > F ? F : a);
    int64_t b = Fc;
    if (n + 1 < N) {
      int64_t t = mesh_first[n + 1];
      t = t < 0 ? 0 : (t > F ? F : t);
      b = dst[t];
    }
    first_c[n] = dst[a];
    count_c[n] = b - dst[a];
  }
}

// gradient of out = interp(pa, pb, w): adds to g_pa, g_pb (3 each) and returns d out / d w contracted with g
__device__ __forceinline__ float interp_bwd(const float* pa, const float* pb, float w, bool persp, float c,
                                            const float* g, float* g_pa, float* g_pb) {
  const float omw = 1.0f - w;
  float gw = 0.0f;
  if (!persp) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      g_pa[j] += g[j] * omw;
      g_pb[j] += g[j] * w;
      gw += g[j] * (pb[j] - pa[j]);
    }
  } else {
    g_pa[2] += g[2] * omw;
    g_pb[2] += g[2] * w;
    gw += g[2] * (pb[2] - pa[2]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float gj = g[j] / c;
      g_pa[j] += gj * pa[2] * omw;
      g_pa[2] += gj * pa[j] * omw;
      g_pb[j] += gj * pb[2] * w;
      g_pb[2] += gj * pb[j] * w;
      gw += gj * (pb[j] * pb[2] - pa[j] * pa[2]);
    }
  }
  return gw;
}

struct ClipBwdArgs {
  const float* fv;
  const int* kase;
  const int64_t* dst;
  const int64_t* r3;
  const int64_t* r4;
  int64_t F, T3, T4;
  ClipParams c;
  const float* g_out;   // (Fc,3,3)
  const float* g_conv;  // (T,3,3) or null
  float* g_fv;          // (F,3,3)
};

__global__ __launch_bounds__(256) void clip_backward_kernel(ClipBwdArgs a) {
  for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < a.F; f += (int64_t)gridDim.x * 256) {
    const int k = a.kase[f];
    float gv[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) gv[j] = 0.0f;
    if (k == 1) {
      const int64_t d = a.dst[f];
#pragma unroll
      for (int j = 0; j < 9; ++j) gv[j] = a.g_out[d * 9 + j];
    } else if (k == 3 || k == 4) {
      const int64_t d = a.dst[f];
      float v[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) v[j] = a.fv[f * 9 + j];
      const ClipPoints q = clip_points(v, k, a.c);
      float g1[3] = {0, 0, 0}, g2[3] = {0, 0, 0}, g3[3] = {0, 0, 0}, g4[3], g5[3];
      const float* go = a.g_out + d * 9;
      int64_t row4;  // conversion row whose column 0 is p4's barycentric
      if (k == 3) {  // (p4, p5, p1)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          g4[j] = go[j];
          g5[j] = go[3 + j];
          g1[j] = go[6 + j];
        }
        row4 = a.r3[f];
      } else {  // (p4, p2, p5), (p5, p2, p3)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          g4[j] = go[j];
          g2[j] = go[3 + j] + go[9 + 3 + j];
          g5[j] = go[6 + j] + go[9 + j];
          g3[j] = go[9 + 6 + j];
        }
        row4 = a.T3 + a.r4[f];
      }
      const bool persp = a.c.persp != 0;
      float gw2 = interp_bwd(q.p1, q.p2, q.w2, persp, a.c.z_clip, g4, g1, g2);
      (void)interp_bwd(q.p1, q.p3, q.w3, persp, a.c.z_clip, g5, g1, g3);  // w3 is detached: its d/dw is dropped
      if (a.g_conv) {
        // p4's barycentric column: (1 - w2) at original vertex i1, w2 at i2
        const float* gm = a.g_conv + row4 * 9;
        gw2 += gm[q.i2 * 3 + 0] - gm[q.i1 * 3 + 0];
      }
      // w2 = (p1.z - c) / (p1.z - p2.z)
      const float den = q.p1[2] - q.p2[2];
      const float inv2 = 1.0f / (den * den);
      g1[2] += gw2 * (a.c.z_clip - q.p2[2]) * inv2;
      g2[2] += gw2 * (q.p1[2] - a.c.z_clip) * inv2;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        gv[q.i1 * 3 + j] = g1[j];
        gv[q.i2 * 3 + j] = g2[j];
        gv[q.i3 * 3 + j] = g3[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) a.g_fv[f * 9 + j] = gv[j];
  }
}

// ---- fragments of the clipped faces -> fragments of the original faces ------------------------------------
__global__ __launch_bounds__(256) void convert_fwd_kernel(const int64_t* __restrict__ p2f_c,
                                                          const float* __restrict__ bary_c,
                                                          const int64_t* __restrict__ c2u,
                                                          const float* __restrict__ conv,
                                                          const int64_t* __restrict__ conv_idx, int64_t S,
                                                          int64_t* __restrict__ p2f_u, float* __restrict__ bary_u) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < S; i += (int64_t)gridDim.x * 256) {
    const int64_t f = p2f_c[i];
    float b0 = bary_c[i * 3], b1 = bary_c[i * 3 + 1], b2 = bary_c[i * 3 + 2];
    int64_t fu = -1;
    if (f != -1) {
      fu = c2u[f];
      const int64_t ci = conv_idx ? conv_idx[f] : -1;
      if (ci != -1) {
        const float* m = conv + ci * 9;
        const float u0 = m[0] * b0 + m[1] * b1 + m[2] * b2;
        const float u1 = m[3] * b0 + m[4] * b1 + m[5] * b2;
        const float u2 = m[6] * b0 + m[7] * b1 + m[8] * b2;
        b0 = u0;
        b1 = u1;
        b2 = u2;
      }
    }
    p2f_u[i] = fu;
    bary_u[i * 3] = b0;
    bary_u[i * 3 + 1] = b1;
    bary_u[i * 3 + 2] = b2;
  }
}

using ConvTable = WaveTable<9, 182>;

__global__ __launch_bounds__(256) void convert_bwd_kernel(const int64_t* __restrict__ p2f_c,
                                                          const float* __restrict__ bary_c,
                                                          const float* __restrict__ conv,
                                                          const int64_t* __restrict__ conv_idx,
                                                          const float* __restrict__ g_u, int64_t S, int64_t span,
                                                          float* __restrict__ g_bary_c, float* __restrict__ g_conv) {
  __shared__ __align__(16) int s_table[4][ConvTable::kLdsInts];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t begin = ((int64_t)blockIdx.x * 4 + w) * span;
  if (begin >= S) return;  // wave-uniform; no workgroup barrier in this kernel
  const int64_t end = begin + span < S ? begin + span : S;
  ConvTable tab;
  tab.init(s_table[w], lane);
  for (int64_t base = begin; base < end; base += 64) {
    const int64_t i = base + lane;
    int key = -1;
    float g[9];
    if (i < end) {
      const int64_t f = p2f_c[i];
      float g0 = g_u[i * 3], g1 = g_u[i * 3 + 1], g2 = g_u[i * 3 + 2];
      const int64_t ci = (f != -1 && conv_idx) ? conv_idx[f] : -1;
      if (ci != -1) {
        const float* m = conv + ci * 9;
        const float b[3] = {bary_c[i * 3], bary_c[i * 3 + 1], bary_c[i * 3 + 2]};
        const float gu[3] = {g0, g1, g2};
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int k = 0; k < 3; ++k) g[j * 3 + k] = gu[j] * b[k];
        g0 = m[0] * gu[0] + m[3] * gu[1] + m[6] * gu[2];
        g1 = m[1] * gu[0] + m[4] * gu[1] + m[7] * gu[2];
        g2 = m[2] * gu[0] + m[5] * gu[1] + m[8] * gu[2];
        key = (int)ci;
      }
      g_bary_c[i * 3] = g0;
      g_bary_c[i * 3 + 1] = g1;
      g_bary_c[i * 3 + 2] = g2;
    }
    if (g_conv == nullptr || __ballot(key >= 0) == 0) continue;  // wave-uniform
    tab.add(g_conv, lane, key, g);
  }
  if (g_conv && tab.used > 0) tab.flush(g_conv, lane);
}

unsigned grid_for(int64_t n) {
  int64_t b = ceil_div(n, 256);
  if (b > 256 * 32) b = 256 * 32;
  return (unsigned)(b < 1 ? 1 : b);
}

ClipParams make_params(const float planes[6], int plane_mask, int cull, int has_z_clip, float z_clip, int persp) {
  ClipParams c;
  for (int i = 0; i < 6; ++i) c.plane[i] = planes ? planes[i] : 0.0f;
  c.plane_mask = cull ? plane_mask : 0;
  c.has_z_clip = has_z_clip;
  c.z_clip = z_clip;
  c.persp = persp;
  return c;
}

// plan buffer layout (caller-allocated, p3d_clip_faces_plan_bytes): case int[F] | tmp int[3][F] | dst, r3, r4
// int64[F + 1] each | blocksum | totals int64[4]
struct PlanView {
  int* kase;
  int* tmp[3];
  int64_t* dst;
  int64_t* r3;
  int64_t* r4;
  long long* blocksum;
  int64_t* totals;
};

bool carve_plan(void* buf, size_t bytes, int64_t F, PlanView* v, size_t* need = nullptr) {
  Arena ar(buf, bytes);
  v->totals = ar.take<int64_t>(4);  // first: the caller reads plan[0..32) as four int64
  v->kase = ar.take<int>((size_t)F + 1);
  for (int i = 0; i < 3; ++i) v->tmp[i] = ar.take<int>((size_t)F + 1);
  v->dst = ar.take<int64_t>((size_t)F + 1);
  v->r3 = ar.take<int64_t>((size_t)F + 1);
  v->r4 = ar.take<int64_t>((size_t)F + 1);
  v->blocksum = ar.take<long long>((size_t)ceil_div(F, 1024) + 2);
  if (need) *need = ar.off;
  return ar.ok();
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API size_t p3d_clip_faces_plan_bytes(int64_t F) {
  if (F < 0) return 0;
  PlanView v;
  size_t need = 0;
  carve_plan(nullptr, 0, F, &v, &need);
  return need + 256;
}

P3D_API int p3d_clip_faces_plan(const float* face_verts, int64_t F, const float planes[6], int plane_mask, int cull,
                                int has_z_clip, float z_clip_value, void* plan, size_t plan_bytes,
                                p3d_stream_t stream) {
  if (F < 0) return P3D_ERR_INVALID_ARG;
  PlanView v;
  if (!plan || !carve_plan(plan, plan_bytes, F, &v)) return P3D_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (F == 0) {
    if (hipMemsetAsync(v.dst, 0, sizeof(int64_t), s) != hipSuccess) return P3D_ERR_LAUNCH;
    if (hipMemsetAsync(v.r3, 0, sizeof(int64_t), s) != hipSuccess) return P3D_ERR_LAUNCH;
    if (hipMemsetAsync(v.r4, 0, sizeof(int64_t), s) != hipSuccess) return P3D_ERR_LAUNCH;
    if (hipMemsetAsync(v.totals, 0, 4 * sizeof(int64_t), s) != hipSuccess) return P3D_ERR_LAUNCH;
    return P3D_OK;
  }
  if (!face_verts) return P3D_ERR_INVALID_ARG;
  const ClipParams c = make_params(planes, plane_mask, cull, has_z_clip, z_clip_value, 0);
  LaunchScope ls("clip_plan", s);
  clip_classify_kernel<<<grid_for(F), 256, 0, s>>>(face_verts, F, c, v.kase, v.tmp[0], v.tmp[1], v.tmp[2]);
  int rc = exclusive_scan_i32(v.tmp[0], F, v.blocksum, v.dst, s);
  if (rc == P3D_OK) rc = exclusive_scan_i32(v.tmp[1], F, v.blocksum, v.r3, s);
  if (rc == P3D_OK) rc = exclusive_scan_i32(v.tmp[2], F, v.blocksum, v.r4, s);
  if (rc != P3D_OK) return rc;
  clip_totals_kernel<<<1, 1, 0, s>>>(v.dst, v.r3, v.r4, F, v.totals);
  return launch_status();
}

P3D_API int p3d_clip_faces_emit(const float* face_verts, int64_t F, const int64_t* mesh_to_face_first_idx, int N,
                                const void* plan, size_t plan_bytes, int64_t F_clipped, int64_t T3, int64_t T4,
                                float z_clip_value, int perspective_correct, float* face_verts_clipped,
                                int64_t* mesh_to_face_first_idx_clipped, int64_t* num_faces_per_mesh_clipped,
                                int64_t* faces_clipped_to_unclipped_idx, float* barycentric_conversion,
                                int64_t* faces_clipped_to_conversion_idx, int64_t* clipped_faces_neighbor_idx,
                                p3d_stream_t stream) {
  if (F < 0 || N < 0 || F_clipped < 0 || T3 < 0 || T4 < 0) return P3D_ERR_INVALID_ARG;
  PlanView v;
  if (!plan || !carve_plan(const_cast<void*>(plan), plan_bytes, F, &v)) return P3D_ERR_WORKSPACE;
  const bool clipped = T3 + T4 > 0;
  if (clipped && (!barycentric_conversion || !faces_clipped_to_conversion_idx || !clipped_faces_neighbor_idx))
    return P3D_ERR_INVALID_ARG;
  if (F_clipped > 0 && (!face_verts_clipped || !faces_clipped_to_unclipped_idx)) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls("clip_emit", s);
  if (N > 0) {
    if (!mesh_to_face_first_idx || !mesh_to_face_first_idx_clipped || !num_faces_per_mesh_clipped)
      return P3D_ERR_INVALID_ARG;
    clip_mesh_index_kernel<<<grid_for(N), 256, 0, s>>>(v.dst, mesh_to_face_first_idx, N, F, F_clipped,
                                                        mesh_to_face_first_idx_clipped, num_faces_per_mesh_clipped);
  }
  if (F > 0) {
    EmitArgs a;
    a.fv = face_verts;
    a.kase = v.kase;
    a.dst = v.dst;
    a.r3 = v.r3;
    a.r4 = v.r4;
    a.F = F;
    a.T3 = T3;
    a.T4 = T4;
    a.c = make_params(nullptr, 0, 0, 1, z_clip_value, perspective_correct);
    a.out_fv = face_verts_clipped;
    a.c2u = faces_clipped_to_unclipped_idx;
    a.conv = clipped ? barycentric_conversion : nullptr;
    a.conv_idx = clipped ? faces_clipped_to_conversion_idx : nullptr;
    a.neighbor = clipped ? clipped_faces_neighbor_idx : nullptr;
    clip_emit_kernel<<<grid_for(F), 256, 0, s>>>(a);
  }
  return launch_status();
}

P3D_API int p3d_clip_faces_backward(const float* face_verts, int64_t F, const void* plan, size_t plan_bytes, int64_t T3,
                                    int64_t T4, float z_clip_value, int perspective_correct,
                                    const float* grad_face_verts_clipped, const float* grad_barycentric_conversion,
                                    float* grad_face_verts, p3d_stream_t stream) {
  if (F < 0) return P3D_ERR_INVALID_ARG;
  if (F == 0) return P3D_OK;
  PlanView v;
  if (!plan || !carve_plan(const_cast<void*>(plan), plan_bytes, F, &v)) return P3D_ERR_WORKSPACE;
  if (!face_verts || !grad_face_verts_clipped || !grad_face_verts) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  ClipBwdArgs a;
  a.fv = face_verts;
  a.kase = v.kase;
  a.dst = v.dst;
  a.r3 = v.r3;
  a.r4 = v.r4;
  a.F = F;
  a.T3 = T3;
  a.T4 = T4;
  a.c = make_params(nullptr, 0, 0, 1, z_clip_value, perspective_correct);
  a.g_out = grad_face_verts_clipped;
  a.g_conv = grad_barycentric_conversion;
  a.g_fv = grad_face_verts;
  LaunchScope ls("clip_backward", s);
  clip_backward_kernel<<<grid_for(F), 256, 0, s>>>(a);
  return launch_status();
}

P3D_API int p3d_convert_clipped_forward(const int64_t* pix_to_face_clipped, const float* bary_coords_clipped,
                                        const int64_t* faces_clipped_to_unclipped_idx,
                                        const float* barycentric_conversion,
                                        const int64_t* faces_clipped_to_conversion_idx, int64_t num_samples,
                                        int64_t* pix_to_face_unclipped, float* bary_coords_unclipped,
                                        p3d_stream_t stream) {
  if (num_samples < 0) return P3D_ERR_INVALID_ARG;
  if (num_samples == 0) return P3D_OK;
  if (!pix_to_face_clipped || !bary_coords_clipped || !faces_clipped_to_unclipped_idx || !pix_to_face_unclipped ||
      !bary_coords_unclipped)
    return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  LaunchScope ls("convert_clipped_fwd", s);
  convert_fwd_kernel<<<grid_for(num_samples), 256, 0, s>>>(pix_to_face_clipped, bary_coords_clipped,
                                                          faces_clipped_to_unclipped_idx, barycentric_conversion,
                                                          barycentric_conversion ? faces_clipped_to_conversion_idx
                                                                                 : nullptr,
                                                          num_samples, pix_to_face_unclipped, bary_coords_unclipped);
  return launch_status();
}

P3D_API int p3d_convert_clipped_backward(const int64_t* pix_to_face_clipped, const float* bary_coords_clipped,
                                         const float* barycentric_conversion,
                                         const int64_t* faces_clipped_to_conversion_idx,
                                         const float* grad_bary_unclipped, int64_t num_samples, int64_t T,
                                         float* grad_bary_clipped, float* grad_barycentric_conversion,
                                         p3d_stream_t stream) {
  if (num_samples < 0 || T < 0) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (grad_barycentric_conversion && T > 0 &&
      hipMemsetAsync(grad_barycentric_conversion, 0, (size_t)T * 9 * sizeof(float), s) != hipSuccess)
    return P3D_ERR_LAUNCH;
  if (num_samples == 0) return P3D_OK;
  if (!pix_to_face_clipped || !bary_coords_clipped || !grad_bary_unclipped || !grad_bary_clipped)
    return P3D_ERR_INVALID_ARG;
  int64_t waves = ceil_div(num_samples, 4096);
  if (waves > 4 * 8192) waves = 4 * 8192;
  const int64_t blocks = ceil_div(waves, 4);
  const int64_t span = ceil_div(ceil_div(num_samples, blocks * 4), 64) * 64;
  LaunchScope ls("convert_clipped_bwd", s);
  convert_bwd_kernel<<<(unsigned)blocks, 256, 0, s>>>(pix_to_face_clipped, bary_coords_clipped, barycentric_conversion,
                                                     barycentric_conversion ? faces_clipped_to_conversion_idx : nullptr,
                                                     grad_bary_unclipped, num_samples, span, grad_bary_clipped,
                                                     T > 0 ? grad_barycentric_conversion : nullptr);
  return launch_status();
}
