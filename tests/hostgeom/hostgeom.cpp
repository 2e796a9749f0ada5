// hostgeom.cpp -- TEST HARNESS (never shipped, never imported by pytorch3d_amd).
//
// Compiles the DEVICE headers pytorch3d_amd/csrc/p3d_geom.h and topk.h for the host with g++
// and wraps them in a brute-force per-pixel loop, so the arithmetic contract and the queue
// logic the gfx950 kernels use can be compared with oracle/ on a machine without a GPU.
// What this cannot cover (binning, LDS staging, culling, stores) is covered by the -m gpu tests.
#include <stdint.h>
#include <string.h>

#include "../../pytorch3d_amd/csrc/p3d_geom.h"
#include "../../pytorch3d_amd/csrc/topk.h"

using namespace p3d;

template <typename Queue>
static void raster_pixel(const float* fv, const int64_t* nbr, int64_t f0, int64_t f1, f2 p, float blur, float sqrt_blur,
                         int K, bool persp, bool clip, bool cull, Queue& q) {
  q.init();
  for (int64_t f = f0; f < f1; ++f) {
    const float* g = fv + f * 9;
    const f3 v0 = mk3(g[0], g[1], g[2]), v1 = mk3(g[3], g[4], g[5]), v2 = mk3(g[6], g[7], g[8]);
    const FaceSetup fs = face_setup(v0, v1, v2, sqrt_blur, cull);
    if (fs.reject || outside_box(fs, p)) continue;
    FaceHit h;
    if (!face_hit(v0, v1, v2, p, blur, persp, clip, &h)) continue;
    const float pl[4] = {h.dist, h.bary.x, h.bary.y, h.bary.z};
    const int nb = (int)nbr[f];
    if (nb != -1) {
      const int at = q.find(nb);
      if (at >= 0) {
        if (fabsf(h.dist) < fabsf(q.payload_at(0, at))) {
          q.erase(at);
          q.insert(K, h.z, (int)f, pl);
        }
      } else {
        q.insert(K, h.z, (int)f, pl);
      }
    } else {
      q.insert(K, h.z, (int)f, pl);
    }
  }
}

template <typename Queue>
static void emit(const Queue& q, int K, int64_t o, int64_t* p2f, float* zbuf, float* bary, float* dists) {
  for (int k = 0; k < K; ++k) {
    const bool ok = q.valid(k);
    p2f[o + k] = ok ? q.idx[k] : -1;
    zbuf[o + k] = ok ? q.z[k] : -1.0f;
    dists[o + k] = ok ? q.pl[0][k] : -1.0f;
    for (int j = 0; j < 3; ++j) bary[(o + k) * 3 + j] = ok ? q.pl[1 + j][k] : -1.0f;
  }
}

extern "C" int hg_rasterize_meshes(const float* fv, const int64_t* first, const int64_t* count, const int64_t* nbr,
                                   int N, int H, int W, float blur, int K, int persp, int clip, int cull, int use_mem,
                                   int64_t* p2f, float* zbuf, float* bary, float* dists) {
  const float sqrt_blur = sqrtf(blur);
  for (int n = 0; n < N; ++n)
    for (int yo = 0; yo < H; ++yo)
      for (int xo = 0; xo < W; ++xo) {
        const int yi = H - 1 - yo, xi = W - 1 - xo;
        const f2 p = mk2(pix_to_ndc(xi, W, H), pix_to_ndc(yi, H, W));
        const int64_t o = (((int64_t)n * H + yo) * W + xo) * K;
        const int64_t f0 = first[n], f1 = first[n] + count[n];
        if (!use_mem && K <= 8) {
          TopKReg<8, 4> q;
          raster_pixel(fv, nbr, f0, f1, p, blur, sqrt_blur, K, persp, clip, cull, q);
          emit(q, K, o, p2f, zbuf, bary, dists);
        } else {
          TopKMem<150, 4> q;
          raster_pixel(fv, nbr, f0, f1, p, blur, sqrt_blur, K, persp, clip, cull, q);
          emit(q, K, o, p2f, zbuf, bary, dists);
        }
      }
  return 0;
}

extern "C" int hg_rasterize_meshes_backward(const float* fv, const int64_t* p2f, const float* gz, const float* gb,
                                            const float* gd, int64_t F, int N, int H, int W, int K, int persp, int clip,
                                            int clip_on_corrected, double* acc /* F*9, zeroed by caller */) {
  (void)F;
  for (int n = 0; n < N; ++n)
    for (int yo = 0; yo < H; ++yo)
      for (int xo = 0; xo < W; ++xo) {
        const f2 p = mk2(pix_to_ndc(W - 1 - xo, W, H), pix_to_ndc(H - 1 - yo, H, W));
        for (int k = 0; k < K; ++k) {
          const int64_t i = (((int64_t)n * H + yo) * W + xo) * K + k;
          const int64_t f = p2f[i];
          if (f < 0) continue;
          const float* g = fv + f * 9;
          const FaceGrad r = face_sample_bwd(mk3(g[0], g[1], g[2]), mk3(g[3], g[4], g[5]), mk3(g[6], g[7], g[8]), p,
                                             gz[i], mk3(gb[i * 3], gb[i * 3 + 1], gb[i * 3 + 2]), gd[i], persp, clip,
                                             clip_on_corrected);
          for (int j = 0; j < 9; ++j) acc[f * 9 + j] += (double)r.g[j];
        }
      }
  return 0;
}

// bin rectangle of a bbox through the monotone edge tables (what binning.hip does per primitive)
extern "C" void hg_bin_rect(float xmin, float xmax, float ymin, float ymax, int H, int W, int bin_size, int* out4) {
  const int BH = 1 + (H - 1) / bin_size, BW = 1 + (W - 1) / bin_size;
  int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
  for (int b = 0; b < BW; ++b) {
    x0 += (xmin <= bin_hi(b, bin_size, W, H)) ? 0 : 1;
    x1 += (bin_lo(b, bin_size, W, H) < xmax) ? 1 : 0;
  }
  for (int b = 0; b < BH; ++b) {
    y0 += (ymin <= bin_hi(b, bin_size, H, W)) ? 0 : 1;
    y1 += (bin_lo(b, bin_size, H, W) < ymax) ? 1 : 0;
  }
  out4[0] = x0;
  out4[1] = x1 - 1;
  out4[2] = y0;
  out4[3] = y1 - 1;
}
