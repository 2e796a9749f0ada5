// hostgeom.cpp -- TEST HARNESS (never shipped, never imported by pytorch3d_amd).
//
// Compiles the DEVICE headers pytorch3d_amd/csrc/p3d_geom.h and topk.h for the host with g++
// and wraps them in a brute-force per-pixel loop, so the arithmetic contract and the queue
// logic the gfx950 kernels use can be compared with oracle/ on a machine without a GPU.
// What this cannot cover (binning, LDS staging, culling, stores) is covered by the -m gpu tests.
#include <stdint.h>
#include <string.h>

#include "../../pytorch3d_amd/csrc/atlas_cell.h"
#include "../../pytorch3d_amd/csrc/p3d_geom.h"
#include "../../pytorch3d_amd/csrc/topk.h"
#include "../../pytorch3d_amd/csrc/uvm_sample.h"

using namespace p3d;

// `rect`: when non-null, the fine kernel's fast path is exercised instead of the plain one -- faces that the
// conservative rectangle test (rect_cannot_hit on rect = x0, x1, y0, y1 around the pixel) rejects are skipped, and faces
// with a FaceRec are evaluated by face_hit_rec (shared reciprocals).  Results must not change.
template <typename Queue>
static void raster_pixel(const float* fv, const int64_t* nbr, int64_t f0, int64_t f1, f2 p, float blur, float sqrt_blur,
                         int K, bool persp, bool clip, bool cull, Queue& q, const float* rect = nullptr) {
  q.init();
  for (int64_t f = f0; f < f1; ++f) {
    const float* g = fv + f * 9;
    const f3 v0 = mk3(g[0], g[1], g[2]), v1 = mk3(g[3], g[4], g[5]), v2 = mk3(g[6], g[7], g[8]);
    const FaceSetup fs = face_setup(v0, v1, v2, sqrt_blur, cull);
    if (fs.reject || outside_box(fs, p)) continue;
    FaceHit h;
    if (rect) {
      if (rect_cannot_hit(mk2(v0.x, v0.y), mk2(v1.x, v1.y), mk2(v2.x, v2.y), rect[0], rect[1], rect[2], rect[3], sqrt_blur))
        continue;
      FaceRec fr;
      face_rec_make(v0, v1, v2, &fr);
      if (!face_hit_rec(fr, p, blur, persp, clip, &h)) continue;
    } else if (!face_hit(v0, v1, v2, p, blur, persp, clip, &h)) {
      continue;
    }
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
        // use_mem bit 1: the fast path, with the 8x8 pixel block around the pixel as the culling rectangle
        float rect_v[4];  // x0, x1, y0, y1
        const float* rect = nullptr;
        if (use_mem & 2) {
          const int bx0 = xi & ~7, by0 = yi & ~7;
          const int bx1 = (bx0 + 7 < W ? bx0 + 7 : W - 1), by1 = (by0 + 7 < H ? by0 + 7 : H - 1);
          rect_v[0] = pix_to_ndc(bx0, W, H);
          rect_v[1] = pix_to_ndc(bx1, W, H);
          rect_v[2] = pix_to_ndc(by0, H, W);
          rect_v[3] = pix_to_ndc(by1, H, W);
          rect = rect_v;
        }
        if (!(use_mem & 1) && K <= 8) {
          TopKReg<8, 4> q;
          raster_pixel(fv, nbr, f0, f1, p, blur, sqrt_blur, K, persp, clip, cull, q, rect);
          emit(q, K, o, p2f, zbuf, bary, dists);
        } else {
          TopKMem<150, 4> q;
          raster_pixel(fv, nbr, f0, f1, p, blur, sqrt_blur, K, persp, clip, cull, q, rect);
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
          const f3 w0 = mk3(g[0], g[1], g[2]), w1 = mk3(g[3], g[4], g[5]), w2 = mk3(g[6], g[7], g[8]);
          const f3 gbv = mk3(gb[i * 3], gb[i * 3 + 1], gb[i * 3 + 2]);
          const FaceGrad r = face_sample_bwd(w0, w1, w2, p, gz[i], gbv, gd[i], persp, clip, (clip_on_corrected & 1) != 0);
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


// exact_div property (p3d_geom.h): (float)((double)n * rd) == n / d for every rd within 1 ulp (double) of 1/d -- the
// device's Newton-refined reciprocal is not correctly rounded.  Returns the number of mismatches over `trials` random
// pairs (xorshift, mantissas and exponents drawn independently; every 4th pair has d's mantissa near all-ones / n's near
// a rounding boundary pattern).
extern "C" int64_t hg_exact_div_check(int64_t trials, uint64_t seed) {
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
  auto next = [&]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };
  int64_t bad = 0;
  for (int64_t t = 0; t < trials; ++t) {
    const uint64_t r = next(), r2 = next();
    uint32_t mn = (uint32_t)(r & 0x7fffff), md = (uint32_t)((r >> 23) & 0x7fffff);
    if ((t & 3) == 3) {
      md |= 0x7fff00;                 // long runs of ones
      mn = (mn & 0xff) | ((uint32_t)(r2 >> 40) & 0x7f0000);
    }
    const uint32_t en = 127 - 30 + (uint32_t)(r2 % 60), ed = 127 - 30 + (uint32_t)((r2 >> 8) % 60);
    uint32_t bn = (en << 23) | mn | ((uint32_t)(r2 >> 20) & 1u) << 31, bd = (ed << 23) | md | ((uint32_t)(r2 >> 21) & 1u) << 31;
    float n, d;
    memcpy(&n, &bn, 4);
    memcpy(&d, &bd, 4);
    const float want = n / d;
    const double rd = 1.0 / (double)d;
    uint64_t rb;
    memcpy(&rb, &rd, 8);
    for (int k = -1; k <= 1; ++k) {
      const uint64_t pb = rb + (uint64_t)(int64_t)k;
      double rp;
      memcpy(&rp, &pb, 8);
      if (exact_div(n, rp) != want) ++bad;
    }
  }
  return bad;
}

// Pixel masks of a bounding box (p3d_geom.h: range_mask16, block_mask_8x8 -- what the fine rasterizer's staging and sub-tile
// cull build) against the reference's per-pixel test px > xhi || px < xlo || py > yhi || py < ylo, on random tiles of random
// image sizes (partial tiles at the image edge included) and boxes whose edges sit between, exactly on and far from pixel
// centres, NaN edges included.  Returns the number of (pixel, box) pairs whose mask bit differs from the per-pixel test.
extern "C" int64_t hg_pixel_mask_check(int64_t trials, uint64_t seed) {
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
  auto next = [&]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };
  auto uni = [&]() { return (float)((double)(next() >> 11) * (1.0 / 9007199254740992.0)); };
  int64_t bad = 0;
  for (int64_t t = 0; t < trials; ++t) {
    const int W = 16 + (int)(next() % 1000), H = 16 + (int)(next() % 1000);
    const int ox = (int)(next() % (unsigned)W) & ~15, oy = (int)(next() % (unsigned)H) & ~15;
    const int cols = (W - ox) < 16 ? (W - ox) : 16, rows = (H - oy) < 16 ? (H - oy) : 16;
    const float cx = pix_to_ndc(ox + (int)(next() % 16), W, H), cy = pix_to_ndc(oy + (int)(next() % 16), H, W);
    const float ex = exp2f(-12.0f + 11.0f * uni()), ey = exp2f(-12.0f + 11.0f * uni());
    float xlo = cx - ex * uni(), xhi = cx + ex * uni(), ylo = cy - ey * uni(), yhi = cy + ey * uni();
    if (t % 7 == 0) xlo = pix_to_ndc(ox + (int)(next() % 16), W, H);  // exactly a centre: the comparisons are strict
    if (t % 11 == 0) xhi = pix_to_ndc(ox + (int)(next() % 16), W, H);
    if (t % 13 == 0) ylo = pix_to_ndc(oy + (int)(next() % 16), H, W);
    if (t % 17 == 0) yhi = pix_to_ndc(oy + (int)(next() % 16), H, W);
    if (t % 1009 == 0) xlo = NAN;
    if (t % 1013 == 0) yhi = NAN;
    int xb = 0, xa = 0, yb = 0, ya = 0;
    for (int c = 0; c < 16; ++c) {
      const float xs = pix_to_ndc(ox + c, W, H), ys = pix_to_ndc(oy + c, H, W);
      xb += xs < xlo ? 1 : 0;
      xa += xs > xhi ? 1 : 0;
      yb += ys < ylo ? 1 : 0;
      ya += ys > yhi ? 1 : 0;
    }
    const unsigned cm = range_mask16(xb, xa) & ((1u << cols) - 1u), rm = range_mask16(yb, ya) & ((1u << rows) - 1u);
    for (int sub = 0; sub < 4; ++sub) {
      unsigned lo, hi;
      block_mask_8x8((cm >> ((sub & 1) * 8)) & 0xffu, (rm >> ((sub >> 1) * 8)) & 0xffu, &lo, &hi);
      const uint64_t m = ((uint64_t)hi << 32) | lo;
      for (int lane = 0; lane < 64; ++lane) {
        const int xi = ox + (sub & 1) * 8 + (lane & 7), yi = oy + (sub >> 1) * 8 + (lane >> 3);
        const float px = pix_to_ndc(xi, W, H), py = pix_to_ndc(yi, H, W);
        const bool want = xi < W && yi < H && !(px > xhi || px < xlo || py > yhi || py < ylo);
        if (want != (bool)((m >> lane) & 1ull)) ++bad;
      }
    }
  }
  return bad;
}

// the atlas cell (atlas_cell.h) of P barycentric samples: cells[p] = row * R + col, or -1 where not addressable
extern "C" void hg_atlas_cells(const float* bary, int64_t P, int R, int64_t* cells) {
  for (int64_t p = 0; p < P; ++p) {
    int row, col;
    cells[p] = atlas_cell(bary[p * 3], bary[p * 3 + 1], R, &row, &col) ? (int64_t)row * R + col : -1;
  }
}

// Multi-map UV sampling: the per-sample code of texture_multi.hip (uvm_sample.h) in a plain loop, with the kernels'
// sample set-up (background -> face 0's map at uv (0,0); faces beyond L or F read as zero) and a plain add in place of
// the atomics.  gmaps / gfuv are accumulated in float, in sample order.
struct PlainAdd {
  void operator()(float* dst, float v) const { *dst += v; }
};

extern "C" void hg_uvm(const int64_t* p2f, const float* bary, const float* fuv, const float* maps, const int64_t* ids,
                       int64_t L, int64_t F, int N, int64_t HWK, int M, int Hm, int Wm, int C, int align, int border,
                       int nearest, const float* gtex, float* texels, float* gbary, float* gfuv, float* gmaps) {
  uvm::Volume v;
  v.M = M;
  v.Hm = Hm;
  v.Wm = Wm;
  v.C = C;
  v.align = align != 0;
  v.border = border != 0;
  v.nearest = nearest != 0;
  const int64_t per = (int64_t)M * Hm * Wm * C;
  for (int n = 0; n < N; ++n) {
    const float* vol = maps + n * per;
    float* gvol = gmaps ? gmaps + n * per : nullptr;
    for (int64_t i = 0; i < HWK; ++i) {
      const int64_t p = (int64_t)n * HWK + i;
      const int64_t f = p2f[p];
      float u = 0.0f, vv = 0.0f;
      const bool has_uv = f >= 0 && f < F;
      if (has_uv) {
        const float* b = bary + p * 3;
        const float* r = fuv + f * 6;
        u = (b[0] * r[0] + b[1] * r[2]) + b[2] * r[4];
        vv = (b[0] * r[1] + b[1] * r[3]) + b[2] * r[5];
      }
      const int64_t fi = f < 0 ? 0 : f;
      const bool valid = fi < L && (f < 0 || f < F);
      if (texels)
        for (int ch = 0; ch < C; ++ch) texels[p * C + ch] = 0.0f;
      if (gbary)
        for (int j = 0; j < 3; ++j) gbary[p * 3 + j] = 0.0f;
      if (!valid) continue;
      const uvm::Coords c = uvm::coords_of(v, u, vv, ids[fi]);
      const uvm::Footprint fp = uvm::footprint_of(v, c);
      if (texels) uvm::sample_forward(v, vol, fp, texels + p * C);
      if (gtex) {
        float du, dv;
        uvm::sample_backward(v, vol, gvol, fp, c, gtex + p * C, PlainAdd(), &du, &dv);
        if (f >= 0 && !v.nearest) {
          const float* b = bary + p * 3;
          const float* r = fuv + f * 6;
          for (int j = 0; j < 3; ++j) {
            gfuv[f * 6 + 2 * j] += b[j] * du;
            gfuv[f * 6 + 2 * j + 1] += b[j] * dv;
            gbary[p * 3 + j] = r[2 * j] * du + r[2 * j + 1] * dv;
          }
        }
      }
    }
  }
}

// TopKPairs<8> (the pair-register queue, masked moves as plain ifs on the host) against TopKReg<8, 4> on random
// operation sequences: inserts gated by admits() as the kernels do, depth ties, repeated indices, find + erase.
// Returns the number of states in which the two queues differ.
extern "C" int64_t hg_queue_pairs_check(int64_t sequences, uint64_t seed) {
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 7;
  auto next = [&]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };
  int64_t bad = 0;
  for (int64_t t = 0; t < sequences; ++t) {
    TopKReg<8, 4> a;
    TopKPairs<8> b;
    TopKPairs<8, true> c;  // one 64-bit key compare per entry (depths here are >= +0, as in the kernels that use it)
    a.init();
    b.init();
    c.init();
    const int ops = 4 + (int)(next() % 40);
    const int zlevels = 1 + (int)(next() % 12);  // few distinct depths: many exact ties
    for (int o = 0; o < ops; ++o) {
      const uint64_t r = next();
      const float z = (float)(r % zlevels) * 0.25f + ((r >> 8) % 5 == 0 ? 0.0f : 1e-3f * (float)((r >> 12) % 3));
      const int idx = (int)((r >> 20) % 64);
      const float pl[4] = {(float)(r >> 30 & 1023) * 0.5f - 100.0f, (float)(o), (float)(t & 255), -(float)idx};
      if ((r >> 40) % 7 == 0) {
        const int want = (int)((r >> 44) % 64);
        const int fa = a.find(want), fb = b.find(want), fc = c.find(want);
        if (fa != fb || fa != fc) ++bad;
        if (fa >= 0) {
          if (a.payload_at(0, fa) != b.payload_at(0, fb) || a.payload_at(0, fa) != c.payload_at(0, fc)) ++bad;
          a.erase(fa);
          b.erase(fb);
          c.erase(fc);
        }
      } else {
        const bool ad_a = a.admits(8, z, idx), ad_b = b.admits(8, z, idx), ad_c = c.admits(8, z, idx);
        if (ad_a != ad_b || ad_a != ad_c) ++bad;
        if (ad_a) a.insert(8, z, idx, pl);
        if (ad_b) b.insert(8, z, idx, pl);
        if (ad_c) c.insert(8, z, idx, pl);
      }
      if ((a.kth_z(8) != b.kth_z(8) || a.kth_z(8) != c.kth_z(8)) && !(a.kth_z(8) != a.kth_z(8))) ++bad;
      for (int k = 0; k < 8; ++k) {
        if (a.valid(k) != b.valid(k) || a.ix(k) != b.ix(k) || a.ix(k) != c.ix(k)) ++bad;
        if (!a.valid(k)) continue;
        if (a.zf(k) != b.zf(k) || a.zf(k) != c.zf(k)) ++bad;
        for (int p = 0; p < 4; ++p)
          if (a.pay(p, k) != b.pay(p, k) || a.pay(p, k) != c.pay(p, k)) ++bad;
      }
    }
  }
  return bad;
}

// The payload-free form (long point queues; mesh queues for 16 < K <= 64): TopKPairs<12, *, 0> against TopKReg<12, 0>, inserts
// gated by admits(), with a live capacity K <= 12 drawn per sequence (entries K.. must stay empty, the K-th entry is the
// admission threshold) and occasional erasures (the clipped-neighbour rule: the queue has room again afterwards).
extern "C" int64_t hg_queue_pairs0_check(int64_t sequences, uint64_t seed) {
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 11;
  auto next = [&]() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  };
  int64_t bad = 0;
  for (int64_t t = 0; t < sequences; ++t) {
    TopKReg<12, 0> a;
    TopKPairs<12, false, 0> b;
    TopKPairs<12, true, 0> c;
    a.init();
    b.init();
    c.init();
    const int ops = 4 + (int)(next() % 60);
    const int zlevels = 1 + (int)(next() % 9);
    const int K = (next() % 3 == 0) ? 12 : 1 + (int)(next() % 12);
    for (int o = 0; o < ops; ++o) {
      const uint64_t r = next();
      const float z = (float)(r % zlevels) * 0.5f + ((r >> 8) % 4 == 0 ? 0.0f : 1e-4f * (float)((r >> 12) % 3));
      const int idx = (int)((r >> 20) % 97);
      const float pl[1] = {0.0f};
      if ((r >> 40) % 11 == 0) {  // erase a queued primitive, if this one is queued
        const int at = a.find(idx);
        if (at != b.find(idx) || at != c.find(idx)) ++bad;
        if (at >= 0) {
          a.erase(at);
          b.erase(at);
          c.erase(at);
        }
        continue;
      }
      const bool ad = a.admits(K, z, idx);
      if (ad != b.admits(K, z, idx) || ad != c.admits(K, z, idx)) ++bad;
      if (ad) {
        a.insert(K, z, idx, pl);
        b.insert(K, z, idx, pl);
        c.insert(K, z, idx, pl);
      }
      if ((a.kth_z(K) != b.kth_z(K) || a.kth_z(K) != c.kth_z(K))) ++bad;
      for (int k = 0; k < 12; ++k) {
        if (a.valid(k) != b.valid(k) || a.valid(k) != c.valid(k) || a.ix(k) != b.ix(k) || a.ix(k) != c.ix(k)) ++bad;
        if (a.valid(k) && (a.zf(k) != b.zf(k) || a.zf(k) != c.zf(k))) ++bad;
      }
    }
  }
  return bad;
}
