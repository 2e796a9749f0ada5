// raster_points.hip -- point-cloud rasterization for gfx950: fine / naive forward, backward.
//
// Replaces RasterizePointsNaiveCudaKernel, RasterizePointsFineCudaKernel and
// RasterizePointsBackwardCudaKernel (pytorch3d/csrc/rasterize_points/rasterize_points.cu:87-149,
// 223-298, 366-411).  Same tile / stage / sub-tile-cull structure as raster_mesh.hip: a workgroup
// owns a 16x16 tile of one bin, streams the bin's points through LDS 256 at a time (dropping
// z < 0 and points whose x+-r, y+-r box misses the tile), each wave culls against its 8x8
// sub-tile with one lane per point, and the survivors are tested per pixel:
// accept iff dist2 = dx*dx + dy*dy < r*r (rasterize_points.cu:55-60).  Queue order is
// (z, point index) -- the CPU variant's tuple order (rasterize_points_cpu.cpp:55-75); the CUDA
// variant compares z only and leaves ties to the (unspecified) bin order.
#include "binning.h"
#include "tile_map.h"
#include "chunk_order.h"
#include "p3d_geom.h"
#include "topk.h"
#include "wave_table.h"

namespace p3d {

namespace {

constexpr int kTile = 16;
constexpr int kStage = 256;

struct PointArgs {
  const float* points;
  const float* radius;
  const int64_t* first;
  const int64_t* count;
  BinCSR csr;
  int N, H, W, K;
  TileMap tm;
  int32_t* idxs;
  float* zbuf;
  float* dists;
  // short workspaces (binning.h): the device flag "the lists did not fit", or null.  A binned launch returns at once when it
  // is up, the naive launch behind it returns at once when it is not (as in raster_mesh.hip: MeshArgs::overflow).
  const int* overflow;
  // The compositor behind the fine stage (p3d_rasterize_points_composite; null features: plain rasterization).  PointsRenderer
  // (renderer/points/renderer.py:56-76) turns the fragments into weights = 1 - dists / r^2 and alpha-composites the points' features
  // front to back (alpha_composite.cu:24-68): when a pixel's K entries are final they are in the workgroup's LDS, so the pixel of the
  // image is formed there instead of by three more launches that read idx and dists back.
  const float* features;  // (P, C) rows
  float* images;          // (N, H, W, C), every element written
  int C;                  // 1..4
  float inv_r2;           // float(1) / float(r * r): torch evaluates `dists / (r * r)` as dists * that
  int comp_mode;          // P3D_COMPOSITE_ALPHA (alpha_composite.cu:24-68) or P3D_COMPOSITE_NORM_SUM (norm_weighted_sum.cu:24-78)
};

// 12 / 16 adjacent bytes at 4-byte alignment: one global_load_dwordx3 / x4
struct __attribute__((packed, aligned(4))) PFeat3 {
  float x, y, z;
};
struct __attribute__((packed, aligned(4))) PFeat4 {
  float x, y, z, w;
};

// One entry of a pixel, front to back (alpha_composite.cu:47-62 with alpha = 1 - dist2 * inv_r2; every product and difference a
// separate float32 operation, in composite.hip's order: the same bits as the operators run one after the other).
constexpr float kSplatEpsNorm = 1e-4f;  // norm_weighted_sum.cu:20
struct SplatPixel {
  float acc[4];
  float cum;
  float norm;  // NORM_SUM: the clamped sum of the pixel's weights (set by the caller before the first add), else unused
  bool norm_mode;
  __device__ __forceinline__ void init(int mode = P3D_COMPOSITE_ALPHA) {
    acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
    cum = 1.0f;
    norm = 0.0f;
    norm_mode = mode == P3D_COMPOSITE_NORM_SUM;
  }
  // NORM_SUM's first walk over the pixel's entries: norm += alpha of the valid ones, in k order (norm_weighted_sum.cu:47-55)
  __device__ __forceinline__ void weigh(int id, float d2, float inv_r2) {
    if (id >= 0) norm += 1.0f - d2 * inv_r2;
  }
  __device__ __forceinline__ void clamp_norm() {
    if (norm < kSplatEpsNorm) norm = kSplatEpsNorm;
  }
  __device__ __forceinline__ void add(const float* __restrict__ features, int C, int id, float d2, float inv_r2) {
    if (id < 0) return;
    const float al = 1.0f - d2 * inv_r2;
    float fv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* fp = features + (int64_t)id * C;
    if (C == 3) {  // uniform
      const PFeat3 v = *reinterpret_cast<const PFeat3*>(fp);
      fv[0] = v.x, fv[1] = v.y, fv[2] = v.z;
    } else if (C == 4) {
      const PFeat4 v = *reinterpret_cast<const PFeat4*>(fp);
      fv[0] = v.x, fv[1] = v.y, fv[2] = v.z, fv[3] = v.w;
    } else {
      fv[0] = fp[0];
      if (C > 1) fv[1] = fp[1];
    }
    if (norm_mode) {  // uniform
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] += fv[c] * al / norm;
      return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] += fv[c] * cum * al;
    cum = cum * (1 - al);
  }
  __device__ __forceinline__ void store(float* __restrict__ px, int C) const {
    if (C == 3) {  // uniform
      *reinterpret_cast<PFeat3*>(px) = PFeat3{acc[0], acc[1], acc[2]};
    } else if (C == 4) {
      *reinterpret_cast<PFeat4*>(px) = PFeat4{acc[0], acc[1], acc[2], acc[3]};
    } else {
      px[0] = acc[0];
      if (C > 1) px[1] = acc[1];
    }
  }
};

// PAYLOAD: the queue carries dist2 next to (z, idx).  Without it (long queues: 2 registers per entry instead of 3) the
// distance is recomputed from the point's coordinates when the pixel is written -- the same two subtractions, two
// products and one sum as in the test (rasterize_points.cu:55-60), so the same bits.
// WAVES: minimum waves per SIMD the register allocation has to leave room for (512 / WAVES VGPRs per lane).
// EXACTK: K == KT is known at compile time (the pair queues' insertion then has no test on K left).
template <typename Queue, int KT, bool IN_REGS, bool BINNED, bool PAYLOAD = true, int WAVES = 2, bool EXACTK = false>
__global__ __launch_bounds__(kStage, WAVES) void point_raster_kernel(PointArgs a) {
  __shared__ float4 s_box[kStage];  // x-r, x+r, y-r, y+r
  __shared__ float4 s_pt[kStage];   // x, y, z, r*r
  __shared__ int s_idx[kStage];
  __shared__ float s_key[kStage];   // depth key of the staged point (its z)
  __shared__ float s_qlow[kStage];  // lower bound of the keys at sorted positions >= i
  __shared__ int s_order[kStage];
  __shared__ ChunkOrderScratch s_ord;
  __shared__ int s_wcnt[kStage / kWave];

  if (a.overflow != nullptr && (*a.overflow != 0) == BINNED) return;  // uniform (scalar load)
  TileCoord tc;
  if (!tile_of_block(a.tm, blockIdx.x, &tc)) return;
  const int n = tc.n, by = tc.by, bx = tc.bx, ty = tc.ty, tx = tc.tx;

  const int H = a.H, W = a.W;
  const int y_end = min(H, (by + 1) * a.tm.bin_size);
  const int x_end = min(W, (bx + 1) * a.tm.bin_size);
  const int ty0 = by * a.tm.bin_size + ty * kTile;
  const int tx0 = bx * a.tm.bin_size + tx * kTile;
  if (ty0 >= y_end || tx0 >= x_end) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int sy0 = ty0 + (w >> 1) * 8;
  const int sx0 = tx0 + (w & 1) * 8;
  const int yi = sy0 + (lane >> 3);
  const int xi = sx0 + (lane & 7);
  const bool pix_ok = yi < y_end && xi < x_end;
  const bool wave_ok = sy0 < y_end && sx0 < x_end;
  // one pix_to_ndc per lane: lane l < 16 evaluates the tile's pixel column tx0 + l, lane 16 + l its row ty0 + l; the lane's own
  // centre and the eight extents are fetched from those lanes -- same function, same argument, same bits (raster_mesh.hip, round 4)
  const float pxy = (lane & 16) ? pix_to_ndc(ty0 + (lane & 15), H, W) : pix_to_ndc(tx0 + (lane & 15), W, H);
  auto col_centre = [&](int x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pxy), x - tx0)); };
  auto row_centre = [&](int y) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pxy), 16 + y - ty0)); };
  const float xf = __int_as_float(__builtin_amdgcn_ds_bpermute((xi - tx0) << 2, __float_as_int(pxy)));
  const float yf = __int_as_float(__builtin_amdgcn_ds_bpermute((16 + yi - ty0) << 2, __float_as_int(pxy)));

  const float tile_x0 = col_centre(tx0), tile_x1 = col_centre(min(tx0 + kTile, x_end) - 1);
  const float tile_y0 = row_centre(ty0), tile_y1 = row_centre(min(ty0 + kTile, y_end) - 1);
  const float sub_x0 = col_centre(min(sx0, x_end - 1)), sub_x1 = col_centre(max(min(sx0 + 8, x_end) - 1, tx0));
  const float sub_y0 = row_centre(min(sy0, y_end - 1)), sub_y1 = row_centre(max(min(sy0 + 8, y_end) - 1, ty0));

  int64_t src_base;
  int count;
  if (BINNED) {
    const int64_t row = ((int64_t)n * a.tm.BH + by) * a.tm.BW + bx;
    src_base = a.csr.offset[row];
    count = a.csr.total[row];
  } else {
    src_base = a.first[n];
    count = (int)a.count[n];
  }

  Queue q;
  q.init();
  const int K = EXACTK ? KT : a.K;

  for (int base = 0; base < count; base += kStage) {
    const int i = base + tid;
    bool keep = false;
    float px = 0.f, py = 0.f, pz = 0.f, r = 0.f;
    int pid = -1;
    if (i < count) {
      pid = BINNED ? a.csr.list[(src_base + i) * a.csr.stride] : (int)(src_base + i);
      const float* g = a.points + (int64_t)pid * 3;
      px = g[0];
      py = g[1];
      pz = g[2];
      r = a.radius[pid];
      // a pixel with dist2 < r*r lies inside [x-r, x+r] x [y-r, y+r], so this cull is exact
      const bool off_tile = tile_x0 > px + r || tile_x1 < px - r || tile_y0 > py + r || tile_y1 < py - r;
      keep = !(pz < 0.0f) && !off_tile;
    }
    const unsigned long long km = __ballot(keep);
    if (lane == 0) s_wcnt[w] = __popcll(km);
    __syncthreads();
    int pos = mask_rank(km);
    int staged = 0;
#pragma unroll
    for (int j = 0; j < kStage / kWave; ++j) {
      const int c = s_wcnt[j];
      if (j < w) pos += c;
      staged += c;
    }
    if (keep) {
      s_box[pos] = make_float4(px - r, px + r, py - r, py + r);
      // a queue ordered by the 64-bit key (z bits, idx) needs +0.0 for a zero depth: -0.0 would sort last.  The sign of an
      // exactly zero depth is then the one bit that differs from the reference (K in {8, 10, 16, 32, 40, 50, 64, 100}).
      s_pt[pos] = make_float4(px, py, Queue::kKeyOrder ? pz + 0.0f : pz, r * r);
      s_idx[pos] = pid;
      s_key[pos] = pz;
    }
    __syncthreads();
    // front-to-back visiting order (chunk_order.h); a point's depth is its sample depth, so the cull is exact
    chunk_bucket_order(s_key, staged, s_order, s_qlow, s_ord, tid);

    if (wave_ok) {
      for (int jb = 0; jb < staged; jb += kWave) {
        // nothing at or behind sorted position jb can enter any queue of this wave any more
        if (__ballot(pix_ok && !(s_qlow[jb] > q.kth_z(K))) == 0) break;
        const int j = jb + lane;
        bool touch = false;
        int oj = 0;
        if (j < staged) {
          oj = s_order[j];
          const float4 b = s_box[oj];
          touch = !(sub_x0 > b.y || sub_x1 < b.x || sub_y0 > b.w || sub_y1 < b.z);
        }
        unsigned long long cand = __ballot(touch);
        while (cand) {
          const int jj = __builtin_amdgcn_readlane(oj, __builtin_ctzll(cand));
          cand &= cand - 1;
          const float4 pt = s_pt[jj];
          const float dx = xf - pt.x;
          const float dy = yf - pt.y;
          const float dist2 = dx * dx + dy * dy;
          if (pix_ok && dist2 < pt.w) {
            const int id = s_idx[jj];
            if (q.admits(K, pt.z, id)) {
              const float pl[1] = {dist2};  // ignored by a queue without payload
              q.insert(K, pt.z, id, pl);
            }
          }
        }
      }
    }
    __syncthreads();
  }

  if (pix_ok) {
    const int64_t base = (((int64_t)n * H + (H - 1 - yi)) * W + (W - 1 - xi)) * K;
    if constexpr (IN_REGS) {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k < K) {
          const bool ok = q.valid(k);
          a.idxs[base + k] = ok ? q.ix(k) : -1;
          a.zbuf[base + k] = ok ? q.zf(k) : -1.0f;
          if constexpr (PAYLOAD) {
            a.dists[base + k] = ok ? q.pay(0, k) : -1.0f;
          } else {
            float d2 = -1.0f;
            if (ok) {
              const float* g = a.points + (int64_t)q.ix(k) * 3;
              const float dx = xf - g[0];
              const float dy = yf - g[1];
              d2 = dx * dx + dy * dy;
            }
            a.dists[base + k] = d2;
          }
        }
      }
    } else {
      for (int k = 0; k < K; ++k) {
        const bool ok = q.valid(k);
        a.idxs[base + k] = ok ? q.ix(k) : -1;
        a.zbuf[base + k] = ok ? q.zf(k) : -1.0f;
        a.dists[base + k] = ok ? q.pay(0, k) : -1.0f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Sorted fine stage (round 5): ONE WAVE = ONE 8x8 SUB-TILE, no workgroup barrier, no sorted-insertion network.
//
// The kernel above walks a tile's list in arrival order, 256 points at a time behind barriers, and keeps every pixel's K nearest
// in a sorted register queue: an insertion costs ~4 VALU per entry and runs for the whole wave whenever ONE of its 64 pixels
// admits the point -- on BASELINE configs[3] (1M points, ~80 splats over every pixel, K = 10) that was ~26 000 VALU
// instructions per wave for ~650 candidate points, with one wave per SIMD slot and no second round to balance (0.175 ms:
// 0.034 of the HBM roofline).  Here the wave first puts ITS candidates -- the points of the bin whose
// box touches its sub-tile -- in exact ascending (depth, index) order, then visits them front to back:
//   pass A  64 list entries at a time: load, sub-tile box cull, ordered ballot compaction of the 64-bit keys (z bits | index)
//           into LDS (at most kSortCap per round; a longer list takes several rounds, see below);
//   pass B  exact sort: 64 depth buckets between the round's min and max depth (one LDS integer atomic per key gives its place
//           in the bucket, a wave scan the bucket starts), then every key's rank INSIDE its bucket by counting the bucket's
//           smaller keys (buckets hold ~10 keys) -- ~50 instructions per 64 keys for the buckets, ~6 per bucket mate for the rank;
//   pass C  front to back, 64 candidates per block (their x, y, r gathered by index: only visited blocks pay for it): a pixel
//           inside the splat APPENDS the key to its queue -- LDS, [k][lane] layout: one ds_write_b64, conflict-free -- because
//           a sorted stream arrives in queue order; the pixel is done when it holds K entries, the wave when all its pixels are
//           (one ballot per block and per candidate): ~170 of the ~650 candidates are ever visited at K = 10.
// A queue in LDS has no capacity classes: the same kernel serves K = 1 .. 150 (dynamic LDS K x 512 B) -- the K = 100 register
// queue took 3.0 ms on this cloud, the private-memory queue above it more.  Lists longer than kSortCap candidates: later rounds
// are sorted the same way but no longer arrive in QUEUE order, so their points are inserted (per lane: shift the larger entries,
// in LDS), and pass A drops every key that is not below the largest K-th key of the wave.  Results are those of the kernel above
// bit for bit: the K smallest keys under the total order (z, index), dist2 recomputed by the same expression at the store.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSortCap = 768;   // keys sorted per round and wave: 2 x 6 KB of LDS
constexpr int kSortSlots = kSortCap / kWave;
constexpr int kBatch = 8;           // sub-batches of 64 list entries loaded at once in pass A

__device__ __forceinline__ int depth_bucket(unsigned zbits, float zlo, float scale) {
  const int b = (int)((__uint_as_float(zbits) - zlo) * scale);  // monotone in z: subtraction, product with scale >= 0, conversion
  return b < 0 ? 0 : (b > kWave - 1 ? kWave - 1 : b);
}

template <bool BINNED>
__global__ __launch_bounds__(kWave, 2) void point_sorted_kernel(PointArgs a) {
  extern __shared__ __align__(16) unsigned long long s_queue[];  // [K][64]: entry k of lane l at k * 64 + l
  __shared__ unsigned long long s_a[kSortCap];
  __shared__ unsigned long long s_b[kSortCap];
  __shared__ int s_hist[kWave];
  __shared__ int s_start[kWave];
  __shared__ unsigned s_zrange[2];

  if (a.overflow != nullptr && (*a.overflow != 0) == BINNED) return;  // uniform (scalar load)
  TileCoord tc;
  if (!tile_of_block(a.tm, blockIdx.x >> 2, &tc)) return;
  const int n = tc.n, H = a.H, W = a.W, K = a.K;
  const int y_end = min(H, (tc.by + 1) * a.tm.bin_size);
  const int x_end = min(W, (tc.bx + 1) * a.tm.bin_size);
  const int sub = (int)(blockIdx.x & 3u);
  const int sy0 = tc.by * a.tm.bin_size + tc.ty * kTile + (sub >> 1) * 8;
  const int sx0 = tc.bx * a.tm.bin_size + tc.tx * kTile + (sub & 1) * 8;
  if (sy0 >= y_end || sx0 >= x_end) return;  // uniform
  const int lane = threadIdx.x;
  const int yi = sy0 + (lane >> 3), xi = sx0 + (lane & 7);
  const bool pix_ok = yi < y_end && xi < x_end;
  // one pix_to_ndc per lane: lane l < 8 the sub-tile's pixel column sx0 + l, lane 8 + l its row sy0 + l (lanes 16.. repeat)
  const float pxy = (lane & 8) ? pix_to_ndc(sy0 + (lane & 7), H, W) : pix_to_ndc(sx0 + (lane & 7), W, H);
  auto centre = [&](int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pxy), l)); };
  const float xf = __int_as_float(__builtin_amdgcn_ds_bpermute((lane & 7) << 2, __float_as_int(pxy)));
  const float yf = __int_as_float(__builtin_amdgcn_ds_bpermute((8 + (lane >> 3)) << 2, __float_as_int(pxy)));
  const float sub_x0 = centre(0), sub_x1 = centre(min(sx0 + 8, x_end) - 1 - sx0);
  const float sub_y0 = centre(8), sub_y1 = centre(8 + min(sy0 + 8, y_end) - 1 - sy0);

  int64_t src_base;
  int count;
  if (BINNED) {
    const int64_t row = ((int64_t)n * a.tm.BH + tc.by) * a.tm.BW + tc.bx;
    src_base = a.csr.offset[row];
    count = a.csr.total[row];
  } else {
    src_base = a.first[n];
    count = (int)a.count[n];
  }

  int cnt = 0;                     // entries of this lane's queue (ascending keys)
  unsigned long long kth = ~0ull;  // the K-th key once the queue is full: nothing at or above it can enter
  bool first = true;               // this round's stream arrives in queue order (the queues were empty when it was sorted)
  int pos = 0;
  while (pos < count) {  // uniform: one round = up to kSortCap candidates
    // a later round: no lane can use a key at or above the largest K-th key of the wave (~0 while some queue has room)
    unsigned long long kmax = 0;
    if (first) {
      kmax = ~0ull;
    } else {
      unsigned long long rem = __ballot(pix_ok);
      while (rem) {  // uniform
        const int l = __builtin_ctzll(rem);
        rem &= rem - 1;
        const unsigned long long kl = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(kth >> 32), l) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)(unsigned)kth, l);
        kmax = kl > kmax ? kl : kmax;
      }
    }
    // ---- pass A ----  kBatch x 64 list entries in flight at once: a lane's loads (list entry -> point, radius) are a chain of two
    // memory round trips, and one entry per lane and trip made this pass the kernel (first build: 27 chains per wave on
    // BASELINE configs[3], ~0.1 ms of 0.13 at K = 1).  The sub-batches are committed in list order; the one that would not fit the
    // round's buffer ends the round and is read again by the next (only the first round is order-free anyway).
    int nc = 0;
    unsigned zlo_l = 0xffffffffu, zhi_l = 0u;
    bool full = false;
    while (pos < count && !full) {  // uniform
      int pid[kBatch];
      float px[kBatch], py[kBatch], pz[kBatch], pr[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int i = pos + u * kWave + lane;
        pid[u] = -1;
        if (i < count) pid[u] = BINNED ? a.csr.list[(src_base + i) * a.csr.stride] : (int)(src_base + i);
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        px[u] = py[u] = pz[u] = pr[u] = 0.0f;
        if (pid[u] >= 0) {
          const float* g = a.points + (int64_t)pid[u] * 3;
          px[u] = g[0];
          py[u] = g[1];
          pz[u] = g[2];
          pr[u] = a.radius[pid[u]];
        }
      }
      int done = 0;
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        if (!full && pos + u * kWave < count) {  // uniform
          // a pixel with dist2 < r*r lies inside [x-r, x+r] x [y-r, y+r]: the cull of point_raster_kernel, on the sub-tile
          const bool off = sub_x0 > px[u] + pr[u] || sub_x1 < px[u] - pr[u] || sub_y0 > py[u] + pr[u] || sub_y1 < py[u] - pr[u];
          // +0.0 canonicalises a zero depth: the keys order by their bits (as the pair queues of the kernel above)
          const unsigned long long key = ((unsigned long long)__float_as_uint(pz[u] + 0.0f) << 32) | (unsigned)pid[u];
          const bool keep = pid[u] >= 0 && !(pz[u] < 0.0f) && !off && key < kmax;
          const unsigned long long km = __ballot(keep);
          const int c = __popcll(km);
          if (nc + c > kSortCap) {
            full = true;
          } else {
            if (keep) {
              s_a[nc + mask_rank(km)] = key;
              const unsigned zb = (unsigned)(key >> 32);
              zlo_l = zb < zlo_l ? zb : zlo_l;
              zhi_l = zb > zhi_l ? zb : zhi_l;
            }
            nc += c;
            done = u + 1;
          }
        }
      }
      pos += done * kWave;
    }
    if (nc == 0) continue;  // uniform
    // ---- pass B: exact ascending order of s_a[0 .. nc) ----
    if (lane == 0) {
      s_zrange[0] = 0xffffffffu;
      s_zrange[1] = 0u;
    }
    s_hist[lane] = 0;
    __syncthreads();
    atomicMin(&s_zrange[0], zlo_l);
    atomicMax(&s_zrange[1], zhi_l);
    __syncthreads();
    const float zlo = __uint_as_float(s_zrange[0]);
    const float zspan = __uint_as_float(s_zrange[1]) - zlo;
    const float scale = zspan > 0.0f && zspan < INFINITY ? (float)kWave / zspan : 0.0f;
    unsigned long long mine[kSortSlots];
    int tag[kSortSlots];  // bucket << 16 | place in the bucket (arrival order)
#pragma unroll
    for (int t = 0; t < kSortSlots; ++t) {
      const int i = t * kWave + lane;
      mine[t] = 0;
      tag[t] = 0;
      if (i < nc) {
        mine[t] = s_a[i];
        const int b = depth_bucket((unsigned)(mine[t] >> 32), zlo, scale);
        tag[t] = (b << 16) | atomicAdd(&s_hist[b], 1);
      }
    }
    __syncthreads();
    const int hcount = s_hist[lane];
    int incl = hcount;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int up = __shfl_up(incl, d, kWave);
      if (lane >= d) incl += up;
    }
    s_start[lane] = incl - hcount;
    int hmax = hcount;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) hmax = max(hmax, __shfl_xor(hmax, d, kWave));
    hmax = __builtin_amdgcn_readfirstlane(hmax);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kSortSlots; ++t) {
      const int i = t * kWave + lane;
      if (i < nc) s_b[s_start[tag[t] >> 16] + (tag[t] & 0xffff)] = mine[t];
    }
    __syncthreads();
#pragma unroll 1
    for (int jb = 0; jb < nc; jb += kWave) {
      const int j = jb + lane;
      if (j < nc) {
        const unsigned long long key = s_b[j];
        const int b = depth_bucket((unsigned)(key >> 32), zlo, scale);
        const int s0 = s_start[b], e0 = s0 + s_hist[b];
        int less = 0;
        for (int t = 0; t < hmax; ++t) {  // uniform bound; lanes of short buckets idle
          const int at = s0 + t;
          if (at < e0) less += s_b[at] < key ? 1 : 0;
        }
        s_a[s0 + less] = key;
      }
    }
    __syncthreads();
    // ---- pass C: front to back ----
    // (the x, y, r of block jb + 64 are requested before block jb is walked: the walk has no memory access of its own)
    unsigned long long ck = lane < nc ? s_a[lane] : ~0ull;
    float cx = 0.0f, cy = 0.0f, cr = 0.0f;
    if (lane < nc) {
      const float* g = a.points + (int64_t)(unsigned)ck * 3;
      cx = g[0];
      cy = g[1];
      cr = a.radius[(unsigned)ck];
    }
#pragma unroll 1
    for (int jb = 0; jb < nc; jb += kWave) {
      const unsigned clo = (unsigned)ck, chi = (unsigned)(ck >> 32);
      {
        // the block's first key is its smallest: is it still below some pixel's K-th key?
        const unsigned long long k0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)chi) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)clo);
        if (__ballot(pix_ok && k0 < kth) == 0) break;  // uniform
      }
      const float bx = cx, by = cy, br2 = cr * cr;
      const int jn = jb + kWave + lane;
      ck = jn < nc ? s_a[jn] : ~0ull;
      if (jn < nc) {
        const float* g = a.points + (int64_t)(unsigned)ck * 3;
        cx = g[0];
        cy = g[1];
        cr = a.radius[(unsigned)ck];
      }
      const int m = min(kWave, nc - jb);
      for (int t = 0; t < m; ++t) {  // uniform
        const unsigned long long key = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)chi, t) << 32) |
                                       (unsigned)__builtin_amdgcn_readlane((int)clo, t);
        const bool live = pix_ok && key < kth;
        if (__ballot(live) == 0) break;  // uniform: keys ascend, so do the misses
        const float dx = xf - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx), t));
        const float dy = yf - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(by), t));
        const float dist2 = dx * dx + dy * dy;
        if (live && dist2 < __int_as_float(__builtin_amdgcn_readlane(__float_as_int(br2), t))) {
          if (first) {  // uniform
            s_queue[cnt * kWave + lane] = key;
            ++cnt;
            if (cnt == K) kth = key;
          } else {
            int at = cnt < K ? cnt : K - 1;  // the hole: a new last entry, or the K-th falls off
            while (at > 0) {
              const unsigned long long prev = s_queue[(at - 1) * kWave + lane];
              if (!(prev > key)) break;
              s_queue[at * kWave + lane] = prev;
              --at;
            }
            s_queue[at * kWave + lane] = key;
            if (cnt < K) ++cnt;
            if (cnt == K) kth = s_queue[(K - 1) * kWave + lane];
          }
        }
      }
    }
    first = false;
    __syncthreads();
  }

  if (pix_ok) {
    // a pixel's K entries are contiguous in each output: 16-byte stores when K % 4 == 0, 8-byte ones when K is even
    const int64_t base = (((int64_t)n * H + (H - 1 - yi)) * W + (W - 1 - xi)) * K;
    auto entry = [&](int k, int* id, float* z, float* d2) {
      *id = -1;
      *z = -1.0f;
      *d2 = -1.0f;
      if (k < cnt) {
        const unsigned long long key = s_queue[k * kWave + lane];
        *id = (int)(unsigned)key;
        *z = __uint_as_float((unsigned)(key >> 32));
        const float* g = a.points + (int64_t)*id * 3;
        const float dx = xf - g[0];
        const float dy = yf - g[1];
        *d2 = dx * dx + dy * dy;
      }
    };
    if ((K & 3) == 0) {
      for (int k = 0; k < K; k += 4) {
        int id[4];
        float z[4], d2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) entry(k + j, &id[j], &z[j], &d2[j]);
        *reinterpret_cast<int4*>(a.idxs + base + k) = make_int4(id[0], id[1], id[2], id[3]);
        *reinterpret_cast<float4*>(a.zbuf + base + k) = make_float4(z[0], z[1], z[2], z[3]);
        *reinterpret_cast<float4*>(a.dists + base + k) = make_float4(d2[0], d2[1], d2[2], d2[3]);
      }
    } else if ((K & 1) == 0) {
      for (int k = 0; k < K; k += 2) {
        int id[2];
        float z[2], d2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) entry(k + j, &id[j], &z[j], &d2[j]);
        *reinterpret_cast<int2*>(a.idxs + base + k) = make_int2(id[0], id[1]);
        *reinterpret_cast<float2*>(a.zbuf + base + k) = make_float2(z[0], z[1]);
        *reinterpret_cast<float2*>(a.dists + base + k) = make_float2(d2[0], d2[1]);
      }
    } else {
      for (int k = 0; k < K; ++k) {
        int id;
        float z, d2;
        entry(k, &id, &z, &d2);
        a.idxs[base + k] = id;
        a.zbuf[base + k] = z;
        a.dists[base + k] = d2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile-sorted kernel (round 5, binned launches with K <= kTileSortedMaxK): the cooperative four-wave workgroup of
// point_raster_kernel (one staging per 16x16 tile, shared by its four 8x8 waves) with the EXACT front-to-back order and the
// append-only LDS queues of point_sorted_kernel.  Why: with the tile pre-sort point_raster_kernel visits ~250 candidates per wave on
// BASELINE configs[3], and for every one that ANY of the wave's 64 pixels admits the whole wave runs the sorted-insertion network of
// its register queue (~5 VALU per entry).  A stream in exact (depth, index) order needs no network -- a hit is appended -- but the
// pre-sort orders by depth BUCKET only.  So:
//   * the tile's list is pre-sorted into 256 depth buckets (bucket_sort_tile) and cut into chunks at bucket boundaries (the
//     buckets that fit 256 entries), so that every key of chunk c + 1 sorts after every key of chunk c;
//   * a staged chunk is put in exact order by rank: same-bucket entries are adjacent after the ordered compaction, and an
//     entry moves by (same-bucket entries behind it with a smaller key) - (those ahead of it with a larger key), a window scan of
//     the bucket's population (~7);
//   * a wave visits the chunk in that order and appends hits to its pixels' queues ([k][lane] in LDS, as point_sorted_kernel);
//     a pixel is done at K entries, a wave when its pixels are, the workgroup when its waves are.
// A bucket of more than 256 points (many equal depths), or a list longer than the pre-sort's LDS copy, makes the stream of that
// tile non-monotone from there on: its hits are INSERTED (per-lane shift in LDS), exact as ever.
// Measured (1M points, 512^2, profiles/r05/c16/, c17/): 3976 VALU per wave against 5333 (register queues + pre-sort; 11 531 in round
// 4), points_fine K = 10 0.108 against 0.114 ms, K = 16 0.229 against 0.251 -- both kernels wait 65-69 % of their wave cycles (seven
// barriers and a memory round trip per chunk, one image = one round of workgroups): instructions are no longer what binds them.
// (Tried on top, no gain and dropped: a chunk table computed once + the next chunk's loads requested a chunk ahead.)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kTileCap = 2048;  // list entries pre-sorted per tile by this kernel (LDS: 8 KB); longer lists: list order + insertion

struct TileSortLds {
  int sorted[kTileCap];
  int hist[kStage];
  int start[kStage];
  unsigned range[2];
  int wsum[kStage / kWave];
};

// The bucket pass: every thread requests its <= 8 list entries and their depths at once (two memory round trips for the list), one
// integer LDS atomic per point, a workgroup scan, a scatter: sorted[] = the list's point ids in bucket order; hist / start / range
// stay valid for the caller.  All 256 threads; count <= kTileCap.  Ends with a barrier.
__device__ __forceinline__ void bucket_sort_tile(const float* __restrict__ points, const int* __restrict__ list, int stride, int count,
                                                 TileSortLds& L, int tid) {
  constexpr int kSlots = kTileCap / kStage;
  const int lane = tid & 63, w = tid >> 6;
  const unsigned kBehind = 0x7f800000u;
  int pid[kSlots];
  unsigned zb[kSlots];
  // the depths: beside the ids when the binning left them there (BinCSR::stride == 2: ONE coalesced round trip for ids and
  // depths), else gathered from the points once the ids are here
  float zs[kSlots];
  if (stride == 2) {  // uniform
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      const int i = s * kStage + tid;
      pid[s] = -1;
      zs[s] = 0.0f;
      if (i < count) {
        const int2 e = reinterpret_cast<const int2*>(list)[i];
        pid[s] = e.x;
        zs[s] = __int_as_float(e.y);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      const int i = s * kStage + tid;
      pid[s] = i < count ? list[i] : -1;
    }
#pragma unroll
    for (int s = 0; s < kSlots; ++s) zs[s] = pid[s] >= 0 ? points[(int64_t)pid[s] * 3 + 2] : 0.0f;
  }
  unsigned lo = 0xffffffffu, hi = 0u;
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    zb[s] = kBehind;
    if (pid[s] >= 0) {
      const float z = zs[s];
      if (z >= 0.0f && z < INFINITY) {
        zb[s] = __float_as_uint(z + 0.0f);
        lo = zb[s] < lo ? zb[s] : lo;
        hi = zb[s] > hi ? zb[s] : hi;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned lo2 = (unsigned)__shfl_xor((int)lo, d), hi2 = (unsigned)__shfl_xor((int)hi, d);
    lo = lo2 < lo ? lo2 : lo;
    hi = hi2 > hi ? hi2 : hi;
  }
  L.hist[tid] = 0;
  if (tid == 0) {
    L.range[0] = 0xffffffffu;
    L.range[1] = 0u;
  }
  __syncthreads();
  if (lane == 0) {
    atomicMin(&L.range[0], lo);
    atomicMax(&L.range[1], hi);
  }
  __syncthreads();
  const float zlo = __uint_as_float(L.range[0]);
  const float span = __uint_as_float(L.range[1]) - zlo;
  const float scale = span > 0.0f && span < INFINITY ? 255.0f / span : 0.0f;
  int tag[kSlots];
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    tag[s] = 0;
    if (pid[s] >= 0) {
      int b = kStage - 1;
      if (zb[s] != kBehind) {
        b = (int)((__uint_as_float(zb[s]) - zlo) * scale);
        b = b < 0 ? 0 : (b > kStage - 1 ? kStage - 1 : b);
      }
      tag[s] = (b << 16) | atomicAdd(&L.hist[b], 1);
    }
  }
  __syncthreads();
  {
    const int c = L.hist[tid];
    int x = c;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == kWave - 1) L.wsum[w] = x;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int j = 0; j < kStage / kWave; ++j)
      if (j < w) before += L.wsum[j];
    L.start[tid] = before + x - c;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < kSlots; ++s)
    if (pid[s] >= 0) L.sorted[L.start[tag[s] >> 16] + (tag[s] & 0xffff)] = pid[s];
  __syncthreads();
}

// the bucket a depth falls into under bucket_sort_tile's scale (the same expression: the same bucket)
__device__ __forceinline__ int tile_bucket_of(unsigned zbits, float zlo, float scale) {
  const int b = (int)((__uint_as_float(zbits) - zlo) * scale);
  return b < 0 ? 0 : (b > kStage - 1 ? kStage - 1 : b);
}

template <bool BINNED>
__global__ __launch_bounds__(kStage, 2) void point_tile_sorted_kernel(PointArgs a) {
  extern __shared__ __align__(16) unsigned long long s_queues[];  // [wave][k][lane]
  __shared__ TileSortLds s_ts;
  __shared__ float4 s_pt[kStage];               // staged point: x, y, z, r (compaction order = bucket order)
  __shared__ unsigned long long s_key[kStage];  // its key (z bits | index)
  __shared__ int s_bk[kStage];                  // its bucket
  __shared__ int s_ord[kStage];                 // staged index of the entry at exact position i
  __shared__ int s_wcnt[kStage / kWave];
  __shared__ int s_cut[2];                      // [0] end of the chunk (list position), [1] largest bucket population of the tile

  if (a.overflow != nullptr && (*a.overflow != 0) == BINNED) return;  // uniform (scalar load)
  TileCoord tc;
  if (!tile_of_block(a.tm, blockIdx.x, &tc)) return;
  const int n = tc.n, by = tc.by, bx = tc.bx, H = a.H, W = a.W, K = a.K;
  const int y_end = min(H, (by + 1) * a.tm.bin_size), x_end = min(W, (bx + 1) * a.tm.bin_size);
  const int ty0 = by * a.tm.bin_size + tc.ty * kTile, tx0 = bx * a.tm.bin_size + tc.tx * kTile;
  if (ty0 >= y_end || tx0 >= x_end) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int sy0 = ty0 + (w >> 1) * 8, sx0 = tx0 + (w & 1) * 8;
  const int yi = sy0 + (lane >> 3), xi = sx0 + (lane & 7);
  const bool pix_ok = yi < y_end && xi < x_end;
  const bool wave_ok = sy0 < y_end && sx0 < x_end;
  const float pxy = (lane & 16) ? pix_to_ndc(ty0 + (lane & 15), H, W) : pix_to_ndc(tx0 + (lane & 15), W, H);
  auto col_centre = [&](int x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pxy), x - tx0)); };
  auto row_centre = [&](int y) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pxy), 16 + y - ty0)); };
  const float xf = __int_as_float(__builtin_amdgcn_ds_bpermute((xi - tx0) << 2, __float_as_int(pxy)));
  const float yf = __int_as_float(__builtin_amdgcn_ds_bpermute((16 + yi - ty0) << 2, __float_as_int(pxy)));
  const float tile_x0 = col_centre(tx0), tile_x1 = col_centre(min(tx0 + kTile, x_end) - 1);
  const float tile_y0 = row_centre(ty0), tile_y1 = row_centre(min(ty0 + kTile, y_end) - 1);
  const float sub_x0 = col_centre(min(sx0, x_end - 1)), sub_x1 = col_centre(max(min(sx0 + 8, x_end) - 1, tx0));
  const float sub_y0 = row_centre(min(sy0, y_end - 1)), sub_y1 = row_centre(max(min(sy0 + 8, y_end) - 1, ty0));

  int64_t src_base;
  int count;
  if (BINNED) {
    const int64_t row = ((int64_t)n * a.tm.BH + by) * a.tm.BW + bx;
    src_base = a.csr.offset[row];
    count = a.csr.total[row];
  } else {
    src_base = a.first[n];
    count = (int)a.count[n];
  }
  unsigned long long* queue = s_queues + (size_t)w * K * kWave;
  int cnt = 0;
  unsigned long long kth = ~0ull;
  // the stream is in queue order (hits can be appended) while every chunk's keys sort after the previous chunk's
  const bool sorted_list = BINNED && count <= kTileCap;
  bool monotone = sorted_list;
  float zlo = 0.0f, scale = 0.0f;
  if (sorted_list && count > 0) {
    bucket_sort_tile(a.points, a.csr.list + src_base * a.csr.stride, a.csr.stride, count, s_ts, tid);
    zlo = __uint_as_float(s_ts.range[0]);
    const float span = __uint_as_float(s_ts.range[1]) - zlo;
    scale = span > 0.0f && span < INFINITY ? 255.0f / span : 0.0f;
    int hmax = s_ts.hist[tid];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) hmax = max(hmax, __shfl_xor(hmax, d));
    if (tid == 0) s_cut[1] = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&s_cut[1], hmax);
    __syncthreads();
  }
  const int window = sorted_list && count > 0 ? s_cut[1] : kStage;  // same-bucket entries sit within this distance of each other

  int lo_pos = 0;
  while (lo_pos < count) {  // uniform
    // ---- the chunk [lo_pos, hi_pos): whole buckets that fit 256 entries ----
    int hi_pos = min(lo_pos + kStage, count);
    if (sorted_list && hi_pos < count) {
      if (tid == 0) s_cut[0] = lo_pos;
      __syncthreads();
      const int st = s_ts.start[tid];  // bucket tid begins here: a legal cut if inside (lo_pos, lo_pos + 256]
      if (st > lo_pos && st <= lo_pos + kStage) atomicMax(&s_cut[0], st);
      __syncthreads();
      const int cut = s_cut[0];
      if (cut > lo_pos)
        hi_pos = cut;
      else
        monotone = false;  // one bucket holds more than 256 points: it is cut in two, in arrival order
      __syncthreads();
    }
    // ---- stage it: tile cull, ordered compaction ----
    const int i = lo_pos + tid;
    bool keep = false;
    float px = 0.f, py = 0.f, pz = 0.f, r = 0.f;
    int pid = -1;
    if (i < hi_pos) {
      pid = !BINNED ? (int)(src_base + i) : (sorted_list ? s_ts.sorted[i] : a.csr.list[(src_base + i) * a.csr.stride]);
      const float* g = a.points + (int64_t)pid * 3;
      px = g[0];
      py = g[1];
      pz = g[2];
      r = a.radius[pid];
      const bool off_tile = tile_x0 > px + r || tile_x1 < px - r || tile_y0 > py + r || tile_y1 < py - r;
      keep = !(pz < 0.0f) && !off_tile;
    }
    const unsigned long long km = __ballot(keep);
    if (lane == 0) s_wcnt[w] = __popcll(km);
    __syncthreads();
    int pos = mask_rank(km);
    int staged = 0;
#pragma unroll
    for (int j = 0; j < kStage / kWave; ++j) {
      const int c = s_wcnt[j];
      if (j < w) pos += c;
      staged += c;
    }
    unsigned long long key = 0;
    int bk = 0;
    if (keep) {
      const unsigned zb = __float_as_uint(pz + 0.0f);  // +0.0: a zero depth's key orders by its bits
      key = ((unsigned long long)zb << 32) | (unsigned)pid;
      // the bucket the pre-sort (bucket_sort_tile) gave this point: depths that are not finite non-negative numbers -- a NaN passes
      // `!(pz < 0)` above, as in the reference -- sit in its last bucket, and the exact order below counts inversions inside a bucket:
      // with another bucket number here two entries could take one position
      bk = sorted_list ? ((pz >= 0.0f && pz < INFINITY) ? tile_bucket_of(zb, zlo, scale) : kStage - 1) : 0;
      s_pt[pos] = make_float4(px, py, pz + 0.0f, r);
      s_key[pos] = key;
      s_bk[pos] = bk;
    }
    __syncthreads();
    // ---- exact order inside the chunk: move by the same-bucket inversions around the entry ----
    if (keep) {
      int fin = pos;
      const int reach = window < staged ? window : staged;
      for (int t = 1; t < reach; ++t) {  // (a bucket's entries are adjacent; its population bounds the distance)
        const int jb = pos - t, ja = pos + t;
        if (jb >= 0 && s_bk[jb] == bk && s_key[jb] > key) --fin;
        if (ja < staged && s_bk[ja] == bk && s_key[ja] < key) ++fin;
      }
      s_ord[fin] = pos;
    }
    __syncthreads();
    // ---- visit: front to back ----
    if (wave_ok) {
      for (int jb = 0; jb < staged; jb += kWave) {
        const int j = jb + lane;
        int oj = 0;
        unsigned long long ck = ~0ull;
        bool touch = false;
        if (j < staged) {
          oj = s_ord[j];
          ck = s_key[oj];
          const float4 b = s_pt[oj];
          touch = !(sub_x0 > b.x + b.w || sub_x1 < b.x - b.w || sub_y0 > b.y + b.w || sub_y1 < b.y - b.w);
        }
        if (monotone) {
          // the block's first key is its smallest, and everything later in the tile sorts after it
          const unsigned long long k0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ck >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ck);
          if (__ballot(pix_ok && k0 < kth) == 0) break;  // uniform
        }
        unsigned long long cand = __ballot(touch);
        while (cand) {  // uniform
          const int cl = __builtin_ctzll(cand);
          cand &= cand - 1;
          const int jj = __builtin_amdgcn_readlane(oj, cl);
          const unsigned long long ckey = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(ck >> 32), cl) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)(unsigned)ck, cl);
          const bool live = pix_ok && ckey < kth;
          if (__ballot(live) == 0) {
            if (monotone) break;  // keys ascend: so do the misses
            continue;
          }
          const float4 pt = s_pt[jj];
          const float dx = xf - pt.x;
          const float dy = yf - pt.y;
          const float dist2 = dx * dx + dy * dy;
          if (live && dist2 < pt.w * pt.w) {
            if (monotone) {  // uniform
              queue[cnt * kWave + lane] = ckey;
              ++cnt;
              if (cnt == K) kth = ckey;
            } else {
              int at = cnt < K ? cnt : K - 1;
              while (at > 0) {
                const unsigned long long prev = queue[(at - 1) * kWave + lane];
                if (!(prev > ckey)) break;
                queue[at * kWave + lane] = prev;
                --at;
              }
              queue[at * kWave + lane] = ckey;
              if (cnt < K) ++cnt;
              if (cnt == K) kth = queue[(K - 1) * kWave + lane];
            }
          }
        }
      }
    }
    lo_pos = hi_pos;
    // done when every pixel of the tile is: in a monotone stream a full queue takes nothing more
    const bool wave_done = !wave_ok || __ballot(pix_ok && kth == ~0ull) == 0;
    if (__syncthreads_and((monotone && wave_done) ? 1 : 0)) break;  // uniform (also the barrier before the next staging)
  }

  if (pix_ok) {
    const int64_t pixel = ((int64_t)n * H + (H - 1 - yi)) * W + (W - 1 - xi);
    const int64_t base = pixel * K;
    auto entry = [&](int k, int* id, float* z, float* d2) {
      *id = -1;
      *z = -1.0f;
      *d2 = -1.0f;
      if (k < cnt) {
        const unsigned long long ekey = queue[k * kWave + lane];
        *id = (int)(unsigned)ekey;
        *z = __uint_as_float((unsigned)(ekey >> 32));
        const float* g = a.points + (int64_t)*id * 3;
        const float dx = xf - g[0];
        const float dy = yf - g[1];
        *d2 = dx * dx + dy * dy;
      }
    };
    const bool splat = a.features != nullptr;  // uniform
    SplatPixel sp;
    sp.init(a.comp_mode);
    if (splat && sp.norm_mode) {  // uniform: the weights' sum first (the entries are in LDS, their points in the L1)
      for (int k = 0; k < K; ++k) {
        int id;
        float z, d2;
        entry(k, &id, &z, &d2);
        sp.weigh(id, d2, a.inv_r2);
      }
      sp.clamp_norm();
    }
    if ((K & 1) == 0) {
      for (int k = 0; k < K; k += 2) {
        int id[2];
        float z[2], d2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) entry(k + j, &id[j], &z[j], &d2[j]);
        *reinterpret_cast<int2*>(a.idxs + base + k) = make_int2(id[0], id[1]);
        *reinterpret_cast<float2*>(a.zbuf + base + k) = make_float2(z[0], z[1]);
        *reinterpret_cast<float2*>(a.dists + base + k) = make_float2(d2[0], d2[1]);
        if (splat) {
          sp.add(a.features, a.C, id[0], d2[0], a.inv_r2);
          sp.add(a.features, a.C, id[1], d2[1], a.inv_r2);
        }
      }
    } else {
      for (int k = 0; k < K; ++k) {
        int id;
        float z, d2;
        entry(k, &id, &z, &d2);
        a.idxs[base + k] = id;
        a.zbuf[base + k] = z;
        a.dists[base + k] = d2;
        if (splat) sp.add(a.features, a.C, id, d2, a.inv_r2);
      }
    }
    if (splat) sp.store(a.images + pixel * a.C, a.C);
  }
}

// The compositor as a pass of its own over finished fragments: what p3d_rasterize_points_composite launches behind the rasterizer
// kernels that do not carry it in their epilogue (K beyond the tile-sorted kernel, naive launches, short workspaces).
__global__ __launch_bounds__(256) void splat_composite_kernel(PointArgs a) {
  // short workspaces: the binned kernel carried the compositor in its epilogue and wrote the image unless the lists did not fit
  if (a.overflow != nullptr && *a.overflow == 0) return;  // uniform (scalar load)
  const int64_t npix = (int64_t)a.N * a.H * a.W;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < npix; t += (int64_t)gridDim.x * 256) {
    SplatPixel sp;
    sp.init(a.comp_mode);
    if (sp.norm_mode) {  // uniform
      for (int k = 0; k < a.K; ++k) sp.weigh(a.idxs[t * a.K + k], a.dists[t * a.K + k], a.inv_r2);
      sp.clamp_norm();
    }
    for (int k = 0; k < a.K; ++k) sp.add(a.features, a.C, a.idxs[t * a.K + k], a.dists[t * a.K + k], a.inv_r2);
    sp.store(a.images + t * a.C, a.C);
  }
}

// Which kernel (1M points, 512^2, r = 0.01, points_fine ms; profiles/r05/c5, c16): register queues + a whole-tile depth pre-sort (the
// form of mid-round 5, superseded by the tile-sorted kernel: K = 10 0.112 -> 0.108, K = 16 0.259 -> 0.229) / sorted kernel
//   K = 1 0.045 / 0.136, 4 0.067 / 0.156, 8 0.098 / 0.186, 10 0.112 / 0.208, 12 0.186 / 0.213, 16 0.259 / 0.291, 24 0.656 / 0.361,
//   32 0.603 / 0.481 -- and beyond 32 (round-4 register / private-memory queues) 50 1.57 / 0.73, 64 2.01 / 0.97, 100 3.0 / 1.35,
//   150 15.4 / 2.63.  (Round 4, no pre-sort: K = 1 0.079, 8 0.152, 10 0.180, 16 0.334.)  The register queues below now serve the
// NAIVE launch only (K <= 16; one list per image: test sizes, and the stand-by of short workspaces).
constexpr int kQueueMaxK = 16;

#define P3D_COMMA ,
template <bool BINNED>
int launch_point_raster_queues(const PointArgs& a, hipStream_t stream) {
  const unsigned grid = tile_grid(a.tm);
  LaunchScope ls(BINNED ? "points_fine" : "points_naive", stream);
  const int K = a.K;
  // The common capacities as payload-free pair queues with one 64-bit key compare per entry (topk.h: TopKPairs<KT, true, 0>;
  // the distance is recomputed at the store; staged depths are >= +0).  Other K take the register queue of the next capacity.
#define P3D_PQ(KT_, WAVES_) \
  point_raster_kernel<TopKPairs<KT_ P3D_COMMA true P3D_COMMA 0>, KT_, true, BINNED, false, WAVES_, true><<<grid, kStage, 0, stream>>>(a)
  switch (K) {
    case 8: P3D_PQ(8, 2); return launch_status();
    case 10: P3D_PQ(10, 2); return launch_status();
    case 16: P3D_PQ(16, 2); return launch_status();
    default: break;
  }
#undef P3D_PQ
  if (K == 1)
    point_raster_kernel<TopKReg<1, 1>, 1, true, BINNED><<<grid, kStage, 0, stream>>>(a);
  else if (K == 2)
    point_raster_kernel<TopKReg<2, 1>, 2, true, BINNED><<<grid, kStage, 0, stream>>>(a);
  else if (K <= 4)
    point_raster_kernel<TopKReg<4, 1>, 4, true, BINNED><<<grid, kStage, 0, stream>>>(a);
  else if (K <= 8)
    point_raster_kernel<TopKReg<8, 1>, 8, true, BINNED><<<grid, kStage, 0, stream>>>(a);
  else if (K <= 10)  // the insertion is the kernel's dominant VALU cost and scales with the queue length
    point_raster_kernel<TopKReg<10, 1>, 10, true, BINNED><<<grid, kStage, 0, stream>>>(a);
  else if (K <= 12)
    point_raster_kernel<TopKReg<12, 1>, 12, true, BINNED><<<grid, kStage, 0, stream>>>(a);
  else
    point_raster_kernel<TopKReg<16, 1>, 16, true, BINNED><<<grid, kStage, 0, stream>>>(a);
  return launch_status();
}

// Binned launches with K <= 28 run the tile-sorted kernel, above that the single-wave sorted kernel: the tile-sorted kernel's four
// queues cost K x 2 KB of LDS per workgroup, and one image needs four workgroups per CU to run as one round (1M points, 512^2,
// points_fine ms, tile-sorted / sorted, profiles/r05/c20/): K = 16 0.227 / 0.291, 20 0.257 / 0.337, 24 0.301 / 0.367, 28 0.352 / 0.452,
// 32 0.604 / 0.485, 48 0.82 / 0.64, 64 1.00 / 0.97.  The naive launch (one list per image, far beyond the pre-sort's LDS copy)
// keeps the register queues up to K = 16.
constexpr int kTileSortedMaxK = 28;

template <bool BINNED>
int launch_point_raster(const PointArgs& a, hipStream_t stream) {
  if constexpr (BINNED) {
    if (a.K <= kTileSortedMaxK) {
      LaunchScope ls("points_fine", stream);
      const size_t dyn = (size_t)(kStage / kWave) * a.K * kWave * sizeof(unsigned long long);
      if (dyn > 48 * 1024 &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(&point_tile_sorted_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)dyn) != hipSuccess)
        return P3D_ERR_LAUNCH;
      point_tile_sorted_kernel<true><<<tile_grid(a.tm), kStage, dyn, stream>>>(a);
      return launch_status();
    }
  }
  if constexpr (!BINNED) {
    if (a.K <= kQueueMaxK) return launch_point_raster_queues<false>(a, stream);
  }
  const size_t grid = (size_t)tile_grid(a.tm) * 4;  // one single-wave workgroup per 8x8 sub-tile
  if (grid > 0x7fffffffull) return P3D_ERR_INVALID_ARG;
  const size_t dyn = (size_t)a.K * kWave * sizeof(unsigned long long);
  if (dyn > 48 * 1024) {
    // K > 96: past the default dynamic-LDS limit (gfx950 has 160 KB per CU).  Set on every such launch, not cached: the attribute
    // belongs to the function ON THE CURRENT DEVICE, and launchers are called from several host threads / devices
    // (nn.DataParallel, tests/test_render_multigpu.py:171) -- no global mutable state in this library (INTEGRATION.md "Contract")
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&point_sorted_kernel<BINNED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            P3D_MAX_K * kWave * (int)sizeof(unsigned long long)) != hipSuccess)
      return P3D_ERR_LAUNCH;
  }
  LaunchScope ls(BINNED ? "points_fine" : "points_naive", stream);
  point_sorted_kernel<BINNED><<<(unsigned)grid, kWave, dyn, stream>>>(a);
  return launch_status();
}

void set_tiles(PointArgs* a, int bin_size, int BH, int BW) { a->tm = make_tile_map(a->N, a->H, a->W, bin_size, BH, BW, true); }

// CUDA tie order for points (p3d_rasterize_points_cuda_order; see raster_mesh.hip: mesh_cuda_order_kernel for the meshes).  The
// reference's point kernels keep the same unsorted array as its mesh kernels (rasterize_points.cu:38-84) but sort it by depth
// ALONE at the end (rasterize_points.cu:26-28, a stable bubble sort): where points tie exactly in depth both the survivors at the
// K-th place and the order of the tied entries follow the array positions.  The replay below re-runs that procedure, points in
// ascending index (the lists are binned in order for it), for every pixel whose K slots came out full -- a pixel with fewer hits
// has dropped nothing and holds its hits in ascending (depth, index), which is the reference's arrival order.
template <bool BINNED>
__global__ __launch_bounds__(kStage) void point_cuda_order_kernel(PointArgs a) {
  if (a.overflow != nullptr && (*a.overflow != 0) == BINNED) return;  // short workspaces: as in point_raster_kernel
  TileCoord tc;
  if (!tile_of_block(a.tm, blockIdx.x, &tc)) return;
  const int n = tc.n, H = a.H, W = a.W, K = a.K;
  const int y_end = min(H, (tc.by + 1) * a.tm.bin_size), x_end = min(W, (tc.bx + 1) * a.tm.bin_size);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int yi = tc.by * a.tm.bin_size + tc.ty * kTile + (w >> 1) * 8 + (lane >> 3);
  const int xi = tc.bx * a.tm.bin_size + tc.tx * kTile + (w & 1) * 8 + (lane & 7);
  const bool pix_ok = yi < y_end && xi < x_end;
  const int64_t base = (((int64_t)n * H + (H - 1 - yi)) * W + (W - 1 - xi)) * K;
  const bool replay = pix_ok && K > 0 && a.idxs[base + (K - 1)] >= 0;
  if (__ballot(replay) == 0) return;  // uniform
  int64_t src;
  int count;
  if (BINNED) {
    const int64_t row = ((int64_t)n * a.tm.BH + tc.by) * a.tm.BW + tc.bx;
    src = a.csr.offset[row];
    count = a.csr.total[row];
  } else {
    src = a.first[n];
    count = (int)a.count[n];
  }
  const float xf = pix_to_ndc(xi, W, H), yf = pix_to_ndc(yi, H, W);
  float qz[P3D_MAX_K];
  int qi[P3D_MAX_K];
  int qn = 0, qmax_i = -1;
  float qmax_z = -1000.0f;
  for (int i = 0; i < count; ++i) {
    const int pid = BINNED ? a.csr.list[(src + i) * a.csr.stride] : (int)(src + i);  // uniform
    const float* g = a.points + (int64_t)pid * 3;
    const float pz = g[2];
    if (pz < 0.0f) continue;  // uniform
    const float r = a.radius[pid];
    const float dx = xf - g[0], dy = yf - g[1];
    const float dist2 = dx * dx + dy * dy;
    if (!replay || !(dist2 < r * r)) continue;
    if (qn < K) {
      qz[qn] = pz;
      qi[qn] = pid;
      if (pz > qmax_z) {
        qmax_z = pz;
        qmax_i = qn;
      }
      ++qn;
    } else if (pz < qmax_z) {
      qz[qmax_i] = pz;
      qi[qmax_i] = pid;
      qmax_z = pz;
      for (int j = 0; j < K; ++j) {
        if (qz[j] > qmax_z) {
          qmax_z = qz[j];
          qmax_i = j;
        }
      }
    }
  }
  if (!replay) return;
  for (int i = 1; i < qn; ++i) {  // stable, by depth alone
    const float z = qz[i];
    const int id = qi[i];
    int j = i - 1;
    while (j >= 0 && qz[j] > z) {
      qz[j + 1] = qz[j];
      qi[j + 1] = qi[j];
      --j;
    }
    qz[j + 1] = z;
    qi[j + 1] = id;
  }
  for (int k = 0; k < K; ++k) {
    if (k < qn) {
      const float* g = a.points + (int64_t)qi[k] * 3;
      const float dx = xf - g[0], dy = yf - g[1];
      a.idxs[base + k] = qi[k];
      a.zbuf[base + k] = qz[k];
      a.dists[base + k] = dx * dx + dy * dy;
    } else {
      a.idxs[base + k] = -1;
      a.zbuf[base + k] = a.dists[base + k] = -1.0f;
    }
  }
}

// Backward (rasterize_points.cu:366-411): every (pixel, k) entry adds (2*gd*dx, 2*gd*dy, gz) to its point.  A point of
// radius r is hit by ~pi*r^2 neighbouring pixels, so a wave owns an 8x8 pixel tile, walks its 64*K entries in
// memory order (rows of 8*K contiguous entries) and merges per point in a wave-private LDS table (wave_table.h):
// one global atomic triple per (tile, point) instead of one per entry.
using PointTable = WaveTable<3, 426>;  // 4 waves x 426 x 24 B = 40 KB

__global__ __launch_bounds__(256) void point_backward_kernel(const float* __restrict__ points,
                                                             const int32_t* __restrict__ idxs,
                                                             const float* __restrict__ grad_zbuf,
                                                             const float* __restrict__ grad_dists, int N, int H, int W,
                                                             int K, int tiles_y, int tiles_x, float* __restrict__ grad_points) {
  __shared__ __align__(16) int s_table[4][PointTable::kLdsInts];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t tile = (int64_t)blockIdx.x * 4 + w;
  const int64_t per_image = (int64_t)tiles_y * tiles_x;
  if (tile >= (int64_t)N * per_image) return;  // wave-uniform; no workgroup barrier in this kernel
  const int n = (int)(tile / per_image);
  const int t = (int)(tile - (int64_t)n * per_image);
  const int y0 = (t / tiles_x) * 8, x0 = (t % tiles_x) * 8;
  const int rows = min(8, H - y0), cols = min(8, W - x0);
  const int run = cols * K;         // contiguous entries per tile row
  const int total = rows * run;
  const float inv_run = 1.0f / (float)run, inv_k = 1.0f / (float)K;
  PointTable tab;
  tab.init(s_table[w], lane);
  // lane l < 8: NDC x of the tile's pixel column x0 + l; lane 8 + l: NDC y of its row y0 + l (rasterize_points.cu:389-393:
  // the backward un-flips the stored indices); once per wave
  float centres = (lane & 8) ? pix_to_ndc(H - 1 - (y0 + (lane & 7)), H, W) : pix_to_ndc(W - 1 - (x0 + (lane & 7)), W, H);
  asm volatile("" : "+v"(centres));
  // Three steps in flight (round 5): a step's chain is idx -> point gather -> table; with one step at a time the wave sat out
  // two memory round trips per 64 entries (0.078 ms for 55 MB on BASELINE configs[3], the wave waiting or stalled 88 % of its
  // cycles).  Now, while step s is summed, step s + 1's point gather (its index arrived during step s - 1) and step s + 2's
  // index and upstream gradients (which need no index) are on their way.
  struct Step {
    int p, xo, yo;
    float gd, gz, qx, qy;
  };
  auto fetch_idx = [&](int base) {  // the index and the two gradients of the lane's entry of the step that starts at `base`
    Step st;
    st.p = -1;
    st.gd = st.gz = st.qx = st.qy = 0.f;
    const int e = base + lane;
    // exact for these small operands: (e + 0.5) / d is at least 0.5 / d away from an integer
    const int r = (int)(((float)e + 0.5f) * inv_run);
    const int ee = e - r * run;
    st.xo = x0 + (int)(((float)ee + 0.5f) * inv_k);
    st.yo = y0 + r;
    if (e < total) {
      const int64_t i = (((int64_t)n * H + st.yo) * W + x0) * K + ee;
      st.p = idxs[i];
      st.gd = grad_dists[i];
      st.gz = grad_zbuf[i];
    }
    return st;
  };
  auto fetch_point = [&](Step* st) {
    if (st->p >= 0) {
      st->qx = points[(int64_t)st->p * 3 + 0];
      st->qy = points[(int64_t)st->p * 3 + 1];
    }
  };
  Step s0 = fetch_idx(0);
  Step s1 = fetch_idx(64);  // (past the end of the tile: no loads, p = -1)
  fetch_point(&s0);
  for (int base = 0; base < total; base += 64) {
    const Step cur = s0;
    fetch_point(&s1);
    const Step s2 = fetch_idx(base + 128);
    s0 = s1;
    s1 = s2;
    float g[3] = {0.f, 0.f, 0.f};
    // the pixel's centre from the lanes that hold the tile's eight column / row centres (two ds_bpermute instead of two
    // pix_to_ndc with an IEEE division each per entry); outside the branch: every lane takes part in the exchange, lanes
    // past the end of the tile read some lane of the table and never use it
    const float xf = __int_as_float(__builtin_amdgcn_ds_bpermute(((cur.xo - x0) & 7) << 2, __float_as_int(centres)));
    const float yf = __int_as_float(__builtin_amdgcn_ds_bpermute((8 + ((cur.yo - y0) & 7)) << 2, __float_as_int(centres)));
    if (cur.p >= 0) {
      const float dx = cur.qx - xf;
      const float dy = cur.qy - yf;
      g[0] = 2.0f * cur.gd * dx;
      g[1] = 2.0f * cur.gd * dy;
      g[2] = cur.gz;
    }
    tab.add(grad_points, lane, cur.p, g);
  }
  if (tab.used > 0) tab.flush(grad_points, lane);
}

int check_common(int N, int H, int W, int K) {
  if (N < 0 || H < 0 || W < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (K > P3D_MAX_K) return P3D_ERR_K_TOO_LARGE;
  return P3D_OK;
}

void fill_args(PointArgs* a, const float* points, const float* radius, int N, int H, int W, int K, int32_t* idxs,
               float* zbuf, float* dists) {
  a->points = points;
  a->radius = radius;
  a->N = N;
  a->H = H;
  a->W = W;
  a->K = K;
  a->idxs = idxs;
  a->zbuf = zbuf;
  a->dists = dists;
}

}  // namespace
}  // namespace p3d

using namespace p3d;

P3D_API size_t p3d_rasterize_points_workspace_bytes(int64_t P, int N, int H, int W, int bin_size,
                                                    int max_points_per_bin) {
  if (bin_size <= 0 || max_points_per_bin <= 0 || N <= 0 || H <= 0 || W <= 0) return 0;
  const size_t user = bin_workspace_bytes(P, N, make_geom(H, W, bin_size), max_points_per_bin);
  const size_t internal = bin_workspace_bytes(P, N, make_internal_geom(H, W, bin_size), max_points_per_bin, -1, /*with_z=*/true);
  return (user > internal ? user : internal) + 256;
}

P3D_API size_t p3d_rasterize_points_short_workspace_bytes(int64_t P, int N, int H, int W, int bin_size, int max_points_per_bin,
                                                          int64_t list_entries) {
  if (bin_size <= 0 || max_points_per_bin <= 0 || N <= 0 || H <= 0 || W <= 0 || list_entries < 0) return 0;
  return bin_workspace_bytes(P, N, make_internal_geom(H, W, bin_size), max_points_per_bin, list_entries, /*with_z=*/true) + 256;
}

P3D_API size_t p3d_rasterize_points_workspace_need_offset(int64_t P, int N, int H, int W, int bin_size, int max_points_per_bin) {
  if (bin_size <= 0 || max_points_per_bin <= 0 || N <= 0 || H <= 0 || W <= 0) return 0;
  const BinGeom g = make_internal_geom(H, W, bin_size);
  Arena probe(nullptr, 0);
  BinWorkspace ws;
  bin_carve(probe, P, N, g, max_points_per_bin, &ws, 1, /*with_z=*/true);
  return ws.need_at;
}

P3D_API int p3d_rasterize_points_naive(const float* points, const int64_t* first, const int64_t* count,
                                       const float* radius, int64_t P, int N, int H, int W, int K, int32_t* idxs,
                                       float* zbuf, float* dists, p3d_stream_t stream) {
  const int rc = check_common(N, H, W, K);
  if (rc != P3D_OK) return rc;
  if ((int64_t)N * H * W * K == 0) return P3D_OK;
  if ((P > 0 && (!points || !radius)) || !first || !count || !idxs || !zbuf || !dists) return P3D_ERR_INVALID_ARG;
  PointArgs a{};
  fill_args(&a, points, radius, N, H, W, K, idxs, zbuf, dists);
  a.first = first;
  a.count = count;
  set_tiles(&a, H > W ? H : W, 1, 1);
  return launch_point_raster<false>(a, (hipStream_t)stream);
}

static int point_cuda_order_replay(const PointArgs& fine, bool binned, const int64_t* first, const int64_t* count,
                                   const int* overflow, hipStream_t s) {
  PointArgs a = fine;
  a.overflow = overflow;
  LaunchScope ls("points_cuda_order", s);
  if (binned) {
    point_cuda_order_kernel<true><<<tile_grid(a.tm), kStage, 0, s>>>(a);
  } else {
    a.first = first;
    a.count = count;
    set_tiles(&a, a.H > a.W ? a.H : a.W, 1, 1);
    point_cuda_order_kernel<false><<<tile_grid(a.tm), kStage, 0, s>>>(a);
  }
  return launch_status();
}

struct SplatArgs {
  const float* features;
  float* images;
  int C;
  float inv_r2;
  int mode;
};

// only_if: device flag; the pass runs when it is set (null: always)
static int splat_composite_pass(const PointArgs& fine, const SplatArgs& sp, hipStream_t s, const int* only_if = nullptr) {
  PointArgs a = fine;
  a.overflow = only_if;
  a.features = sp.features;
  a.images = sp.images;
  a.C = sp.C;
  a.inv_r2 = sp.inv_r2;
  a.comp_mode = sp.mode;
  const int64_t npix = (int64_t)a.N * a.H * a.W;
  int64_t blocks = ceil_div(npix, 256);
  if (blocks > 256 * 32) blocks = 256 * 32;
  LaunchScope ls("points_composite", s);
  splat_composite_kernel<<<(unsigned)blocks, 256, 0, s>>>(a);
  return launch_status();
}

static int raster_points_impl(const float* points, const int64_t* first, const int64_t* count, const float* radius, int64_t P, int N,
                              int H, int W, int K, int bin_size, int max_points_per_bin, int32_t* idxs, float* zbuf, float* dists,
                              void* workspace, size_t workspace_bytes, p3d_stream_t stream, bool cuda_order,
                              const SplatArgs* splat = nullptr) {
  hipStream_t s = (hipStream_t)stream;
  if (splat && (bin_size <= 0 || max_points_per_bin <= 0)) {
    const int st = p3d_rasterize_points_naive(points, first, count, radius, P, N, H, W, K, idxs, zbuf, dists, stream);
    if (st != P3D_OK || (int64_t)N * H * W == 0) return st;
    PointArgs a{};
    fill_args(&a, points, radius, N, H, W, K, idxs, zbuf, dists);
    return splat_composite_pass(a, *splat, s);
  }
  if (bin_size <= 0 || max_points_per_bin <= 0) {
    const int st = p3d_rasterize_points_naive(points, first, count, radius, P, N, H, W, K, idxs, zbuf, dists, stream);
    if (st != P3D_OK || !cuda_order || (int64_t)N * H * W * K == 0) return st;
    PointArgs a{};
    fill_args(&a, points, radius, N, H, W, K, idxs, zbuf, dists);
    return point_cuda_order_replay(a, false, first, count, nullptr, s);
  }
  const int rc = check_common(N, H, W, K);
  if (rc != P3D_OK) return rc;
  if ((int64_t)N * H * W * K == 0) return P3D_OK;
  if ((P > 0 && (!points || !radius)) || !first || !count || !idxs || !zbuf || !dists) return P3D_ERR_INVALID_ARG;
  const BinGeom gu = make_geom(H, W, bin_size);
  if (gu.BH > P3D_MAX_BINS_PER_SIDE || gu.BW > P3D_MAX_BINS_PER_SIDE) return P3D_ERR_TOO_MANY_BINS;
  const BinGeom g = make_internal_geom(H, W, bin_size);  // tile-sized bins: results do not depend on the binning
  Arena arena(workspace, workspace_bytes);
  BinWorkspace ws;
  // a short workspace is welcome here (binning.h): the list takes what the caller gave, and the naive kernel stands by
  // (with_z: bin_fill leaves every entry's depth beside its id -- the tile-sorted kernel's depth sort reads them with the list)
  if (!workspace || !bin_carve(arena, P, N, g, max_points_per_bin, &ws, /*list_entries=*/1, /*with_z=*/true)) return P3D_ERR_WORKSPACE;
  const bool is_short = ws.capacity < ws.worst;
  const int* overflow = is_short ? ws.plan_hdr + 2 : nullptr;
  // the K nearest under (z, point index) do not depend on the order inside a bin: unordered fast binning (the replay of the
  // CUDA tie order walks the lists in ascending index: ordered binning then)
  int st = bin_build(kPoints, points, radius, first, count, P, N, g, max_points_per_bin, 0.0f, ws, s, /*ordered=*/cuda_order);
  if (st != P3D_OK) return st;
  PointArgs a{};
  fill_args(&a, points, radius, N, H, W, K, idxs, zbuf, dists);
  a.csr = BinCSR{ws.offset, ws.total, ws.list, TilePlan{ws.arank, ws.bg_list, ws.plan_hdr, ws.order}, ws.stride};
  a.overflow = overflow;
  set_tiles(&a, g.bin_size, g.BH, g.BW);
  // the compositor rides in the tile-sorted kernel's epilogue when that kernel writes every pixel; otherwise it is a pass behind
  // (short workspaces: the binned kernel returns at once when its lists did not fit; the naive kernel then writes the fragments and the
  // pass behind it, gated by the same device flag, the image)
  const bool splat_fused = splat && !cuda_order && K <= kTileSortedMaxK;
  if (splat_fused) {
    a.features = splat->features;
    a.images = splat->images;
    a.C = splat->C;
    a.inv_r2 = splat->inv_r2;
    a.comp_mode = splat->mode;
  }
  st = launch_point_raster<true>(a, s);
  if (splat) {
    if (st == P3D_OK && is_short) {
      PointArgs b{};
      fill_args(&b, points, radius, N, H, W, K, idxs, zbuf, dists);
      b.first = first;
      b.count = count;
      b.overflow = overflow;
      set_tiles(&b, H > W ? H : W, 1, 1);
      st = launch_point_raster<false>(b, s);
    }
    if (st != P3D_OK) return st;
    if (splat_fused && !is_short) return st;
    return splat_composite_pass(a, *splat, s, splat_fused ? overflow : nullptr);
  }
  if (st == P3D_OK && is_short) {
    PointArgs b{};
    fill_args(&b, points, radius, N, H, W, K, idxs, zbuf, dists);
    b.first = first;
    b.count = count;
    b.overflow = overflow;
    set_tiles(&b, H > W ? H : W, 1, 1);
    st = launch_point_raster<false>(b, s);
  }
  if (st != P3D_OK || !cuda_order) return st;
  st = point_cuda_order_replay(a, true, first, count, overflow, s);
  if (st != P3D_OK || !is_short) return st;
  return point_cuda_order_replay(a, false, first, count, overflow, s);
}

P3D_API int p3d_rasterize_points(const float* points, const int64_t* first, const int64_t* count, const float* radius,
                                 int64_t P, int N, int H, int W, int K, int bin_size, int max_points_per_bin,
                                 int32_t* idxs, float* zbuf, float* dists, void* workspace, size_t workspace_bytes,
                                 p3d_stream_t stream) {
  return raster_points_impl(points, first, count, radius, P, N, H, W, K, bin_size, max_points_per_bin, idxs, zbuf, dists, workspace,
                            workspace_bytes, stream, false);
}

P3D_API int p3d_rasterize_points_composite(int mode, const float* points, const int64_t* first, const int64_t* count,
                                           const float* radius, const float* features, int64_t P, int C, int N, int H, int W, int K,
                                           int bin_size, int max_points_per_bin, float inv_r2, int32_t* idxs, float* zbuf, float* dists,
                                           float* images, void* workspace, size_t workspace_bytes, p3d_stream_t stream) {
  if (C < 1 || C > 4 || (mode != P3D_COMPOSITE_ALPHA && mode != P3D_COMPOSITE_NORM_SUM)) return P3D_ERR_INVALID_ARG;
  const int rc = check_common(N, H, W, K);
  if (rc != P3D_OK) return rc;
  if ((int64_t)N * H * W == 0) return P3D_OK;
  if (!images || (P > 0 && !features)) return P3D_ERR_INVALID_ARG;
  if (K == 0 || P == 0) {  // nothing to composite: the fragments (K > 0: all empty) and a black image
    const int st = raster_points_impl(points, first, count, radius, P, N, H, W, K, bin_size, max_points_per_bin, idxs, zbuf, dists,
                                      workspace, workspace_bytes, stream, false);
    if (st != P3D_OK) return st;
    return hipMemsetAsync(images, 0, (size_t)N * H * W * C * sizeof(float), (hipStream_t)stream) == hipSuccess ? P3D_OK : P3D_ERR_LAUNCH;
  }
  const SplatArgs sp{features, images, C, inv_r2, mode};
  return raster_points_impl(points, first, count, radius, P, N, H, W, K, bin_size, max_points_per_bin, idxs, zbuf, dists, workspace,
                            workspace_bytes, stream, false, &sp);
}

P3D_API int p3d_rasterize_points_cuda_order(const float* points, const int64_t* first, const int64_t* count, const float* radius,
                                            int64_t P, int N, int H, int W, int K, int bin_size, int max_points_per_bin,
                                            int32_t* idxs, float* zbuf, float* dists, void* workspace, size_t workspace_bytes,
                                            p3d_stream_t stream) {
  return raster_points_impl(points, first, count, radius, P, N, H, W, K, bin_size, max_points_per_bin, idxs, zbuf, dists, workspace,
                            workspace_bytes, stream, true);
}

P3D_API int p3d_rasterize_points_coarse(const float* points, const int64_t* first, const int64_t* count,
                                        const float* radius, int64_t P, int N, int H, int W, int bin_size,
                                        int max_points_per_bin, int32_t* bin_points, void* workspace,
                                        size_t workspace_bytes, p3d_stream_t stream) {
  if (N < 0 || H <= 0 || W <= 0 || bin_size <= 0 || max_points_per_bin < 0) return P3D_ERR_INVALID_ARG;
  const BinGeom g = make_geom(H, W, bin_size);
  if (g.BH > P3D_MAX_BINS_PER_SIDE || g.BW > P3D_MAX_BINS_PER_SIDE) return P3D_ERR_TOO_MANY_BINS;
  if ((int64_t)N * g.nbins * max_points_per_bin == 0) return P3D_OK;
  if ((P > 0 && (!points || !radius)) || !first || !count || !bin_points) return P3D_ERR_INVALID_ARG;
  Arena arena(workspace, workspace_bytes);
  BinWorkspace ws;
  if (!workspace || !bin_carve(arena, P, N, g, max_points_per_bin, &ws)) return P3D_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int st = bin_build(kPoints, points, radius, first, count, P, N, g, max_points_per_bin, 0.0f, ws, s);
  if (st != P3D_OK) return st;
  return bin_expand_padded(ws, N, g, max_points_per_bin, bin_points, s);
}

P3D_API int p3d_rasterize_points_fine(const float* points, const int32_t* bin_points, const float* radius, int64_t P,
                                      int N, int BH, int BW, int M, int H, int W, int bin_size, int K, int32_t* idxs,
                                      float* zbuf, float* dists, void* workspace, size_t workspace_bytes,
                                      p3d_stream_t stream) {
  const int rc = check_common(N, H, W, K);
  if (rc != P3D_OK) return rc;
  if ((int64_t)N * H * W * K == 0) return P3D_OK;
  if (bin_size <= 0 || BH <= 0 || BW <= 0 || M < 0) return P3D_ERR_INVALID_ARG;
  if ((P > 0 && (!points || !radius)) || !idxs || !zbuf || !dists) return P3D_ERR_INVALID_ARG;
  if (BH > P3D_MAX_BINS_PER_SIDE || BW > P3D_MAX_BINS_PER_SIDE) return P3D_ERR_TOO_MANY_BINS;
  if ((int64_t)BH * bin_size < H || (int64_t)BW * bin_size < W) return P3D_ERR_INVALID_ARG;
  if (workspace_bytes < p3d_rasterize_fine_workspace_bytes(N, BH, BW, M) || !workspace) return P3D_ERR_WORKSPACE;
  if (M > 0 && !bin_points) return P3D_ERR_INVALID_ARG;
  const int64_t rows = (int64_t)N * BH * BW;
  Arena arena(workspace, workspace_bytes);
  int* list = arena.take<int>((size_t)rows * M);
  int* total = arena.take<int>((size_t)rows);
  int64_t* offset = arena.take<int64_t>((size_t)rows + 1);
  hipStream_t s = (hipStream_t)stream;
  int st = bin_compact_padded(bin_points, rows, M, list, total, offset, s);
  if (st != P3D_OK) return st;
  PointArgs a{};
  fill_args(&a, points, radius, N, H, W, K, idxs, zbuf, dists);
  a.csr = BinCSR{offset, total, list, TilePlan{nullptr, nullptr, nullptr, nullptr}};
  set_tiles(&a, bin_size, BH, BW);
  return launch_point_raster<true>(a, s);
}

P3D_API int p3d_rasterize_points_backward(const float* points, const int32_t* idxs, const float* grad_zbuf,
                                          const float* grad_dists, int64_t P, int N, int H, int W, int K,
                                          float* grad_points, p3d_stream_t stream) {
  if (P < 0 || N < 0 || H < 0 || W < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (P == 0) return P3D_OK;
  if (!grad_points || !points) return P3D_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_points, 0, (size_t)P * 3 * sizeof(float), s) != hipSuccess) return P3D_ERR_LAUNCH;
  const int64_t total = (int64_t)N * H * W * K;
  if (total == 0) return P3D_OK;
  if (!idxs || !grad_zbuf || !grad_dists) return P3D_ERR_INVALID_ARG;
  // (8 x 8 tiles also when one image leaves the chip with one wave per SIMD: with tiles of 4 / 2 / 1 rows BASELINE configs[3]
  // measured 0.077 -> 0.087 / 0.113 / 0.176 ms, profiles/r04/r04c15/point_bwd_th.txt -- the launch is bound by the flush
  // atomics, one triple per (tile, point), not by a wave's chain of round trips)
  const int tiles_y = (int)ceil_div(H, 8), tiles_x = (int)ceil_div(W, 8);
  const int64_t blocks = ceil_div((int64_t)N * tiles_y * tiles_x, 4);
  if (blocks > 0x7fffffff) return P3D_ERR_INVALID_ARG;
  LaunchScope ls("points_backward", s);
  point_backward_kernel<<<(unsigned)blocks, 256, 0, s>>>(points, idxs, grad_zbuf, grad_dists, N, H, W, K, tiles_y, tiles_x,
                                                        grad_points);
  return launch_status();
}
