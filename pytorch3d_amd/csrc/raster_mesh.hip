// raster_mesh.hip -- mesh rasterization for gfx950: fine / naive forward and SoftRas backward.
//
// Forward (replaces RasterizeMeshesFineCudaKernel and RasterizeMeshesNaiveCudaKernel,
// pytorch3d/csrc/rasterize_meshes/rasterize_meshes.cu:245-334, 630-736):
//   * one 256-thread workgroup per 16x16-pixel tile of one bin; the four waves own the four
//     8x8 sub-tiles, one pixel per lane;
//   * the bin's face list is streamed through LDS 256 faces at a time.  While staging, each
//     thread does the pixel-independent part of CheckPixelInsideFace ONCE per face (blur-expanded
//     bbox, zmax < 0, back-face, zero area, zmin < eps -- the reference redoes it per pixel) and
//     drops faces that cannot touch the tile; survivors are compacted in order (ballot + mbcnt);
//   * each wave then culls the staged faces against its own 8x8 sub-tile 64 faces at a time
//     (one lane per face, conflict-free LDS reads), and only the surviving faces are evaluated
//     per pixel, their vertex records read as LDS broadcasts;
//   * the per-(pixel, face) test runs on a per-face record with shared reciprocals (p3d_geom.h: FaceRec,
//     face_hit_rec -- bit-identical to the reference's twelve IEEE divisions, a third of their instructions);
//     faces with a clipped neighbour are handled by a second loop nest that a tile enters at most once;
//   * the per-pixel queue lives in VGPRs (topk.h); every output element, -1 padding included, is
//     written exactly once by the kernel, a pixel's K values as 16-byte stores; tiles without faces are
//     filled cooperatively in memory order;
//   * launches of few tiles (one image) run one workgroup per 8x8 sub-tile with the list dealt to its four
//     waves (SPLIT, see mesh_raster_kernel).
// The naive operator is the same kernel with "the bin" being the whole image and "the list"
// being the mesh's face range.  Naive and binned results agree by construction, which is why neither is
// tested against the other alone: both are compared with the C oracle and with the reference's own CPU and
// device kernels (tests/test_gpu_baseline_sizes.py, tests/test_gpu_vs_reference_device_kernels.py).
//
// The SoftRas backward lives in raster_mesh_bwd.hip.
#include "binning.h"
#include "tile_map.h"
#include "chunk_order.h"
#include "p3d_geom.h"
#include "topk.h"

#include <stdio.h>
#include <stdlib.h>


namespace p3d {

namespace {

constexpr int kTile = 16;       // pixels per tile side (4 waves x 8x8)
constexpr int kStage = 256;     // faces staged per round = threads per workgroup
constexpr int kMeshPayload = 4; // dist, bary.x, bary.y, bary.z
// Launches with at most this many 16x16 tiles (two per CU) use the split kernel (see mesh_raster_kernel).  Measured with
// bench meshes under SoftRas blur (full queues): 256 tiles 0.151 -> 0.084 ms, 1024 tiles 0.095 -> 0.115 ms (four queues
// per pixel cull later than one), 2048 tiles 0.094 -> 0.224 ms.
constexpr int kSplitMaxTiles = 512;

constexpr int kFineWaves = 4;  // waves per SIMD the fine kernels are built for: caps them at 128 VGPRs (measured: 3 cost 17 %)

struct MeshArgs {
  const float* face_verts;
  const int64_t* neighbor;
  const int64_t* mesh_first;
  const int64_t* mesh_count;
  BinCSR csr;
  int N, H, W, K;
  TileMap tm;
  float blur, sqrt_blur;
  int persp, clip, cull;
  int walk_plan;  // blocks take their tiles from csr.plan (active tiles longest-first, then background): launcher decides
  int64_t* p2f;
  float* zbuf;
  float* bary;
  float* dists;
  int* cover;  // row cover of the output (include/p3d_amd.h: p3d_rasterize_meshes_with_cover), zeroed by the launcher; or null
  int CY, CX;  // its 16 x 16 pixel blocks per image
  // p3d_rasterize_meshes_with_cover_list: the words of the cover that hold a face, appended by the wave that sets a word's first bit
  // (its atomicOr returns 0): area_count[0] entries of area_list, zeroed with the cover.  Null without.
  int* area_count;
  int* area_list;
  // ... or, when the tiles are the bins of a tile plan and coincide with the cover's words (image sides multiples of 16): every active
  // tile writes its word into slot arank[tile] of the list at its START (a plain store, no atomic, nothing to wait for) and the count is
  // the plan's number of active tiles -- a tile whose faces hit no pixel is listed too (the backward's wave finds no row and returns)
  int list_by_plan;
  // short workspaces (binning.h): the device flag "the lists did not fit", or null.  A binned launch returns at once when
  // it is up, the naive launch that follows it returns at once when it is not: exactly one of the two writes the output.
  const int* overflow;
  int ties;  // CUDA tie order (TIES kernels): pixels whose survivors may differ from the reference's CUDA procedure are marked for the replay
  // ... in place (-2 in the pixel's first pix_to_face entry) and, when the workspace has room, as one 64-bit lane mask per 8 x 8
  // sub-tile ((N, SY, SX) words, zeroed by the launcher): the replay then finds its pixels with one scalar load per wave
  unsigned long long* tie_words;
  int SY, SX;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
  u32x4 v = {a, b, c, d};
  *reinterpret_cast<u32x4*>(p) = v;  // (non-temporal stores measured in round 2: equal or slower, profiles/r02_fill_modes.txt)
}

template <int KT>
__device__ __forceinline__ void store_row(float* dst, const float (&v)[KT]) {
  if constexpr (KT % 4 == 0) {
#pragma unroll
    for (int k = 0; k < KT; k += 4) {
      float4 t;
      t.x = v[k];
      t.y = v[k + 1];
      t.z = v[k + 2];
      t.w = v[k + 3];
      *reinterpret_cast<float4*>(dst + k) = t;
    }
  } else if constexpr (KT == 2) {
    float2 t;
    t.x = v[0];
    t.y = v[1];
    *reinterpret_cast<float2*>(dst) = t;
  } else {
#pragma unroll
    for (int k = 0; k < KT; ++k) dst[k] = v[k];
  }
}

// One pixel's K = KT rows of the four outputs from the register queue: 16-byte stores.
template <typename Queue, int KT, bool IN_REGS>
__device__ __forceinline__ void write_pixel(const MeshArgs& a, const Queue& q, int64_t opix) {
  static_assert(IN_REGS, "vector-row stores need the register queue");
  const int64_t base = opix * KT;
  float zv[KT], dv[KT], bv[3 * KT];
  long long iv[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const bool ok = q.valid(k);
    iv[k] = ok ? (long long)q.ix(k) : -1ll;
    zv[k] = ok ? q.zf(k) : -1.0f;
    dv[k] = ok ? q.pay(0, k) : -1.0f;
    bv[3 * k + 0] = ok ? q.pay(1, k) : -1.0f;
    bv[3 * k + 1] = ok ? q.pay(2, k) : -1.0f;
    bv[3 * k + 2] = ok ? q.pay(3, k) : -1.0f;
  }
  if constexpr (KT % 4 == 0) {
#pragma unroll
    for (int k = 0; k < KT; k += 4) {
      store16(a.zbuf + base + k, __float_as_uint(zv[k]), __float_as_uint(zv[k + 1]), __float_as_uint(zv[k + 2]),
                  __float_as_uint(zv[k + 3]));
      store16(a.dists + base + k, __float_as_uint(dv[k]), __float_as_uint(dv[k + 1]), __float_as_uint(dv[k + 2]),
                  __float_as_uint(dv[k + 3]));
    }
#pragma unroll
    for (int k = 0; k < 3 * KT; k += 4)
      store16(a.bary + base * 3 + k, __float_as_uint(bv[k]), __float_as_uint(bv[k + 1]), __float_as_uint(bv[k + 2]),
                  __float_as_uint(bv[k + 3]));
#pragma unroll
    for (int k = 0; k < KT; k += 2)
      store16(a.p2f + base + k, (unsigned)iv[k], (unsigned)(iv[k] >> 32), (unsigned)iv[k + 1], (unsigned)(iv[k + 1] >> 32));
  } else {
    store_row<KT>(a.zbuf + base, zv);
    store_row<KT>(a.dists + base, dv);
#pragma unroll
    for (int k = 0; k < 3 * KT; ++k) a.bary[base * 3 + k] = bv[k];
    if constexpr (KT % 2 == 0) {
      long long* ip = reinterpret_cast<long long*>(a.p2f + base);
#pragma unroll
      for (int k = 0; k < KT; k += 2) {
        longlong2 t;
        t.x = iv[k];
        t.y = iv[k + 1];
        *reinterpret_cast<longlong2*>(ip + k) = t;
      }
    } else {
#pragma unroll
      for (int k = 0; k < KT; ++k) a.p2f[base + k] = iv[k];
    }
  }
}

// Output of one wave's 8x8 sub-tile when K has no vector-row path (K != KT, or the queue lives in memory).
// A lane writing its own pixel's K-row dword by dword stores 4 bytes at a stride of 4*K: every store is a partial
// line and the kernel spends its time there (measured at K = 100: 9.5 of 11.5 ms; K = 16: 1.2 of 1.7 ms).  Instead:
//   (A) all 64 lanes fill the sub-tile's rows of the four outputs with -1 in memory order (each instruction one
//       contiguous 256-byte piece) -- at most K of a row's entries are ever valid, usually far fewer;
//   (B) once those stores have been acknowledged, every lane patches in its pixel's valid entries.
// `have` = false: background, (A) only.
template <typename Queue, int KT, bool IN_REGS>
__device__ __forceinline__ void write_subtile_fill_patch(const MeshArgs& a, const Queue& q, bool have, int n, int sy0,
                                                         int sx0, int y_end, int x_end, int lane, bool pix_ok, int yi,
                                                         int xi) {
  const int H = a.H, W = a.W, K = a.K;
  const int rows = min(8, y_end - sy0), cols = min(8, x_end - sx0);
  const int seg = cols * K;             // contiguous entries per sub-tile row (outputs are stored flipped: x_out = W-1-x)
  const int64_t col0 = W - sx0 - cols;  // first output column of the sub-tile
  const bool patch = have;
  for (int r = 0; r < rows; ++r) {
    // ---- (A) fill row r of the sub-tile: one contiguous piece of each output ----
    const int64_t px = ((int64_t)n * H + (H - 1 - (sy0 + r))) * W + col0;
    if ((K & 3) == 0) {
      // every pixel's K-row starts on a 16-byte boundary in all four outputs: 16-byte stores
      const int q4 = seg >> 2;  // float4 pieces of this row in zbuf / dists; p2f has 2x, bary 3x as many
      const float4 m1 = make_float4(-1.0f, -1.0f, -1.0f, -1.0f);
      longlong2 i1;
      i1.x = -1;
      i1.y = -1;
      float4* zb = reinterpret_cast<float4*>(a.zbuf + px * K);
      float4* db = reinterpret_cast<float4*>(a.dists + px * K);
      float4* bb = reinterpret_cast<float4*>(a.bary + px * K * 3);
      longlong2* ib = reinterpret_cast<longlong2*>(a.p2f + px * K);
      for (int e = lane; e < 3 * q4; e += 64) {
        bb[e] = m1;
        if (e < 2 * q4) ib[e] = i1;
        if (e < q4) {
          zb[e] = m1;
          db[e] = m1;
        }
      }
    } else {
      for (int e = lane; e < 3 * seg; e += 64) {
        a.bary[px * K * 3 + e] = -1.0f;
        if (e < seg) {
          a.p2f[px * K + e] = -1;
          a.zbuf[px * K + e] = -1.0f;
          a.dists[px * K + e] = -1.0f;
        }
      }
    }
    if (!patch) continue;
    // ---- (B) the 8 lanes that own the pixels of row r patch in their valid entries, while the lines are still in
    // L2.  It must land after (A) although other lanes issued (A)'s stores to these addresses: wait until this wave's
    // stores have been acknowledged (same wave, same cache hierarchy: no cache write-back needed, only the counter).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!pix_ok || (lane >> 3) != r) continue;
    const int64_t base = (((int64_t)n * H + (H - 1 - yi)) * W + (W - 1 - xi)) * K;
    if constexpr (Queue::kPayload == 0) {
      // queues without payload are written by write_pixel_long; they come here for background tiles only (patch == false)
    } else if constexpr (IN_REGS) {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k < K && q.valid(k)) {
          a.p2f[base + k] = (int64_t)q.ix(k);
          a.zbuf[base + k] = q.zf(k);
          a.dists[base + k] = q.pay(0, k);
          a.bary[(base + k) * 3 + 0] = q.pay(1, k);
          a.bary[(base + k) * 3 + 1] = q.pay(2, k);
          a.bary[(base + k) * 3 + 2] = q.pay(3, k);
        }
      }
    } else {
      for (int k = 0; k < K; ++k) {
        if (!q.valid(k)) break;
        a.p2f[base + k] = (int64_t)q.ix(k);
        a.zbuf[base + k] = q.zf(k);
        a.dists[base + k] = q.pay(0, k);
        a.bary[(base + k) * 3 + 0] = q.pay(1, k);
        a.bary[(base + k) * 3 + 1] = q.pay(2, k);
        a.bary[(base + k) * 3 + 2] = q.pay(3, k);
      }
    }
  }
}

// Background tile with K % 4 == 0: the 16x16 tile's rows of the four outputs are filled with -1 in memory order by all
// 256 threads -- every store instruction of a wave covers one contiguous 1 KiB piece (a lane writing its own pixel's
// K-row stores 16 bytes at a stride of 4*K..12*K).  Per tile row the outputs hold cols*K floats (zbuf, dists),
// 3*cols*K floats (bary) and cols*K int64 (pix_to_face) = 7 * cols*K/4 16-byte pieces.
template <int THREADS>
__device__ __forceinline__ void fill_tile_background(const MeshArgs& a, int n, int ty0, int tx0, int y_end, int x_end,
                                                     int tid, int side = kTile) {
  const int H = a.H, W = a.W, K = a.K;
  const int rows = min(side, y_end - ty0), cols = min(side, x_end - tx0);
  const int q4 = (cols * K) >> 2;       // 16-byte pieces of one tile row of zbuf
  const int per_row = 7 * q4;           // zbuf q4 | dists q4 | bary 3 q4 | p2f 2 q4
  const int64_t col0 = W - tx0 - cols;  // outputs are stored flipped: x_out = W-1-x
  for (int r = 0; r < rows; ++r) {
    const int64_t px = ((int64_t)n * H + (H - 1 - (ty0 + r))) * W + col0;
    for (int e = tid; e < per_row; e += THREADS) {
      // branch-free choice of (output, piece): lanes of one wave straddle the boundaries between the outputs
      const bool in_z = e < q4, in_d = e < 2 * q4, in_b = e < 5 * q4;
      char* base = in_z ? reinterpret_cast<char*>(a.zbuf + px * K)
                        : (in_d ? reinterpret_cast<char*>(a.dists + px * K)
                                : (in_b ? reinterpret_cast<char*>(a.bary + px * K * 3) : reinterpret_cast<char*>(a.p2f + px * K)));
      const int piece = e - (in_z ? 0 : (in_d ? q4 : (in_b ? 2 * q4 : 5 * q4)));
      const unsigned v = in_b ? 0xbf800000u : ~0u;  // four -1.0f, or two int64 -1 (a select of two uint4 constants compiles to a scratch array)
      store16(base + (size_t)piece * 16, v, v, v, v);
    }
  }
}

// A full 16x16 background tile with K = KT, KT % 4 == 0: thread (row = tid / 16, c = tid % 16) owns the 16-byte pieces
// c, c + 16, ... of its row of each output -- four base pointers, then KT/4 + KT/4 + 3 KT/4 + KT/2 stores at constant
// offsets (K = 8: 14 stores and ~25 VALU per thread; the generic loop above spends ~15 VALU per store on choosing the
// output).  A wave's store covers four rows x 256 contiguous bytes.  This is what an ACTIVE workgroup runs for the
// background tiles it has been dealt (piggyback fill, see the kernel).
template <int KT>
__device__ __forceinline__ void fill_tile_full(const MeshArgs& a, int n, int ty0, int tx0, int tid) {
  static_assert(KT % 4 == 0, "16-byte pieces");
  const int H = a.H, W = a.W;
  const int r = tid >> 4, c = tid & 15;
  const int64_t px = ((int64_t)n * H + (H - 1 - (ty0 + r))) * W + (W - tx0 - kTile);  // outputs are stored flipped
  constexpr int Q = KT / 4;  // pieces per 16 pixels of zbuf, in units of 16 lanes
  char* zb = reinterpret_cast<char*>(a.zbuf + px * KT) + c * 16;
  char* db = reinterpret_cast<char*>(a.dists + px * KT) + c * 16;
  char* bb = reinterpret_cast<char*>(a.bary + px * KT * 3) + c * 16;
  char* ib = reinterpret_cast<char*>(a.p2f + px * KT) + c * 16;
  const unsigned m1 = 0xbf800000u, i1 = ~0u;
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    store16(zb + i * 256, m1, m1, m1, m1);
    store16(db + i * 256, m1, m1, m1, m1);
  }
#pragma unroll
  for (int i = 0; i < 3 * Q; ++i) store16(bb + i * 256, m1, m1, m1, m1);
#pragma unroll
  for (int i = 0; i < 2 * Q; ++i) store16(ib + i * 256, i1, i1, i1, i1);
}

// ---------------------------------------------------------------------------------------------------------------------
// Queues without payload (8 < K <= 48, round 4).  With SoftRas blur a covered pixel holds ~17 entries at K = 32..100, and a
// queue in private memory (TopKMem, what K > 16 ran on until round 3) moves O(K) entries of 24 bytes through scratch for
// every admitted face: K = 32 1.57 ms where K = 16 takes 0.44 (8 bench meshes, profiles/r04).  Here an entry is the ONE
// register pair (z | index) of topk.h: TopKPairs<KT, ., 0> -- the insertion is one 64-bit compare and two v_pk_mov per
// entry, as in the point rasterizer (raster_points.hip) -- and the signed distance and the barycentrics of the entries that
// survive are recomputed when the pixel is written: the same inlined functions (p3d_geom.h) on the same operands give the
// same bits.  The reference keeps everything in its 150-entry local array (rasterize_meshes.cu:216-237).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ FaceHit recompute_face(const MeshArgs& a, int f, f2 p, bool persp, bool clip) {
  const float* g = a.face_verts + (int64_t)f * 9;
  FaceRec fr;
  face_rec_make(mk3(g[0], g[1], g[2]), mk3(g[3], g[4], g[5]), mk3(g[6], g[7], g[8]), &fr);
  FaceHit h;
  const f3 bp = face_depth_rec(fr, p, persp, clip, &h);
  face_dist_rec(fr, p, a.blur, bp, &h);
  return h;
}

// One pixel's rows from a queue without payload.  Stage 1: pix_to_face and zbuf straight from the registers (16-byte
// stores when K % 4 == 0).  Stage 2: groups of four entries -- the lane reads its own four indices back (they were
// acknowledged by the L2: s_waitcnt vmcnt(0); the queue registers are dead by then, and a runtime loop cannot index
// registers), gathers the four faces' vertices in one round trip, recomputes (dist, bary) and writes the group with
// 16-byte stores, -1 padding included; a group no lane has an entry in is four -1 stores.
template <typename Queue, int KT, bool EXACT>
__device__ __forceinline__ void write_pixel_long(const MeshArgs& a, const Queue& q, int64_t opix, f2 p, bool pix_ok, bool persp,
                                                 bool clip) {
  const int K = EXACT ? KT : a.K;
  const int64_t base = opix * K;
  int nvalid = 0;
  if (pix_ok) {
#pragma unroll
    for (int k = 0; k < KT; ++k) nvalid += (k < K && q.valid(k)) ? 1 : 0;
    if ((K & 3) == 0) {
#pragma unroll
      for (int k = 0; k < KT; k += 4) {
        if (k < K) {
          long long iv[4];
          unsigned zv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ok = q.valid((k + j) % KT);
            iv[j] = ok ? (long long)q.ix((k + j) % KT) : -1ll;
            zv[j] = __float_as_uint(ok ? q.zf((k + j) % KT) : -1.0f);
          }
          store16(a.zbuf + base + k, zv[0], zv[1], zv[2], zv[3]);
          store16(a.p2f + base + k, (unsigned)iv[0], (unsigned)(iv[0] >> 32), (unsigned)iv[1], (unsigned)(iv[1] >> 32));
          store16(a.p2f + base + k + 2, (unsigned)iv[2], (unsigned)(iv[2] >> 32), (unsigned)iv[3], (unsigned)(iv[3] >> 32));
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k < K) {
          const bool ok = q.valid(k);
          a.p2f[base + k] = ok ? (int64_t)q.ix(k) : -1;
          a.zbuf[base + k] = ok ? q.zf(k) : -1.0f;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const bool vec = (K & 3) == 0;
#pragma unroll 1
  for (int k0 = 0; k0 < K; k0 += 4) {
    if (__ballot(pix_ok && k0 < nvalid) == 0) {  // uniform: nobody holds an entry here (the rows' -1 tail)
      if (pix_ok) {
        if (vec) {
          const unsigned m1 = 0xbf800000u;
          store16(a.dists + base + k0, m1, m1, m1, m1);
#pragma unroll
          for (int c = 0; c < 3; ++c) store16(a.bary + (base + k0) * 3 + 4 * c, m1, m1, m1, m1);
        } else {
          for (int j = 0; j < 4 && k0 + j < K; ++j) {
            a.dists[base + k0 + j] = -1.0f;
            for (int c = 0; c < 3; ++c) a.bary[(base + k0 + j) * 3 + c] = -1.0f;
          }
        }
      }
      continue;
    }
    int f[4];
    float gv[4][9];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[j] = -1;
      if (pix_ok && k0 + j < nvalid)
        f[j] = (int)__hip_atomic_load(a.p2f + base + k0 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // past the L1: this lane's own store
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* g = a.face_verts + (int64_t)(f[j] < 0 ? 0 : f[j]) * 9;
#pragma unroll
      for (int c = 0; c < 9; ++c) gv[j][c] = f[j] >= 0 ? g[c] : 0.0f;
    }
    float out[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      FaceHit h;
      h.dist = -1.0f;
      h.bary = mk3(-1.0f, -1.0f, -1.0f);
      if (f[j] >= 0) {
        FaceRec fr;
        face_rec_make(mk3(gv[j][0], gv[j][1], gv[j][2]), mk3(gv[j][3], gv[j][4], gv[j][5]), mk3(gv[j][6], gv[j][7], gv[j][8]), &fr);
        const f3 bp = face_depth_rec(fr, p, persp, clip, &h);
        face_dist_rec(fr, p, a.blur, bp, &h);
      }
      out[j] = h.dist;
      out[4 + 3 * j + 0] = h.bary.x;
      out[4 + 3 * j + 1] = h.bary.y;
      out[4 + 3 * j + 2] = h.bary.z;
    }
    if (pix_ok) {
      if (vec) {
        store16(a.dists + base + k0, __float_as_uint(out[0]), __float_as_uint(out[1]), __float_as_uint(out[2]), __float_as_uint(out[3]));
#pragma unroll
        for (int c = 0; c < 3; ++c)
          store16(a.bary + (base + k0) * 3 + 4 * c, __float_as_uint(out[4 + 4 * c]), __float_as_uint(out[5 + 4 * c]),
                  __float_as_uint(out[6 + 4 * c]), __float_as_uint(out[7 + 4 * c]));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (k0 + j < K) {
            a.dists[base + k0 + j] = out[j];
#pragma unroll
            for (int c = 0; c < 3; ++c) a.bary[(base + k0 + j) * 3 + c] = out[4 + 3 * j + c];
          }
        }
      }
    }
  }
}

// Staged face in LDS: five 16-byte words, each read by a wave as one broadcast (all lanes, same address).
//   [0] v0x v0y v1x v1y   [1] v2x v2y z0 z1   [2] z2 fid nb wide   [3] rd_area, rd_l01 (doubles)   [4] rd_l02, rd_l12
//   [5] d01x d01y d12x d12y   [6] d20x d20y - -      (the edge vectors, FaceRec::d01 ...)
constexpr int kRecWords = 7;

// One 64-face group of a staged chunk against one wave's 8x8 sub-tile: every candidate is evaluated on its FaceRec
// (shared-reciprocal arithmetic, p3d_geom.h: same bits as the reference's expression tree).  GENERAL adds the
// clipped-neighbour rule (rasterize_meshes.cu:186-215).  Two instantiations, chosen per tile region (see the kernel):
// with the rule in the common loop body the compiler copies the whole queue (48 moves) after every hit to feed the
// rule's control flow.
// Which pixels of the wave's 8x8 sub-tile lie inside the face's blur-expanded bounding box comes as a 64-bit LANE MASK
// (lane = 8 * row + column), built once per (face, sub-tile) by the face's lane in wave_chunk from the column / row masks of
// stage_chunk: a visit reads it with two v_readlane and uses it as the execution mask as it is -- no box in LDS, no four
// compares per visit, and a candidate that no pixel can see costs no LDS round trip at all.
// TIES (CUDA tie order, see mesh_cuda_order_kernel).  `tie_z` = the depth at which an entry was last dropped IN A TIE with the
// queue's K-th entry -- a candidate at exactly that depth that is not admitted, or an insertion into a full queue that leaves
// the K-th depth where it was (the entry pushed off the end tied with its successor) -- and `tie_drop` = the smallest face
// index among the entries dropped at that depth.  Everything the culls discard lies strictly behind the K-th entry, and the
// K-th depth only ever decreases, so at the end the entries dropped at the depth zK of the last survivor are known (tie_z ==
// zK) or there are none.  Which pixels then differ from the reference's CUDA procedure (an unsorted array filled in ascending
// face index; a newcomer replaces the first-placed largest entry iff it is STRICTLY nearer): with S = the hits nearer than zK
// (all kept by both), T = the hits at zK, m = K - |S| places for them, the array is "full without an entry beyond zK" from
// the moment K hits with z <= zK have arrived; members of T arriving before that moment enter, later ones are turned away, and
// every member of S arriving after it evicts the first-placed member of T.  If all of S precedes t* = the first member of T
// that does not fit (the smallest index the total order drops), exactly the m first members of T enter and stay: the total
// order's choice.  Otherwise the pixel is marked: max index over S > tie_drop (mesh_raster_kernel's epilogue).  Under the
// neighbour rule (GENERAL) every evaluated pixel is replayed (tie_z = +inf).
template <bool GENERAL, typename Queue, bool PC = false, bool TIES = false>
__device__ __forceinline__ void eval_candidates(const MeshArgs& a, int K, Queue& q, unsigned long long cand, int oj, unsigned mlo,
                                                unsigned mhi, float zcv, f2 p, bool pix_ok, bool persp, bool clip,
                                                const float4 (*s_rec)[kRecWords], float& tie_z, int& tie_drop) {
  while (cand) {
    const int ci = __builtin_ctzll(cand);
    cand &= cand - 1;
    int jj = __builtin_amdgcn_readlane(oj, ci);
    const unsigned alo = (unsigned)__builtin_amdgcn_readlane((int)mlo, ci), ahi = (unsigned)__builtin_amdgcn_readlane((int)mhi, ci);
    unsigned long long inside = ((unsigned long long)ahi << 32) | (unsigned long long)alo;
    float zc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zcv), ci));
    if constexpr (!GENERAL) {
      // Disjoint pairs.  A face's box covers 40 of a sub-tile's 64 pixels on average, so an evaluation runs with a third
      // of its lanes idle.  The next candidate whose pixel mask does not intersect this one's is evaluated in the SAME
      // pass: its lanes read its record (the LDS reads become two-address gathers), everything after that is per lane
      // anyway.  Any order of the candidates gives the same queues here (top-K under a total order; the neighbour rule,
      // which is order-dependent, lives in the GENERAL nest), so taking a later candidate early changes nothing but the
      // number of passes: 20 % fewer on the bench batch by bounding boxes alone (profiles/next/README.md).
      // (Measured and dropped in round 4: pairing against the lanes that really EVALUATE the candidate -- inside its box and not
      // behind the lane's K-th entry -- instead of its whole box: more pairs, +2 % time, profiles/r04/exp_active_pairs.txt.)
      const unsigned long long free_of_a = __ballot(((mlo & alo) | (mhi & ahi)) == 0u) & cand;  // lane = candidate face
      if (free_of_a) {  // uniform
        const int cb = __builtin_ctzll(free_of_a);
        cand &= ~(1ull << cb);
        const unsigned long long in_b = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mhi, cb) << 32) |
                                        (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mlo, cb);
        const bool mine_is_b = __builtin_amdgcn_inverse_ballot_w64(in_b);
        const int jb = __builtin_amdgcn_readlane(oj, cb);
        const float zb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zcv), cb));
        jj = mine_is_b ? jb : jj;
        zc = mine_is_b ? zb : zc;
        inside |= in_b;
      }
    }
    const bool in_box = __builtin_amdgcn_inverse_ballot_w64(inside);  // the mask IS the predicate: no VALU
    const bool too_deep = zc > q.kth_z(K);
    if (pix_ok && in_box && !too_deep) {
      const float kz0 = TIES ? q.kth_z(K) : 0.0f;
      const int ki0 = TIES ? q.kth_i(K) : 0;
      if constexpr (TIES && GENERAL) tie_z = INFINITY;
      // all LDS reads of this candidate are issued together (one round trip instead of dependent ones)
      // (jj differs per lane when two candidates share the pass: the record address is a per-lane product -- a 24-bit multiply,
      // v_mul_u32_u24, instead of the quarter-rate v_mul_lo_u32 the plain index compiled to)
      const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_rec) + __umul24((unsigned)jj, (unsigned)(kRecWords * sizeof(float4))));
      const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
      const int f = __float_as_int(r2.y);
      FaceHit h;
      bool hit = false;
      {
        const double2 d0 = *reinterpret_cast<const double2*>(&rec[3]);
        const double2 d1 = *reinterpret_cast<const double2*>(&rec[4]);
        FaceRec fr;
        fr.v0 = mk3(r0.x, r0.y, r1.z);
        fr.v1 = mk3(r0.z, r0.w, r1.w);
        fr.v2 = mk3(r1.x, r1.y, r2.x);
        fr.rd_area = d0.x;
        fr.rd_l01 = d0.y;
        fr.rd_l02 = d1.x;
        fr.rd_l12 = d1.y;
        const float4 r5 = rec[5];
        const float2 r6 = *reinterpret_cast<const float2*>(&rec[6]);
        fr.d01 = mk2(r5.x, r5.y);
        fr.d12 = mk2(r5.z, r5.w);
        fr.d20 = mk2(r6.x, r6.y);
        const bool pcs = PC && !GENERAL;  // wide faces make their chunk general (stage_chunk)
        fr.wide = pcs ? false : __float_as_int(r2.w) != 0;
        const f3 bp = face_depth_rec(fr, p, pcs || persp, pcs || clip, &h);
        // Depth first: a sample behind the camera or one that sorts after the K-th entry of a full queue is never
        // stored, whatever its distance -- half of the evaluations at the bench workload end here for every lane
        // (profiles/r03/probe_counts.txt).  Not under the neighbour rule: a face may replace its queued other half.
        const bool adm = !(h.z < 0.0f) & q.admits(K, h.z, f);  // (one exec region, not two: `&&` put a branch between the tests)
        if constexpr (TIES && !GENERAL) {
          // turned away at the depth of the K-th entry (whether it lies within the blur radius is not looked at); selects, not
          // branches: control flow between the stages of the evaluation costs the kernel 60 registers
          const bool ev = (h.z == kz0) & !adm;
          const int d = kz0 == tie_z ? min(tie_drop, f) : f;
          tie_drop = ev ? d : tie_drop;
          tie_z = ev ? kz0 : tie_z;
        }
        if (GENERAL || adm) hit = face_dist_rec<PC && !GENERAL>(fr, p, a.blur, bp, &h);  // (faces with a degenerate edge make their chunk general)
      }
      if (hit) {
        bool ins = true;
        if constexpr (GENERAL) {
          const int nb = __float_as_int(r2.z);
          if (nb != -1) {
            // clipped-face neighbour rule (rasterize_meshes.cu:186-215, rasterize_meshes_cpu.cpp:249-277): at most one of
            // the two halves of a split face stays in the queue -- the one closer to the pixel.
            const int at = q.find(nb);
            if (at >= 0) {
              float queued;
              if constexpr (Queue::kPayload > 0) {
                queued = q.payload_at(0, at);
              } else {
                queued = recompute_face(a, nb, p, persp, clip).dist;  // a queue without payload: the same function, the same bits
              }
              if (fabsf(h.dist) < fabsf(queued))
                q.erase(at);
              else
                ins = false;
            }
          }
        }
        // a candidate that sorts after the K-th entry of a full queue would fall straight off the end of the insertion
        // network: skip the network for it
        if (ins && q.admits(K, h.z, f)) {
          if constexpr (Queue::kPayload > 0) {
            const float pl[kMeshPayload] = {h.dist, h.bary.x, h.bary.y, h.bary.z};
            q.insert(K, h.z, f, pl);
          } else {
            const float none[1] = {0.0f};  // (z, index) only: distance and barycentrics are recomputed when the pixel is written
            q.insert(K, h.z, f, none);
          }
          if constexpr (TIES && !GENERAL) {
            // a full queue pushed its K-th entry off, and the new K-th entry ties with it
            const bool ev = (kz0 < INFINITY) & (q.kth_z(K) == kz0);
            const int d = kz0 == tie_z ? min(tie_drop, ki0) : ki0;
            tie_drop = ev ? d : tie_drop;
            tie_z = ev ? kz0 : tie_z;
          }
        }
      }
    }
  }
}

// Row cover (MeshArgs::cover): one wave's 8x8 sub-tile, lane -> input pixel (sy0 + lane / 8, sx0 + lane % 8), `any` = the
// pixel holds at least one face.  Output pixel (H - 1 - y, W - 1 - x) belongs to block (yo / 16, xo / 16), bit yo % 16.
// A sub-tile touches at most 2 x 2 blocks (one when H and W are multiples of 8); everything but the ballots is scalar.
__device__ __forceinline__ void cover_mark(const MeshArgs& a, int n, int sy0, bool any, int yo, int xo, int lane) {
  unsigned long long rem = __ballot(any);
  const int wid = (yo >> 4) * a.CX + (xo >> 4);
  while (rem) {
    const int w0 = __builtin_amdgcn_readlane(wid, __builtin_ctzll(rem));
    const unsigned long long same = __ballot(any && wid == w0);
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if ((same >> (8 * j)) & 0xffull) bits |= 1u << ((a.H - 1 - (sy0 + j)) & 15);
    rem &= ~same;
    if (lane == 0) {
      const int word = (n * a.CY) * a.CX + w0;
      if (a.area_list != nullptr && !a.list_by_plan) {
        // the first to mark this word lists it for the backward (which then needs no pass over the cover to find its work)
        const int before = atomicOr(a.cover + word, (int)bits);
        if (before == 0 && bits != 0u) a.area_list[atomicAdd(a.area_count, 1)] = word;
      } else {
        atomicOr(a.cover + word, (int)bits);
      }
    }
  }
}

__device__ __forceinline__ float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

struct StageLds {
  unsigned* pm;     // per staged face: column mask (bits 0..15) | row mask (bits 16..31) of the tile's pixels inside its box
  float pxy;        // lane c < 16: NDC centre of the tile's pixel column c; lane 16 + r: of its pixel row r (every wave alike)
  unsigned valid_c, valid_r;  // columns / rows of the tile that exist in the image
  float4 (*rec)[kRecWords];
  float* zc;
  int* order;
  float* qlow;
  ChunkOrderScratch* ord;
  int* wcnt;
};

struct TileRect {
  float x0, x1, y0, y1;  // pixel-centre extent of the workgroup's tile (NDC)
};

// Stage up to 256 faces of the tile's list into LDS: per-face setup (done once per workgroup -- the reference redoes it
// per pixel), tile cull, ordered compaction (ballot + mbcnt), FaceRec.  Returns the number of staged faces;
// *general = some staged face has a clipped neighbour (workgroup-uniform).  Ends with a barrier.
template <bool BINNED, bool PC = false>
__device__ __forceinline__ int stage_chunk(const MeshArgs& a, const StageLds& l, const TileRect& tile, int64_t src_base,
                                           int count, int base, int tid, bool cull, bool clip, bool prune, bool* general) {
  const int lane = tid & 63, w = tid >> 6;
  const int i = base + tid;
  bool keep = false;
  f3 v0, v1, v2;
  FaceSetup fs;
  int fid = -1, nb = -1;
  unsigned cm = 0, rm = 0;
  const bool has = i < count;  // this lane stages a face of the chunk
  fs.xlo = fs.xhi = fs.ylo = fs.yhi = 0.0f;
  fs.reject = true;
  if (has) {
    fid = BINNED ? a.csr.list[src_base + i] : (int)(src_base + i);
    const float* g = a.face_verts + (int64_t)fid * 9;
    nb = (int)a.neighbor[fid];  // requested together with the vertices: one memory round trip, not two
    v0 = mk3(g[0], g[1], g[2]);
    v1 = mk3(g[3], g[4], g[5]);
    v2 = mk3(g[6], g[7], g[8]);
    fs = face_setup(v0, v1, v2, a.sqrt_blur, cull);
  }
  // the tile's pixel columns / rows whose centres lie inside the blur-expanded box: the per-pixel bbox test of the
  // reference (rasterize_meshes.cu:94-97, strict comparisons) done once per (face, column) and (face, row)
  // (the centres are monotone in the pixel index, so the pixels inside form the range [#(centre < lo), 15 - #(centre > hi)]).
  // The 32 centres are read from the lanes of `pxy` BETWEEN the two halves of the staging branch, under a scalar branch ("some lane of
  // this wave stages a face"): every lane of the wave is active here.  Inside `if (has)` the lanes past the end of the list are not,
  // and a lane table may only be read where its source lanes cannot have been restored under a narrower exec mask
  // (profiles/r06/spill_root_cause.md: what cost round 4's spilling kernels their last faces).  Lanes without a face compare zeros.
  int xb = 0, xa = 0, yb = 0, ya = 0;  // centres below the box's low edge / above its high edge
  if (__ballot(has) != 0) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float xs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l.pxy), c));
      const float ys = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(l.pxy), 16 + c));
      xb += xs < fs.xlo ? 1 : 0;
      xa += xs > fs.xhi ? 1 : 0;
      yb += ys < fs.ylo ? 1 : 0;
      ya += ys > fs.yhi ? 1 : 0;
    }
  }
  if (has) {
    cm = range_mask16(xb, xa) & l.valid_c;
    rm = range_mask16(yb, ya) & l.valid_r;
    keep = !fs.reject && cm != 0 && rm != 0;  // some pixel centre of the tile is inside the box
    if (keep && prune)
      keep = !rect_cannot_hit(mk2(v0.x, v0.y), mk2(v1.x, v1.y), mk2(v2.x, v2.y), tile.x0, tile.x1, tile.y0, tile.y1, a.sqrt_blur);
  }
  const unsigned long long km = __ballot(keep);
  if (lane == 0) l.wcnt[w] = __popcll(km);
  __syncthreads();
  int pos = mask_rank(km);
  int staged = 0;
#pragma unroll
  for (int j = 0; j < kStage / kWave; ++j) {
    const int c = l.wcnt[j];
    if (j < w) pos += c;
    staged += c;
  }
  bool gen = false;
  if (keep) {
    FaceRec fr;
    face_rec_make(v0, v1, v2, &fr);
    // PC: the fast nest evaluates ordinary faces only -- per-pixel reciprocals from the f32 seed (`wide` false) and segment
    // distances without the degenerate-edge alternative (an edge shorter than 1e-4 NDC on a face that is not of zero area)
    gen = nb != -1 || (PC && (fr.wide || face_rec_degenerate(fr)));
    l.pm[pos] = cm | (rm << 16);
    l.rec[pos][0] = make_float4(v0.x, v0.y, v1.x, v1.y);
    l.rec[pos][1] = make_float4(v2.x, v2.y, v0.z, v1.z);
    l.rec[pos][2] = make_float4(v2.z, __int_as_float(fid), __int_as_float(nb), __int_as_float(fr.wide ? 1 : 0));
    double2 d0, d1;
    d0.x = fr.rd_area;
    d0.y = fr.rd_l01;
    d1.x = fr.rd_l02;
    d1.y = fr.rd_l12;
    *reinterpret_cast<double2*>(&l.rec[pos][3]) = d0;
    *reinterpret_cast<double2*>(&l.rec[pos][4]) = d1;
    l.rec[pos][5] = make_float4(fr.d01.x, fr.d01.y, fr.d12.x, fr.d12.y);
    l.rec[pos][6] = make_float4(fr.d20.x, fr.d20.y, 0.0f, 0.0f);
    // Depth cull (exact): with clipped barycentrics a sample's depth is a convex combination of
    // the vertex depths, so pz >= zmin * (1 - 4e-7) in float arithmetic (three roundings each in
    // the normalisation and the dot product); a lane whose queue is full with K-th depth below
    // that bound can never admit the face.  Not applicable when barycentrics are unclipped (pz may
    // leave [zmin, zmax]), when the face has a clipped neighbour (it may REPLACE a queued entry,
    // rasterize_meshes.cu:186-215), or for depths so small that bary_clip's 1e-5 floor could bite.
    const float zmin = min3(v0.z, v1.z, v2.z);
    l.zc[pos] = (clip && nb == -1 && zmin >= 1e-3f) ? zmin * 0.999998f : -INFINITY;
  }
  // also the barrier that publishes the staged records
  *general = __syncthreads_or(gen ? 1 : 0) != 0;
  return staged;
}

struct SubTile {
  float x0, x1, y0, y1;  // pixel-centre extent of the wave's 8x8 sub-tile (NDC)
};

// One wave's pass over a staged chunk: sub-tile cull 64 faces at a time (one lane per face), then the per-pixel loop.
// DEAL >= 0 (split mode, see the kernel): the four waves of the workgroup share ONE sub-tile and this wave (number DEAL)
// takes positions 16*(4q + DEAL) .. +15 of the visiting order -- at most one group of 64 per chunk, dealt in blocks of
// 16 so that every wave sees a front-to-back subsequence of about the same depth range.
template <bool GENERAL, typename Queue, bool PC = false, bool TIES = false>
__device__ __forceinline__ void wave_chunk(const MeshArgs& a, int K, Queue& q, int staged, bool sorted, const SubTile& st, f2 p,
                                           bool pix_ok, int lane, bool persp, bool clip, bool prune, const unsigned* s_pm,
                                           const float4 (*s_rec)[kRecWords], const float* s_zc, const int* s_order,
                                           const float* s_qlow, int cshift, int rshift, float& tie_z, int& tie_drop, int deal = -1) {
  for (int jb = 0; jb < staged; jb += kWave) {
    const int jfirst = deal >= 0 ? deal * 16 : jb;  // this wave's first position
    if (jfirst >= staged) break;
    // sorted chunk: once the nearest remaining face is too deep for every pixel of this wave, so is everything behind it
    if (sorted && __ballot(pix_ok && !(s_qlow[jfirst] > q.kth_z(K))) == 0) break;
    const int j = deal >= 0 ? (((lane >> 4) * 4 + deal) * 16 + (lane & 15)) : jb + lane;
    bool touch = false;
    int oj = 0;
    unsigned mlo = 0, mhi = 0;
    float zcv = 0.0f;
    if (j < staged) {
      oj = s_order[j];
      const unsigned pm = s_pm[oj];
      // this wave's 8 columns / 8 rows of the tile's masks; the sub-tile's pixels inside the box as a 64-bit lane mask
      // (lane = 8 * row + column; p3d_geom.h: block_mask_8x8)
      const unsigned cmask = (pm >> cshift) & 0xffu, rmask = (pm >> (16 + rshift)) & 0xffu;
      touch = (cmask != 0) & (rmask != 0);
      if (touch && prune) {
        const float4 r0 = s_rec[oj][0], r1 = s_rec[oj][1];
        touch = !rect_cannot_hit(mk2(r0.x, r0.y), mk2(r0.z, r0.w), mk2(r1.x, r1.y), st.x0, st.x1, st.y0, st.y1, a.sqrt_blur);
      }
      block_mask_8x8(cmask, rmask, &mlo, &mhi);
      zcv = s_zc[oj];
    }
    const unsigned long long cand = __ballot(touch);
    eval_candidates<GENERAL, Queue, PC, TIES>(a, K, q, cand, oj, mlo, mhi, zcv, p, pix_ok, persp, clip, s_rec, tie_z, tie_drop);
    if (deal >= 0) break;  // a chunk holds at most 256 positions: one group per wave
  }
}

// Split mode: queue entries of waves 1..3 pass through LDS to wave 0.  Layout [word][entry k][lane] (conflict-free).
constexpr int kMergeWords = 2 + kMeshPayload;  // z, idx, payload

template <typename Queue, int KT>
__device__ __forceinline__ void merge_dump(const Queue& q, float* slab, int lane) {
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    slab[(0 * KT + k) * kWave + lane] = q.zf(k);
    slab[(1 * KT + k) * kWave + lane] = __int_as_float(q.ix(k));
#pragma unroll
    for (int pp = 0; pp < kMeshPayload; ++pp) slab[((2 + pp) * KT + k) * kWave + lane] = q.pay(pp, k);
  }
}

template <typename Queue, int KT>
__device__ __forceinline__ void merge_absorb(Queue& q, int K, const float* slab, int lane) {
#pragma unroll 1
  for (int k = 0; k < KT; ++k) {
    const int idx = __float_as_int(slab[(1 * KT + k) * kWave + lane]);
    if (__ballot(idx != kEmptyIdx) == 0) break;  // entries are sorted: empty ones are last in every lane
    const float z = slab[(0 * KT + k) * kWave + lane];
    float pl[kMeshPayload];
#pragma unroll
    for (int pp = 0; pp < kMeshPayload; ++pp) pl[pp] = slab[((2 + pp) * KT + k) * kWave + lane];
    if (idx != kEmptyIdx && q.admits(K, z, idx)) q.insert(K, z, idx, pl);
  }
}

// EXACT: K == KT is known at compile time (K = 1, 2, 4, 8): the queue's live capacity folds to a constant and the
// generic epilogue (fill + patch, for K that has no vector-row path) is not even compiled into the hot kernels.
// WAVES: minimum waves per SIMD the register allocation leaves room for (512 / WAVES registers per lane).
// PC: perspective_correct && clip_barycentric_coords are known to be set (what MeshRasterizer uses for perspective
// cameras with blur): the per-(pixel, face) test of the common loop nest becomes one basic block -- the flag branches
// between its stages kept the compiler from overlapping the distance arithmetic with the latency of the double-precision
// reciprocal chains -- and faces with `wide` reciprocals are sent to the general nest instead of being tested per candidate.
// SPLIT (few tiles: a single image or a small batch): one workgroup per 8x8 SUB-tile, its four waves share the 64 pixels
// and each takes a quarter of every staged chunk (wave_chunk's `deal`); the four queues meet in wave 0 through LDS
// before the pixel is written (the K nearest of a union are among the K nearest of its parts).  A 16x16 tile per
// workgroup makes every wave walk the tile's whole list for its own sub-tile: with one image on the chip (BASELINE
// configs[1]: 256 tiles for 256 CUs) the launch lasts as long as the longest such walk (~0.2 ms on the cow), while three
// quarters of the CUs idle.
template <typename Queue, int KT, bool IN_REGS, bool BINNED, bool EXACT, int WAVES = kFineWaves, bool PC = false,
          bool SPLIT = false, bool TIES = false>
__global__ __launch_bounds__(kStage, WAVES) void mesh_raster_kernel(MeshArgs a) {
  static_assert(!(SPLIT && TIES), "the tie marks of four queues that merge are not tracked: CUDA tie order runs the plain kernels");
  __shared__ float s_merge[SPLIT ? 3 * kMergeWords * KT * kWave : 1];
  __shared__ unsigned s_pm[kStage];                 // pixel column | row masks of the staged faces (StageLds::pm)
  __shared__ float4 s_rec[kStage][kRecWords];       // see kRecWords
  __shared__ __align__(16) float s_zc[kStage];      // depth-cull key: every sample of the face has z >= s_zc (or -inf)
  __shared__ int s_order[kStage];                   // visiting order of the staged faces: ascending s_zc (by bucket) when order is free
  __shared__ float s_qlow[kStage];                  // lower bound of s_zc over sorted positions >= i
  __shared__ ChunkOrderScratch s_ord;
  __shared__ int s_wcnt[kStage / kWave];

  // Which tile.  With a tile plan (binning.h: the lists came from bin_build and a tile is a bin) the blocks walk the ACTIVE
  // tiles, longest list first, and then the background ones.  Workgroups reach the CUs round robin by index, not by load
  // (profiles/r03/bwd_timeline.txt: the backward ran with half of its wave slots empty until every workgroup carried
  // work), so the order of the work items is the load balance: in image order (3 of 5 tiles of the bench launch are
  // background, and the few tiles with many hundred faces -- 250-330 us each, profiles/r02_fine_timeline.txt -- start at
  // arbitrary times) some CUs idle while others queue; dealt longest-first every CU draws the same mix and the launch has no
  // tail.  Measured: 1.19 -> 1.04 ms.  Otherwise (naive path, caller's bins, split mode): the XCD-aware tile map.
  if (a.overflow != nullptr && (*a.overflow != 0) == BINNED) return;  // uniform (scalar load)
  TileCoord tc;
  unsigned blk = SPLIT ? blockIdx.x >> 2 : blockIdx.x;
  bool listed = false;
  if (BINNED && !SPLIT && a.walk_plan && a.csr.plan.hdr[3] != 0) {  // uniform
    const unsigned A = (unsigned)a.csr.plan.hdr[0], B = (unsigned)a.csr.plan.hdr[1];
    int row;
    if (blk < A) {
      row = a.csr.plan.order[blk];
      if (a.list_by_plan && threadIdx.x == 0) {
        // this tile's word of the cover into slot `blk` of the list: the backward then walks the areas longest list first, as this launch does
        const int per = a.tm.BH * a.tm.BW;
        const int ln = row / per, lrem = row - ln * per, lby = lrem / a.tm.BW, lbx = lrem - lby * a.tm.BW;
        a.area_list[blk] = (ln * a.CY + (a.CY - 1 - lby)) * a.CX + (a.CX - 1 - lbx);
        if (blk == 0) *a.area_count = (int)A;
      }
    } else if (blk < A + B) {
      if (EXACT && (KT & 3) == 0 && A > 0) return;  // piggyback fill (below): the active workgroups write this tile
      row = a.csr.plan.bg_list[blk - A];
    } else {
      return;
    }
    const int per_image = a.tm.BH * a.tm.BW;
    tc.n = row / per_image;
    const int rem = row - tc.n * per_image;
    tc.by = rem / a.tm.BW;
    tc.bx = rem - tc.by * a.tm.BW;
    tc.ty = tc.tx = 0;
    listed = true;
  }
  if (!listed && !tile_of_block(a.tm, blk, &tc)) return;
  const int n = tc.n, by = tc.by, bx = tc.bx, ty = tc.ty, tx = tc.tx;

  const int H = a.H, W = a.W;
  const int y_end = min(H, (by + 1) * a.tm.bin_size);
  const int x_end = min(W, (bx + 1) * a.tm.bin_size);
  const int ty0 = by * a.tm.bin_size + ty * kTile;
  const int tx0 = bx * a.tm.bin_size + tx * kTile;
  if (ty0 >= y_end || tx0 >= x_end) return;  // tile has no pixel (uniform)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: the sub-tile origin stays in SGPRs
  const int sub = SPLIT ? (int)(blockIdx.x & 3u) : w;  // which 8x8 sub-tile of the tile this wave works on
  const int sy0 = ty0 + (sub >> 1) * 8;
  const int sx0 = tx0 + (sub & 1) * 8;
  if (SPLIT && (sy0 >= y_end || sx0 >= x_end)) return;  // uniform
  const int yi = sy0 + (lane >> 3);
  const int xi = sx0 + (lane & 7);
  const bool pix_ok = yi < y_end && xi < x_end;
  const bool wave_ok = sy0 < y_end && sx0 < x_end;

  int64_t src_base;
  int count;
  if (BINNED) {
    // both words of the bin's CSR row are requested at once (offset has rows + 1 entries, so the load is always valid)
    const int64_t row = ((int64_t)n * a.tm.BH + by) * a.tm.BW + bx;
    count = a.csr.total[row];
    src_base = a.csr.offset[row];
  } else {
    src_base = a.mesh_first[n];
    count = (int)a.mesh_count[n];
  }
  // Piggyback fill.  Tiles without faces ("background", 3 of 5 at the bench workload) are pure -1 stores, but a workgroup
  // that only stores still occupies a full slot of this kernel (256 threads x 128 registers; the chip holds 1024) for
  // ~5.6 us -- 0.2 ms of a 1.5 ms launch were slots held by background tiles (profiles/r02_ablate_fwd.txt: tiles with
  // faces alone 1.3 ms).  With the tile plan of the offsets scan (binning.h: TilePlan) the ACTIVE workgroups issue those
  // stores instead, at the end of their own work (fire and forget: ~40 instructions per tile, fill_tile_full), and a
  // background workgroup returns at once.  Active row number r takes background rows [r q, (r + 1) q), q = ceil(B / A).
  // Only when a bin is ONE tile (walk_plan: Ty == Tx == 1, images up to 512 pixels a side): with several tiles per bin a
  // background bin's tile (ty, tx) would be dealt to workgroup (ty, tx) of an active bin, and the workgroups of an active
  // bin in the partial last bin row / column whose own tile has no pixel return before they get here -- those tiles of
  // the background bins dealt to them were never written (tests/test_gpu_meshes.py:
  // test_large_images_background_is_written).  Larger images fill their background tiles with their own workgroups.
  const bool piggy = BINNED && !SPLIT && EXACT && (KT & 3) == 0 && a.walk_plan != 0 && a.csr.plan.hdr != nullptr;
  int plan_a = 0, plan_b = 0;
  if (piggy) {
    plan_a = a.csr.plan.hdr[0];
    plan_b = a.csr.plan.hdr[1];
  }
  if (BINNED && !SPLIT && a.list_by_plan && !listed && tid == 0) {
    // (a plan without a tile order -- small launches: the slot is the tile's rank among the active tiles)
    const int64_t row = ((int64_t)n * a.tm.BH + by) * a.tm.BW + bx;
    if (count > 0) a.area_list[a.csr.plan.arank[row]] = (n * a.CY + (a.CY - 1 - by)) * a.CX + (a.CX - 1 - bx);
    if (blockIdx.x == 0) *a.area_count = a.csr.plan.hdr[0];
  }
  if (count <= 0) {
    if (piggy && plan_a > 0) return;  // an active workgroup writes this tile (uniform)
    const int Kbg = EXACT ? KT : a.K;  // compile-time where it can be: the row-fill branch is then the only one left
    // background tile: nothing but the -1 stores; skip the NDC set-up below
    if (SPLIT && (Kbg & 3) == 0) {
      fill_tile_background<kStage>(a, n, sy0, sx0, y_end, x_end, tid, 8);
    } else if (SPLIT && w != 0) {
      // the four waves cover the same pixels: wave 0 writes them
    } else if ((Kbg & 3) == 0) {
      fill_tile_background<kStage>(a, n, ty0, tx0, y_end, x_end, tid);
    } else if constexpr (EXACT && Queue::kPayload > 0) {
      Queue e;
      e.init();
      if (pix_ok) write_pixel<Queue, KT, IN_REGS>(a, e, ((int64_t)n * H + (H - 1 - yi)) * W + (W - 1 - xi));
    } else if (wave_ok) {
      Queue e;
      e.init();
      write_subtile_fill_patch<Queue, KT, IN_REGS>(a, e, false, n, sy0, sx0, y_end, x_end, lane, pix_ok, yi, xi);
    }
    return;
  }

  // Pixel centres: ONE pix_to_ndc per lane for the whole set-up (round 4).  Lane l < 16 evaluates pixel column ox + l of the
  // tile, lane 16 + l pixel row oy + l (ox, oy: the workgroup's 16 x 16 tile, or in split mode its one 8 x 8 sub-tile; lanes
  // 32..63 repeat); the lane's own pixel centre and the eight pixel-centre extents of the tile and the sub-tile are then
  // FETCHED from those lanes (two ds_bpermute, eight v_readlane) -- same function, same integer argument, same bits.  Before,
  // every wave evaluated pix_to_ndc eleven times: ~24 IEEE divisions (11 instructions each), ~400 of the ~5000 VALU
  // instructions of a wave that holds faces.
  const int ox = SPLIT ? sx0 : tx0, oy = SPLIT ? sy0 : ty0;
  const float pxy = (lane & 16) ? pix_to_ndc(oy + (lane & 15), H, W) : pix_to_ndc(ox + (lane & 15), W, H);
  auto col_centre = [&](int x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pxy), x - ox)); };        // uniform x
  auto row_centre = [&](int y) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pxy), 16 + y - oy)); };  // uniform y
  const f2 p = mk2(__int_as_float(__builtin_amdgcn_ds_bpermute((xi - ox) << 2, __float_as_int(pxy))),
                   __int_as_float(__builtin_amdgcn_ds_bpermute((16 + yi - oy) << 2, __float_as_int(pxy))));
  // pixel-centre extents (pix_to_ndc is monotone in the pixel index); wave-uniform: SGPRs.  (Split mode stages against its one
  // sub-tile: the tile extents are not used there.)
  const float tile_x0 = SPLIT ? 0.0f : col_centre(tx0), tile_x1 = SPLIT ? 0.0f : col_centre(min(tx0 + kTile, x_end) - 1);
  const float tile_y0 = SPLIT ? 0.0f : row_centre(ty0), tile_y1 = SPLIT ? 0.0f : row_centre(min(ty0 + kTile, y_end) - 1);
  const float sub_x0 = col_centre(sx0), sub_x1 = col_centre(min(sx0 + 8, x_end) - 1);
  const float sub_y0 = row_centre(sy0), sub_y1 = row_centre(min(sy0 + 8, y_end) - 1);

  Queue q;
  q.init();
  const int K = EXACT ? KT : a.K;
  const bool persp = PC || a.persp != 0, clip = PC || a.clip != 0, cull = a.cull != 0;
  const bool prune = true;  // the conservative rectangle-vs-face reject of the tile / sub-tile culls
  StageLds lds;
  lds.pm = s_pm;
  {
    // the tile the masks refer to: the workgroup's 16 x 16 tile, or (split mode) its one 8 x 8 sub-tile
    const int side = SPLIT ? 8 : kTile;
    const int cols = min(side, x_end - ox), rows = min(side, y_end - oy);
    lds.valid_c = (1u << cols) - 1u;
    lds.valid_r = (1u << rows) - 1u;
    lds.pxy = pxy;
  }
  lds.rec = s_rec;
  lds.zc = s_zc;
  lds.order = s_order;
  lds.qlow = s_qlow;
  lds.ord = &s_ord;
  lds.wcnt = s_wcnt;
  TileRect tile;  // what a face has to touch to be staged: the tile, or (split mode) the workgroup's one sub-tile
  tile.x0 = SPLIT ? sub_x0 : tile_x0;
  tile.x1 = SPLIT ? sub_x1 : tile_x1;
  tile.y0 = SPLIT ? sub_y0 : tile_y0;
  tile.y1 = SPLIT ? sub_y1 : tile_y1;
  SubTile st;
  st.x0 = sub_x0;
  st.x1 = sub_x1;
  st.y0 = sub_y0;
  st.y1 = sub_y1;
  const bool run_waves = wave_ok;

  // Two loop nests, entered one after the other and never re-entered: chunks are processed by the FAST nest (shared-
  // reciprocal evaluation, front-to-back order) until the first chunk that holds a face with a clipped neighbour;
  // from that chunk on the GENERAL nest (the same evaluation + the neighbour rule, ascending index order) finishes
  // the tile.  Per-chunk switching back and forth would be equally exact, but with both bodies inside one
  // loop the register allocator spills the queue (150 VGPRs of scratch, and every wave of the launch -- background
  // tiles included -- then pays for scratch set-up: the pure fill ran 0.94 -> 1.7 ms).
  int base = 0;
  bool general = false;
  float tie_z = -1.0f;  // TIES only (depths are >= 0)
  int tie_drop = kEmptyIdx;
  int staged = 0;
  for (; base < count; base += kStage) {
    staged = stage_chunk<BINNED, PC>(a, lds, tile, src_base, count, base, tid, cull, clip, prune, &general);
    if (general) break;  // uniform
    chunk_bucket_order(s_zc, staged, s_order, s_qlow, s_ord, tid);
    if (run_waves)
      wave_chunk<false, Queue, PC, TIES>(a, K, q, staged, true, st, p, pix_ok, lane, persp, clip, prune, s_pm, s_rec, s_zc, s_order,
                                         s_qlow, SPLIT ? 0 : (sub & 1) * 8, SPLIT ? 0 : (sub >> 1) * 8, tie_z, tie_drop, SPLIT ? w : -1);
    __syncthreads();
  }
  if constexpr (SPLIT && Queue::kPayload > 0) {
    // the queues of waves 1..3 join wave 0's: before the general nest (its neighbour rule must see every entry queued so
    // far, and only wave 0 walks it), or before the pixels are written
    if (w != 0) merge_dump<Queue, KT>(q, s_merge + (w - 1) * kMergeWords * KT * kWave, lane);
    __syncthreads();
    if (w == 0) {
#pragma unroll 1
      for (int src = 0; src < 3; ++src) merge_absorb<Queue, KT>(q, K, s_merge + src * kMergeWords * KT * kWave, lane);
    }
  }
  if (general) {
    for (;;) {
      // chunk `base` is staged; faces keep ascending index order (the neighbour rule depends on it)
      if (tid < staged) s_order[tid] = tid;
      __syncthreads();
      if (run_waves && (!SPLIT || w == 0))
        wave_chunk<true, Queue, false, TIES>(a, K, q, staged, false, st, p, pix_ok, lane, persp, clip, prune, s_pm, s_rec, s_zc, s_order,
                                             s_qlow, SPLIT ? 0 : (sub & 1) * 8, SPLIT ? 0 : (sub >> 1) * 8, tie_z, tie_drop);
      __syncthreads();
      base += kStage;
      if (base >= count) break;
      bool dummy = false;
      staged = stage_chunk<BINNED, PC>(a, lds, tile, src_base, count, base, tid, cull, clip, prune, &dummy);
    }
  }

  // TIES: is the pixel a candidate for the replay (eval_candidates)?  The queue's part of the answer is taken here, while it
  // is in registers (the writes below consume it); S's largest index is read back from the pixel's own rows after them -- by
  // the few lanes that need it (reading the queue here as well cost 60 registers: the scheduler ran it into the row stores).
  float tie_kz = 0.0f;
  bool tie_cand = false, tie_gen = false;
  if constexpr (TIES) {
    tie_kz = q.kth_z(K);
    tie_gen = tie_z == INFINITY;
    // (K = 1: a one-entry array keeps the first of several equally deep faces in ascending index -- the total order's choice;
    // only the neighbour rule can differ there)
    tie_cand = K > 1 && tie_z == tie_kz && tie_kz < INFINITY;
  }
  {
    if constexpr (Queue::kPayload == 0) {
      static_assert(!SPLIT, "queues without payload do not run in split mode");
      if (wave_ok) write_pixel_long<Queue, KT, EXACT>(a, q, ((int64_t)n * H + (H - 1 - yi)) * W + (W - 1 - xi), p, pix_ok, persp, clip);
      if (a.cover != nullptr && wave_ok) cover_mark(a, n, sy0, pix_ok && q.valid(0), H - 1 - yi, W - 1 - xi, lane);
    } else if constexpr (EXACT) {
      if (pix_ok && (!SPLIT || w == 0)) {
        // the pixel's coordinates are rebuilt from a fresh lane id: keeping yi / xi alive across the chunk loops costs the
        // two registers that separate this kernel from the 120-VGPR allocation step
        int l2;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
        const int yo = sy0 + (l2 >> 3), xo = sx0 + (l2 & 7);
        write_pixel<Queue, KT, IN_REGS>(a, q, ((int64_t)n * H + (H - 1 - yo)) * W + (W - 1 - xo));
      }
      if (a.cover != nullptr && (!SPLIT || w == 0)) {  // uniform
        int l2;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
        const int yo = sy0 + (l2 >> 3), xo = sx0 + (l2 & 7);
        cover_mark(a, n, sy0, yo < y_end && xo < x_end && q.valid(0), H - 1 - yo, W - 1 - xo, l2);
      }
    } else {
      if (wave_ok) write_subtile_fill_patch<Queue, KT, IN_REGS>(a, q, true, n, sy0, sx0, y_end, x_end, lane, pix_ok, yi, xi);
      if (a.cover != nullptr && wave_ok && (!SPLIT || w == 0)) cover_mark(a, n, sy0, pix_ok && q.valid(0), H - 1 - yi, W - 1 - xi, lane);
    }
  }
  if constexpr (TIES) {
    // the mark: -2 in the pixel's first pix_to_face entry, after its rows have been acknowledged (other lanes may have written
    // them: write_subtile_fill_patch).  mesh_cuda_order_kernel rewrites every row of a marked pixel.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int l2;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
    const int yo = sy0 + (l2 >> 3), xo = sx0 + (l2 & 7);
    const bool inside = yo < y_end && xo < x_end;
    const int64_t tbase = (((int64_t)n * H + (H - 1 - yo)) * W + (W - 1 - xo)) * K;
    bool tie = tie_gen && inside;
    if (tie_cand && inside) {
      int s_max = -1;  // S's largest face index: the survivors in front of the last survivor's depth
      for (int k = 0; k < K; ++k) {  // (past the L1: rows other lanes may have filled)
        const long long fi = __hip_atomic_load(a.p2f + tbase + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float fz = __int_as_float(__hip_atomic_load(reinterpret_cast<const int*>(a.zbuf + tbase + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (fi >= 0 && fz < tie_kz) s_max = max(s_max, (int)fi);
      }
      tie = s_max > tie_drop;
    }
    if (tie) a.p2f[tbase] = -2;
    const unsigned long long tmask = __ballot(tie);
    if (a.tie_words != nullptr && wave_ok && l2 == 0) a.tie_words[((int64_t)n * a.SY + (sy0 >> 3)) * a.SX + (sx0 >> 3)] = tmask;
  }
  if constexpr (BINNED && !SPLIT && EXACT && (KT & 3) == 0) {
    if (piggy && plan_b > 0) {
      const int64_t row = ((int64_t)n * a.tm.BH + by) * a.tm.BW + bx;
      const int q_bg = (plan_b + plan_a - 1) / plan_a;
      const long long j0 = (long long)a.csr.plan.arank[row] * q_bg;
      const long long j1 = j0 + q_bg < (long long)plan_b ? j0 + q_bg : (long long)plan_b;
      const int per_image = a.tm.BH * a.tm.BW;
      for (long long j = j0; j < j1; ++j) {  // uniform: scalar loads and arithmetic
        const int brow = a.csr.plan.bg_list[j];
        const int bn = brow / per_image, brem = brow - bn * per_image;
        const int bby = brem / a.tm.BW, bbx = brem - bby * a.tm.BW;
        const int by_end = min(H, (bby + 1) * a.tm.bin_size), bx_end = min(W, (bbx + 1) * a.tm.bin_size);
        const int bty0 = bby * a.tm.bin_size + ty * kTile, btx0 = bbx * a.tm.bin_size + tx * kTile;  // the same tile of that bin
        if (bty0 >= by_end || btx0 >= bx_end) continue;
        if (bty0 + kTile <= by_end && btx0 + kTile <= bx_end)
          fill_tile_full<KT>(a, bn, bty0, btx0, tid);
        else
          fill_tile_background<kStage>(a, bn, bty0, btx0, by_end, bx_end, tid);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// CUDA tie order (p3d_rasterize_meshes_cuda_order).  The kernels above keep the K nearest faces of a pixel under the total
// order (z, face index) -- what the reference's CPU and Python implementations return.  Its CUDA kernels return the same
// depths but, where faces tie EXACTLY in depth at the K-th place, possibly other faces: they keep an unsorted array, replace
// "the" largest entry only by a strictly nearer candidate, and which of several equally far entries is "the" largest depends
// on the array positions, i.e. on the whole history of the pixel (rasterize_meshes.cu:216-237; the final sort is by (z, index):
// rasterize_meshes.cu:30-32, so only the survivors differ, never the order).  2 in 10^4 entries of the bench launch.  For users
// who diff against CUDA renders this kernel REPLAYS the reference's procedure, faces in ascending index, for the pixels the
// TIES instantiations of the fine kernel marked (-2 in the pixel's first pix_to_face entry: an entry was dropped at the depth of
// the last survivor, or the neighbour rule was in play) and rewrites their rows.  K = 1 needs none of this: a full one-entry
// array keeps the first face in ascending index among equals, as the total order does.
// Round 5: a WAVE per marked pixel.  Lanes are faces: 64 faces of the tile's list are set up and tested against the pixel at
// once (p3d_geom.h: face_setup, face_hit -- the same functions as everywhere else, the same bits), the hits are then fed to the
// reference's array one by one in list order (the array lives in LDS; finding "the" largest entry is a wave reduction), and
// the survivors are ranked by (z, index), re-evaluated by one lane each and written.  Until round 5 every lane of a tile with
// a full pixel walked the tile's whole list on its own: ~10 x the fine kernel's time; now `profiles/r05/c24/`.
// ---------------------------------------------------------------------------------------------------------------------
// One marked pixel, the whole wave on it (see above).  qz / qi: the wave's K-entry array in LDS.
template <bool BINNED>
__device__ __forceinline__ void replay_pixel(const MeshArgs& a, int n, int yi, int xi, volatile float* qz, volatile int* qi, int lane) {
  const int H = a.H, W = a.W, K = a.K;
  int64_t src;
  int count;
  if (BINNED) {
    const int64_t row = ((int64_t)n * a.tm.BH + yi / a.tm.bin_size) * a.tm.BW + xi / a.tm.bin_size;
    src = a.csr.offset[row];
    count = a.csr.total[row];
  } else {
    src = a.mesh_first[n];
    count = (int)a.mesh_count[n];
  }
  const bool persp = a.persp != 0, clip = a.clip != 0, cull = a.cull != 0;
  auto verts = [&](int f, f3* v0, f3* v1, f3* v2) {
    const float* g = a.face_verts + (int64_t)f * 9;
    *v0 = mk3(g[0], g[1], g[2]);
    *v1 = mk3(g[3], g[4], g[5]);
    *v2 = mk3(g[6], g[7], g[8]);
  };
  struct Batch {  // 64 faces of the list, one per lane
    int f;
    int nb;
    f3 v0, v1, v2;
  };
  auto fetch = [&](int i0) {
    Batch b;
    b.f = -1;
    b.nb = -1;
    b.v0 = b.v1 = b.v2 = mk3(0.0f, 0.0f, 0.0f);
    const int i = i0 + lane;
    if (i < count) {
      b.f = BINNED ? a.csr.list[src + i] : (int)(src + i);
      verts(b.f, &b.v0, &b.v1, &b.v2);
      b.nb = (int)a.neighbor[b.f];  // (face ids fit 31 bits; -1 stays -1)
    }
    return b;
  };
  const f2 p = mk2(pix_to_ndc(xi, W, H), pix_to_ndc(yi, H, W));
  const int64_t opix = ((int64_t)n * H + (H - 1 - yi)) * W + (W - 1 - xi);
  // the reference's array (rasterize_meshes.cu:291-294), depths and indices only: distance and barycentrics of the K
  // survivors are recomputed at the end.  qn, qmax_z, qmax_i are wave-uniform.
  int qn = 0, qmax_i = -1;
  float qmax_z = -1000.0f;
  Batch nxt = fetch(0);
  for (int i0 = 0; i0 < count; i0 += kWave) {
    const Batch cur = nxt;
    if (i0 + kWave < count) nxt = fetch(i0 + kWave);  // in flight while this batch's hits are fed to the array
    FaceHit h;
    bool hit = false;
    if (cur.f >= 0) {
      const FaceSetup fs = face_setup(cur.v0, cur.v1, cur.v2, a.sqrt_blur, cull);
      if (!fs.reject && !outside_box(fs, p)) hit = face_hit(cur.v0, cur.v1, cur.v2, p, a.blur, persp, clip, &h);
    }
    unsigned long long hits = __ballot(hit);
    while (hits) {  // in list order = ascending face index
      const int c = __builtin_ctzll(hits);
      hits &= hits - 1;
      const int cf = __builtin_amdgcn_readlane(cur.f, c);
      const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.z), c));
      const int cnb = __builtin_amdgcn_readlane(cur.nb, c);
      int at = -1;
      if (cnb != -1) {  // uniform
        for (int j0 = 0; j0 < qn; j0 += kWave) {  // the first match in array order, as the reference's loop finds it
          const unsigned long long is = __ballot(j0 + lane < qn && qi[j0 + lane] == cnb);
          if (is) {
            at = j0 + __builtin_ctzll(is);
            break;
          }
        }
      }
      if (at != -1) {
        // the clipped-face rule (rasterize_meshes.cu:186-213): the nearer of the two halves keeps the slot -- in place
        f3 n0, n1, n2;
        verts(cnb, &n0, &n1, &n2);
        FaceHit hn;
        face_hit(n0, n1, n2, p, a.blur, persp, clip, &hn);
        const float cdist = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h.dist), c));
        if (fabsf(cdist) < fabsf(hn.dist)) {  // uniform
          if (lane == 0) {
            qz[at] = cz;
            qi[at] = cf;
          }
          if (cz > qmax_z) {
            qmax_z = cz;
            qmax_i = at;
          }
        }
      } else if (qn < K) {
        if (lane == 0) {
          qz[qn] = cz;
          qi[qn] = cf;
        }
        if (cz > qmax_z) {
          qmax_z = cz;
          qmax_i = qn;
        }
        ++qn;
      } else if (cz < qmax_z) {
        if (lane == 0) {
          qz[qmax_i] = cz;
          qi[qmax_i] = cf;
        }
        // "the" largest entry: the first one in array order that exceeds the newcomer, the newcomer's slot if none does
        // (rasterize_meshes.cu:228-236: a scan with a strict comparison that starts from the newcomer)
        qmax_z = cz;
        __builtin_amdgcn_wave_barrier();
        float m = -INFINITY;
        for (int j = lane; j < K; j += kWave) m = fmaxf(m, qz[j]);
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        if (m > cz) {  // uniform
          for (int j0 = 0; j0 < K; j0 += kWave) {
            const unsigned long long is = __ballot(j0 + lane < K && qz[j0 + lane] == m);
            if (is) {
              qmax_i = j0 + __builtin_ctzll(is);
              break;
            }
          }
          qmax_z = m;
        }
      }
      __builtin_amdgcn_wave_barrier();  // lane 0's LDS writes before the wave's next reads (same wave: program order)
    }
  }
  // survivors in ascending (z, index) (rasterize_meshes.cu:30-32): entry j goes to place #{entries that sort before it}
  for (int j = lane; j < K; j += kWave) {
    const int64_t o = opix * K + j;
    if (j >= qn) {
      a.p2f[o] = -1;
      a.zbuf[o] = a.dists[o] = a.bary[3 * o] = a.bary[3 * o + 1] = a.bary[3 * o + 2] = -1.0f;
    }
  }
  for (int j = lane; j < qn; j += kWave) {
    const float z = qz[j];
    const int f = qi[j];
    int rank = 0;
    for (int t = 0; t < qn; ++t) {
      const float zt = qz[t];
      const int ft = qi[t];
      rank += (zt < z || (zt == z && ft < f)) ? 1 : 0;
    }
    f3 v0, v1, v2;
    verts(f, &v0, &v1, &v2);
    FaceHit h;
    face_hit(v0, v1, v2, p, a.blur, persp, clip, &h);
    const int64_t o = opix * K + rank;
    a.p2f[o] = f;
    a.zbuf[o] = h.z;
    a.dists[o] = h.dist;
    a.bary[3 * o] = h.bary.x;
    a.bary[3 * o + 1] = h.bary.y;
    a.bary[3 * o + 2] = h.bary.z;
  }
  __builtin_amdgcn_wave_barrier();
}

// With the lane masks of the TIES kernels (MeshArgs::tie_words): a workgroup of eight waves takes 16 of the (N, SY, SX) words --
// scattered over the images by a multiplicative permutation: marks come in clusters (4.8 pixels per marked sub-tile at the
// bench workload, some sub-tiles dozens, in neighbouring sub-tiles) and a pixel is one long chain of dependent round trips --
// and its waves replay the marked pixels in turn.  Words without a mark (97 % at the bench workload) cost a sixteenth of a
// load instruction per wave.
constexpr int kTieWordsPerWave = 16;
constexpr int kTieWaves = 8;
template <bool BINNED>
__global__ __launch_bounds__(kTieWaves * kWave) void mesh_cuda_order_words_kernel(MeshArgs a, unsigned nwords, unsigned mult) {
  __shared__ float s_qzw[kTieWaves][P3D_MAX_K];
  __shared__ int s_qiw[kTieWaves][P3D_MAX_K];
  if (a.K <= 0) return;
  if (a.overflow != nullptr && (*a.overflow != 0) == BINNED) return;  // short workspaces: as in mesh_raster_kernel
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  volatile float* s_qz = s_qzw[wv];
  volatile int* s_qi = s_qiw[wv];
  int turn = 0;
  const unsigned slot = blockIdx.x * kTieWordsPerWave + lane;
  unsigned wi = 0;
  unsigned long long word = 0;
  if (lane < kTieWordsPerWave && slot < nwords) {
    wi = (unsigned)(((unsigned long long)slot * mult) % nwords);
    word = a.tie_words[wi];
  }
  unsigned long long have = __ballot(word != 0);
  while (have) {  // uniform
    const int l = __builtin_ctzll(have);
    have &= have - 1;
    const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)wi, l);
    unsigned long long todo = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(word >> 32), l) << 32) |
                              (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)word, l);
    const int sx = (int)(w % (unsigned)a.SX);
    const unsigned t = w / (unsigned)a.SX;
    const int sy = (int)(t % (unsigned)a.SY), n = (int)(t / (unsigned)a.SY);
    while (todo) {
      const int L = __builtin_ctzll(todo);
      todo &= todo - 1;
      if (((turn++) & (kTieWaves - 1)) != wv) continue;  // uniform: another wave's pixel
      replay_pixel<BINNED>(a, n, sy * 8 + (L >> 3), sx * 8 + (L & 7), s_qz, s_qi, lane);
    }
  }
}

// Without a workspace for the lane masks: every wave reads the marks of its own 8 x 8 sub-tile in place (a pix_to_face entry
// of every pixel of the launch) and replays what they mark.
template <bool BINNED>
__global__ __launch_bounds__(kStage) void mesh_cuda_order_kernel(MeshArgs a) {
  __shared__ float s_qz[kStage / kWave][P3D_MAX_K];
  __shared__ int s_qi[kStage / kWave][P3D_MAX_K];
  if (a.overflow != nullptr && (*a.overflow != 0) == BINNED) return;  // short workspaces: as in mesh_raster_kernel
  TileCoord tc;
  if (!tile_of_block(a.tm, blockIdx.x, &tc)) return;
  const int n = tc.n, H = a.H, W = a.W, K = a.K;
  if (K <= 0) return;
  const int y_end = min(H, (tc.by + 1) * a.tm.bin_size), x_end = min(W, (tc.bx + 1) * a.tm.bin_size);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int sy0 = tc.by * a.tm.bin_size + tc.ty * kTile + (w >> 1) * 8, sx0 = tc.bx * a.tm.bin_size + tc.tx * kTile + (w & 1) * 8;
  const int yi = sy0 + (lane >> 3), xi = sx0 + (lane & 7);
  const bool pix_ok = yi < y_end && xi < x_end;
  unsigned long long todo = __ballot(pix_ok && a.p2f[(((int64_t)n * H + (H - 1 - yi)) * W + (W - 1 - xi)) * K] == -2);
  while (todo) {  // uniform (no workgroup barriers in this kernel)
    const int L = __builtin_ctzll(todo);
    todo &= todo - 1;
    replay_pixel<BINNED>(a, n, sy0 + (L >> 3), sx0 + (L & 7), s_qz[w], s_qi[w], lane);
  }
}

// The instantiations one (Queue, K) pair can run as: split (few tiles), compile-time persp & clip, or plain.
template <typename Q, int KT, bool REGS, bool BINNED, bool EXACT, bool TIES>
void launch_fine_variant(const MeshArgs& a, unsigned grid, bool split, size_t dyn_lds, hipStream_t stream) {
  if constexpr (REGS && EXACT && BINNED && !TIES) {
    if (split) {
      // the merge slab (3 x 6 x KT x 64 floats beside the staging arrays) bounds the split kernel's occupancy: declare what
      // the LDS allows (K = 8: 66 KB -> 2 workgroups per CU, K = 4: 47 KB -> 3) instead of an unattainable 4
      constexpr int kSplitWaves = KT >= 8 ? 2 : (KT >= 4 ? 3 : kFineWaves);
      mesh_raster_kernel<Q, KT, REGS, BINNED, EXACT, kSplitWaves, false, true><<<grid * 4, kStage, 0, stream>>>(a);
      return;
    }
  }
  if constexpr (REGS && EXACT) {
    if (a.persp && a.clip) {
      mesh_raster_kernel<typename PcQueue<Q>::type, KT, REGS, BINNED, EXACT, kFineWaves, true, false, TIES><<<grid, kStage, dyn_lds, stream>>>(a);
      return;
    }
  }
  mesh_raster_kernel<Q, KT, REGS, BINNED, EXACT, kFineWaves, false, false, TIES><<<grid, kStage, dyn_lds, stream>>>(a);
}

// Queues without payload (K <= KT live entries): the perspective + clip instantiation (64-bit key compares; depths are
// >= +0 there: clipped barycentrics are >= +0 and faces with a vertex depth below 1e-8 never reach the queue, face_setup) when
// both flags are set.
template <int KT, bool BINNED, int WAVES, bool TIES>
void launch_long_variant(const MeshArgs& a, unsigned grid, hipStream_t stream) {
  if (a.persp && a.clip)
    mesh_raster_kernel<TopKPairs<KT, true, 0>, KT, true, BINNED, false, WAVES, true, false, TIES><<<grid, kStage, 0, stream>>>(a);
  else
    mesh_raster_kernel<TopKPairs<KT, false, 0>, KT, true, BINNED, false, WAVES, false, false, TIES><<<grid, kStage, 0, stream>>>(a);
}

#define P3D_COMMA ,
template <bool BINNED, bool TIES>
int launch_mesh_raster_t(const MeshArgs& a0, hipStream_t stream) {
  MeshArgs a = a0;
  unsigned grid = tile_grid(a.tm);
  const char* name = BINNED ? "mesh_fine" : "mesh_naive";
  LaunchScope ls(name, stream);
  const int K = a.K;
  // few tiles (one image, a small batch): one workgroup per sub-tile with the candidate list dealt to its four waves
  const bool split = BINNED && !TIES && grid <= (unsigned)kSplitMaxTiles;
  // tiles in the plan's order: when the lists carry a tile plan and a tile is a bin (grid = bins = active + background rows)
  a.walk_plan = BINNED && !split && a.csr.plan.hdr != nullptr && a.csr.plan.order != nullptr && a.tm.Ty == 1 && a.tm.Tx == 1;
  // the list of the cover's non-empty words from the plan instead of from atomics: tiles = bins = cover words
  a.list_by_plan = a.walk_plan && a.area_list != nullptr && a.overflow == nullptr && a.tm.bin_size == kTile && a.H % kTile == 0 && a.W % kTile == 0 &&
                   a.csr.plan.arank != nullptr;
  const size_t dyn_lds = 0;
#define P3D_LAUNCH_FINE(KT_, REGS_, EXACT_, Q_) launch_fine_variant<Q_, KT_, REGS_, BINNED, EXACT_, TIES>(a, grid, split, dyn_lds, stream)
#define P3D_LAUNCH_FINE_W(KT_, WAVES_, Q_) mesh_raster_kernel<Q_, KT_, true, BINNED, false, WAVES_, false, false, TIES><<<grid, kStage, 0, stream>>>(a)
#define P3D_LAUNCH_LONG(KT_, WAVES_) launch_long_variant<KT_, BINNED, WAVES_, TIES>(a, grid, stream)
  // Up to 8: queues WITH payload in registers.  K = 1, 2, 4, 8 have exact instantiations (vector-row epilogue, no
  // test on K left); 3 and 5..7 run the pair queue of the next capacity with K live entries (topk.h: TopKPairs::insert
  // skips the steps of the dead entries with scalar branches).  Until round 3 those K ran TopKReg queues of 8 / 12 entries
  // whose kernels sit at the register limit of their launch bounds; with round 4's pixel masks they spilled, and a kernel of
  // this file that spills VGPRs next to its SGPR spills loses queue entries (profiles/r04/spill_miscompile.md) -- the
  // build now refuses such kernels (pytorch3d_amd/build.py).
  if (K == 1)
    P3D_LAUNCH_FINE(1, true, true, TopKReg<1 P3D_COMMA kMeshPayload>);
  else if (K == 2)
    P3D_LAUNCH_FINE(2, true, true, TopKReg<2 P3D_COMMA kMeshPayload>);
  else if (K == 3)
    P3D_LAUNCH_FINE_W(4, kFineWaves, TopKPairs<4>);
  else if (K == 4)  // K = 4, 8: the queue in register pairs (topk.h: TopKPairs; one 64-bit key compare per entry in the
    P3D_LAUNCH_FINE(4, true, true, TopKPairs<4>);  // perspective + clip kernels): measured -3 % / -11 % / -10 % (profiles/r03)
  else if (K < 8)  // (2 VGPRs short of four waves per SIMD with the fill + patch epilogue: three)
    P3D_LAUNCH_FINE_W(8, 3, TopKPairs<8>);
  else if (K == 8)  // (the queue WITHOUT payload at five waves per SIMD, 81 registers: 1.36 -> 1.89 ms, profiles/r04/r04c10/k8long.txt)
    P3D_LAUNCH_FINE(8, true, true, TopKPairs<8>);
  // 9..48: queues without payload in register pairs (write_pixel_long recomputes distance and barycentrics of the survivors).
  // Five capacities; a queue serves every K up to its capacity at the cost of K entries (topk.h: TopKPairs::insert skips the
  // steps of the dead entries with scalar branches), so no exact-K instantiations here.  (A TopKReg of 32+ entries keeps its
  // 32+ comparison masks in SGPRs: they spill, and the build with spills lost candidates -- measured in round 4, not pursued.)
  // Measured against the payload queue of 16 entries at two waves per SIMD (8 / 64 bench meshes, profiles/r04/k_long16*.txt):
  // K = 12 0.40 -> 0.28 ms / 3.44 -> 2.33, K = 16 0.49 -> 0.39 / 4.41 -> 3.05 -- 100 registers, four waves per SIMD.
  else if (K <= 16)
    P3D_LAUNCH_LONG(16, 4);
  else if (K <= 24)
    P3D_LAUNCH_LONG(24, 3);
  else if (K <= 32)
    P3D_LAUNCH_LONG(32, 3);
  else if (K <= 40)
    P3D_LAUNCH_LONG(40, 2);
  else if (K <= 48)
    P3D_LAUNCH_LONG(48, 2);
  // (49..64: a 64-entry pair queue needs 128 + ~130 registers -- 70 VGPRs spilled at two waves per SIMD, refused by the build;
  // at one wave per SIMD the compiler parks 32 of them in AGPRs and that kernel lost entries on the GPU, run to run
  // differently, whatever was tried in round 4 (profiles/r04/spill_miscompile.md): the private-memory queue keeps K > 48.)
  else
    P3D_LAUNCH_FINE(P3D_MAX_K, false, false, TopKMem<P3D_MAX_K P3D_COMMA kMeshPayload>);
#undef P3D_LAUNCH_FINE
#undef P3D_LAUNCH_FINE_W
#undef P3D_LAUNCH_LONG
  return launch_status();
}

template <bool BINNED>
int launch_mesh_raster(const MeshArgs& a, hipStream_t stream) {
  return a.ties ? launch_mesh_raster_t<BINNED, true>(a, stream) : launch_mesh_raster_t<BINNED, false>(a, stream);
}

void set_tiles(MeshArgs* a, int bin_size, int BH, int BW) { a->tm = make_tile_map(a->N, a->H, a->W, bin_size, BH, BW, true); }

int check_common(int N, int H, int W, int K) {
  if (N < 0 || H < 0 || W < 0 || K < 0) return P3D_ERR_INVALID_ARG;
  if (K > P3D_MAX_K) return P3D_ERR_K_TOO_LARGE;
  return P3D_OK;
}

}  // namespace

}  // namespace p3d

using namespace p3d;

// CUDA tie order: one 64-bit lane mask per 8 x 8 sub-tile (MeshArgs::tie_words)
static size_t tie_marks_bytes(int N, int H, int W) {
  return align_up((size_t)N * (size_t)((H + 7) / 8) * (size_t)((W + 7) / 8) * sizeof(unsigned long long), 256);
}

P3D_API size_t p3d_rasterize_meshes_workspace_bytes(int64_t F, int N, int H, int W, int bin_size,
                                                    int max_faces_per_bin) {
  if (bin_size <= 0 || max_faces_per_bin <= 0 || N <= 0 || H <= 0 || W <= 0) return 0;
  // enough for both the caller's geometry (_rasterize_meshes_coarse) and the internal one (rasterize_meshes)
  const size_t user = bin_workspace_bytes(F, N, make_geom(H, W, bin_size), max_faces_per_bin);
  const size_t internal = bin_workspace_bytes(F, N, make_internal_geom(H, W, bin_size), max_faces_per_bin);
  return (user > internal ? user : internal) + 256 + tie_marks_bytes(N, H, W) + 256;  // (+ the marks of the CUDA tie order)
}

P3D_API size_t p3d_rasterize_meshes_short_workspace_bytes(int64_t F, int N, int H, int W, int bin_size, int max_faces_per_bin,
                                                          int64_t list_entries) {
  if (bin_size <= 0 || max_faces_per_bin <= 0 || N <= 0 || H <= 0 || W <= 0 || list_entries < 0) return 0;
  return bin_workspace_bytes(F, N, make_internal_geom(H, W, bin_size), max_faces_per_bin, list_entries) + 256;
}

P3D_API size_t p3d_rasterize_meshes_workspace_need_offset(int64_t F, int N, int H, int W, int bin_size,
                                                          int max_faces_per_bin) {
  if (bin_size <= 0 || max_faces_per_bin <= 0 || N <= 0 || H <= 0 || W <= 0) return 0;
  const BinGeom g = make_internal_geom(H, W, bin_size);
  Arena probe(nullptr, 0);
  BinWorkspace ws;
  bin_carve(probe, F, N, g, max_faces_per_bin, &ws, 1);
  return ws.need_at;
}

P3D_API size_t p3d_rasterize_fine_workspace_bytes(int N, int BH, int BW, int M) {
  const size_t rows = (size_t)N * BH * BW;
  return align_up(rows * (size_t)M * sizeof(int), 256) + align_up(rows * sizeof(int), 256) +
         align_up((rows + 1) * sizeof(int64_t), 256) + 256;
}

P3D_API size_t p3d_rasterize_meshes_cover_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)N * (size_t)((H + 15) / 16) * (size_t)((W + 15) / 16) * sizeof(int32_t);
}

P3D_API size_t p3d_rasterize_meshes_cover_list_bytes(int N, int H, int W) {
  const size_t words = p3d_rasterize_meshes_cover_bytes(N, H, W) / sizeof(int32_t);
  return words == 0 ? 0 : (2 * words + 16) * sizeof(int32_t);  // cover | count + 15 spare | list
}

// the cover starts empty; the waves that write a pixel with a face set its bit.  with_list: cover is a buffer of
// p3d_rasterize_meshes_cover_list_bytes -- the counter of the area list behind the words is zeroed in the same memset
static int cover_begin(MeshArgs* a, int32_t* cover, hipStream_t s, bool with_list = false) {
  a->cover = cover;
  a->CY = (a->H + 15) / 16;
  a->CX = (a->W + 15) / 16;
  a->area_count = nullptr;
  a->area_list = nullptr;
  if (cover == nullptr) return P3D_OK;
  const size_t bytes = p3d_rasterize_meshes_cover_bytes(a->N, a->H, a->W);
  if (with_list && bytes > 0) {
    a->area_count = cover + bytes / sizeof(int32_t);
    a->area_list = a->area_count + 16;
  }
  return (bytes == 0 || hipMemsetAsync(cover, 0, bytes + (with_list ? 16 * sizeof(int32_t) : 0), s) == hipSuccess) ? P3D_OK : P3D_ERR_LAUNCH;
}

// CUDA tie order: where the TIES kernels leave their lane masks (MeshArgs::tie_words), or words == null: marks in place only
struct TieMarks {
  unsigned long long* words = nullptr;
  int SY = 0, SX = 0;
};

static int mesh_naive_impl(const float* face_verts, const int64_t* mesh_first, const int64_t* mesh_count,
                           const int64_t* neighbor, int64_t F, int N, int H, int W, float blur_radius, int K, int persp,
                           int clip, int cull, int64_t* p2f, float* zbuf, float* bary, float* dists, int32_t* cover,
                           p3d_stream_t stream, const int* overflow = nullptr, bool ties = false, TieMarks marks = TieMarks(),
                           bool with_list = false) {
  (void)F;
  const int rc = check_common(N, H, W, K);
  if (rc != P3D_OK) return rc;
  MeshArgs z{};
  z.N = N;
  z.H = H;
  z.W = W;
  if (cover != nullptr && overflow == nullptr) {  // (as the fallback of a binned launch: that one zeroed the cover)
    const int st = cover_begin(&z, cover, (hipStream_t)stream, with_list);  // also for K == 0: nothing is covered
    if (st != P3D_OK) return st;
  } else if (cover != nullptr && with_list) {
    const size_t words = p3d_rasterize_meshes_cover_bytes(N, H, W) / sizeof(int32_t);
    z.area_count = cover + words;
    z.area_list = z.area_count + 16;
  }
  if ((int64_t)N * H * W * K == 0) return P3D_OK;
  if ((!face_verts || !neighbor) && F > 0) return P3D_ERR_INVALID_ARG;
  if (!mesh_first || !mesh_count || !p2f || !zbuf || !bary || !dists) return P3D_ERR_INVALID_ARG;
  MeshArgs a{};
  a.face_verts = face_verts;
  a.neighbor = neighbor;
  a.mesh_first = mesh_first;
  a.mesh_count = mesh_count;
  a.N = N;
  a.H = H;
  a.W = W;
  a.K = K;
  a.blur = blur_radius;
  a.sqrt_blur = sqrtf(blur_radius);
  a.persp = persp;
  a.clip = clip;
  a.cull = cull;
  a.p2f = p2f;
  a.zbuf = zbuf;
  a.bary = bary;
  a.dists = dists;
  a.cover = cover;  // zeroed above
  a.CY = (H + 15) / 16;
  a.CX = (W + 15) / 16;
  a.area_count = cover != nullptr ? z.area_count : nullptr;
  a.area_list = cover != nullptr ? z.area_list : nullptr;
  a.overflow = overflow;
  a.ties = ties ? 1 : 0;
  a.tie_words = marks.words;
  a.SY = marks.SY;
  a.SX = marks.SX;
  set_tiles(&a, H > W ? H : W, 1, 1);
  return launch_mesh_raster<false>(a, (hipStream_t)stream);
}

P3D_API int p3d_rasterize_meshes_naive(const float* face_verts, const int64_t* mesh_first, const int64_t* mesh_count,
                                       const int64_t* neighbor, int64_t F, int N, int H, int W, float blur_radius, int K,
                                       int persp, int clip, int cull, int64_t* p2f, float* zbuf, float* bary,
                                       float* dists, p3d_stream_t stream) {
  return mesh_naive_impl(face_verts, mesh_first, mesh_count, neighbor, F, N, H, W, blur_radius, K, persp, clip, cull, p2f,
                         zbuf, bary, dists, nullptr, stream);
}

static int mesh_fine_from_csr(const float* face_verts, const int64_t* neighbor, const BinCSR& csr, int N, int H, int W,
                              const BinGeom& g, float blur_radius, int K, int persp, int clip, int cull, int64_t* p2f,
                              float* zbuf, float* bary, float* dists, hipStream_t stream, int32_t* cover = nullptr,
                              const int* overflow = nullptr, bool ties = false, TieMarks marks = TieMarks(), bool with_list = false) {
  MeshArgs a{};
  a.face_verts = face_verts;
  a.neighbor = neighbor;
  a.csr = csr;
  a.N = N;
  a.H = H;
  a.W = W;
  a.K = K;
  a.blur = blur_radius;
  a.sqrt_blur = sqrtf(blur_radius);
  a.persp = persp;
  a.clip = clip;
  a.cull = cull;
  a.p2f = p2f;
  a.zbuf = zbuf;
  a.bary = bary;
  a.dists = dists;
  const int st = cover_begin(&a, cover, stream, with_list);
  if (st != P3D_OK) return st;
  a.overflow = overflow;
  a.ties = ties ? 1 : 0;
  a.tie_words = marks.words;
  a.SY = marks.SY;
  a.SX = marks.SX;
  set_tiles(&a, g.bin_size, g.BH, g.BW);
  return launch_mesh_raster<true>(a, stream);
}

// b -> (b * multiplier) mod items is a bijection for an odd multiplier coprime to items; near items / golden ratio it spreads
// neighbours far apart
static unsigned scatter_multiplier(uint64_t items) {
  if (items <= 1) return 1;
  auto gcd = [](uint64_t x, uint64_t y) {
    while (y) {
      const uint64_t r = x % y;
      x = y;
      y = r;
    }
    return x;
  };
  uint64_t m = (uint64_t)((double)items * 0.6180339887) | 1u;
  while (gcd(m, items) != 1) m += 2;
  return (unsigned)(m % items);
}

// the replay of p3d_rasterize_meshes_cuda_order over the outputs of the launches before it; csr: the bin lists (null: every
// face of the pixel's mesh), overflow: the short-workspace flag the launch obeys (or null)
static int cuda_order_replay(const float* face_verts, const int64_t* mesh_first, const int64_t* mesh_count,
                             const int64_t* neighbor, const BinCSR* csr, const BinGeom* g, int N, int H, int W, float blur_radius,
                             int K, int persp, int clip, int cull, int64_t* p2f, float* zbuf, float* bary, float* dists,
                             const int* overflow, hipStream_t s, TieMarks marks) {
  // diagnostic (profiles/tie_order_timing.py --count-marks): leave the marks in the output instead of replaying them
  static const bool skip = getenv("P3D_TIE_SKIP_REPLAY") != nullptr;
  if (skip) return P3D_OK;
  MeshArgs a{};
  a.tie_words = marks.words;
  a.SY = marks.SY;
  a.SX = marks.SX;
  a.face_verts = face_verts;
  a.neighbor = neighbor;
  a.mesh_first = mesh_first;
  a.mesh_count = mesh_count;
  a.N = N;
  a.H = H;
  a.W = W;
  a.K = K;
  a.blur = blur_radius;
  a.sqrt_blur = sqrtf(blur_radius);
  a.persp = persp;
  a.clip = clip;
  a.cull = cull;
  a.p2f = p2f;
  a.zbuf = zbuf;
  a.bary = bary;
  a.dists = dists;
  a.overflow = overflow;
  LaunchScope ls("mesh_cuda_order", s);
  const uint64_t nwords = (uint64_t)N * (uint64_t)marks.SY * (uint64_t)marks.SX;
  const bool words = marks.words != nullptr && nwords > 0 && nwords < 0xffffffffull;
  const unsigned wgrid = (unsigned)ceil_div((int64_t)nwords, kTieWordsPerWave);
  const unsigned mult = words ? scatter_multiplier(nwords) : 1u;
  if (!words) a.tie_words = nullptr;
  if (csr != nullptr) {
    a.csr = *csr;
    set_tiles(&a, g->bin_size, g->BH, g->BW);
    if (words)
      mesh_cuda_order_words_kernel<true><<<wgrid, kTieWaves * kWave, 0, s>>>(a, (unsigned)nwords, mult);
    else
      mesh_cuda_order_kernel<true><<<tile_grid(a.tm), kStage, 0, s>>>(a);
  } else {
    set_tiles(&a, H > W ? H : W, 1, 1);
    if (words)
      mesh_cuda_order_words_kernel<false><<<wgrid, kTieWaves * kWave, 0, s>>>(a, (unsigned)nwords, mult);
    else
      mesh_cuda_order_kernel<false><<<tile_grid(a.tm), kStage, 0, s>>>(a);
  }
  return launch_status();
}

static int raster_meshes_impl(const float* face_verts, const int64_t* mesh_first, const int64_t* mesh_count,
                              const int64_t* neighbor, int64_t F, int N, int H, int W, float blur_radius, int K, int bin_size,
                              int max_faces_per_bin, int persp, int clip, int cull, int64_t* p2f, float* zbuf, float* bary,
                              float* dists, int32_t* cover, void* workspace, size_t workspace_bytes, p3d_stream_t stream,
                              bool cuda_order, bool with_list = false) {
  hipStream_t s = (hipStream_t)stream;
  const bool any_output = (int64_t)N * H * W * K != 0;
  // CUDA tie order: the lane masks of the marked pixels live in the LAST bytes of the workspace when it has room for them
  // (p3d_rasterize_meshes_workspace_bytes counts them in); the binning carves the rest as ever
  TieMarks marks;
  const size_t full_workspace_bytes = workspace_bytes;
  // The words are indexed by (sy0 >> 3, sx0 >> 3) and decoded as pixels (8 sy + L / 8, 8 sx + L % 8): sub-tile origins must be
  // multiples of 8.  They are whenever a bin is a whole number of 8 x 8 sub-tiles -- not with a caller's bin_size of 9..15
  // (or any other non-multiple below a tile), which make_internal_geom honours: sub-tiles of neighbouring bins then share a
  // word and the last store wins.  Those launches mark in place only.
  const bool words_ok = bin_size <= 0 || max_faces_per_bin <= 0 || make_internal_geom(H > 0 ? H : 1, W > 0 ? W : 1, bin_size).bin_size % 8 == 0;
  if (cuda_order && words_ok && any_output && workspace != nullptr && N > 0 && H > 0 && W > 0) {
    const size_t mb = tie_marks_bytes(N, H, W);
    if (workspace_bytes >= mb + 256) {
      const size_t at = (workspace_bytes - mb) & ~(size_t)255;
      if (((uintptr_t)workspace & 7u) == 0) {
        marks.words = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + at);
        marks.SY = (H + 7) / 8;
        marks.SX = (W + 7) / 8;
        workspace_bytes = at;
        if (hipMemsetAsync(marks.words, 0, mb, s) != hipSuccess) return P3D_ERR_LAUNCH;
      }
    }
  }
  if (bin_size <= 0 || max_faces_per_bin <= 0) {
    const int st = mesh_naive_impl(face_verts, mesh_first, mesh_count, neighbor, F, N, H, W, blur_radius, K, persp, clip, cull,
                                   p2f, zbuf, bary, dists, cover, stream, nullptr, cuda_order, marks, with_list);
    if (st != P3D_OK || !cuda_order || !any_output) return st;
    return cuda_order_replay(face_verts, mesh_first, mesh_count, neighbor, nullptr, nullptr, N, H, W, blur_radius, K, persp, clip,
                             cull, p2f, zbuf, bary, dists, nullptr, s, marks);
  }
  const int rc = check_common(N, H, W, K);
  if (rc != P3D_OK) return rc;
  if (cover != nullptr && !any_output) {
    MeshArgs z{};
    z.N = N;
    z.H = H;
    z.W = W;
    return cover_begin(&z, cover, s, with_list);
  }
  if (!any_output) return P3D_OK;
  if ((!face_verts || !neighbor) && F > 0) return P3D_ERR_INVALID_ARG;
  if (!mesh_first || !mesh_count || !p2f || !zbuf || !bary || !dists) return P3D_ERR_INVALID_ARG;
  const BinGeom gu = make_geom(H, W, bin_size);
  if (gu.BH > P3D_MAX_BINS_PER_SIDE || gu.BW > P3D_MAX_BINS_PER_SIDE) return P3D_ERR_TOO_MANY_BINS;
  const BinGeom g = make_internal_geom(H, W, bin_size);  // tile-sized bins: results do not depend on the binning
  Arena arena(workspace, workspace_bytes);
  BinWorkspace ws;
  // a short workspace is welcome here (binning.h): the list takes what the caller gave, and the naive kernel stands by
  if (!workspace) return P3D_ERR_WORKSPACE;
  if (!bin_carve(arena, F, N, g, max_faces_per_bin, &ws, /*list_entries=*/1)) {
    // the marks of the CUDA tie order took the room the fixed arrays need: give it back (the replay reads the marks in place)
    if (marks.words == nullptr) return P3D_ERR_WORKSPACE;
    marks = TieMarks();
    workspace_bytes = full_workspace_bytes;
    arena = Arena(workspace, workspace_bytes);
    if (!bin_carve(arena, F, N, g, max_faces_per_bin, &ws, /*list_entries=*/1)) return P3D_ERR_WORKSPACE;
  }
  const bool is_short = ws.capacity < ws.worst;
  const int* overflow = is_short ? ws.plan_hdr + 2 : nullptr;
  int st = bin_build(kTriangles, face_verts, nullptr, mesh_first, mesh_count, F, N, g, max_faces_per_bin,
                     sqrtf(blur_radius), ws, s);
  if (st != P3D_OK) return st;
  BinCSR csr{ws.offset, ws.total, ws.list, TilePlan{ws.arank, ws.bg_list, ws.plan_hdr, ws.order}};
  st = mesh_fine_from_csr(face_verts, neighbor, csr, N, H, W, g, blur_radius, K, persp, clip, cull, p2f, zbuf, bary, dists, s,
                          cover, overflow, cuda_order, marks, with_list);
  if (st == P3D_OK && is_short)
    st = mesh_naive_impl(face_verts, mesh_first, mesh_count, neighbor, F, N, H, W, blur_radius, K, persp, clip, cull, p2f, zbuf,
                         bary, dists, cover, stream, overflow, cuda_order, marks, with_list);
  if (st != P3D_OK || !cuda_order) return st;
  st = cuda_order_replay(face_verts, mesh_first, mesh_count, neighbor, &csr, &g, N, H, W, blur_radius, K, persp, clip, cull, p2f,
                         zbuf, bary, dists, overflow, s, marks);
  if (st != P3D_OK || !is_short) return st;
  return cuda_order_replay(face_verts, mesh_first, mesh_count, neighbor, nullptr, nullptr, N, H, W, blur_radius, K, persp, clip,
                           cull, p2f, zbuf, bary, dists, overflow, s, marks);
}

P3D_API int p3d_rasterize_meshes_with_cover(const float* face_verts, const int64_t* mesh_first, const int64_t* mesh_count,
                                            const int64_t* neighbor, int64_t F, int N, int H, int W, float blur_radius,
                                            int K, int bin_size, int max_faces_per_bin, int persp, int clip, int cull,
                                            int64_t* p2f, float* zbuf, float* bary, float* dists, int32_t* cover,
                                            void* workspace, size_t workspace_bytes, p3d_stream_t stream) {
  return raster_meshes_impl(face_verts, mesh_first, mesh_count, neighbor, F, N, H, W, blur_radius, K, bin_size, max_faces_per_bin,
                            persp, clip, cull, p2f, zbuf, bary, dists, cover, workspace, workspace_bytes, stream, false);
}

P3D_API int p3d_rasterize_meshes_with_cover_list(const float* face_verts, const int64_t* mesh_first, const int64_t* mesh_count,
                                                 const int64_t* neighbor, int64_t F, int N, int H, int W, float blur_radius,
                                                 int K, int bin_size, int max_faces_per_bin, int persp, int clip, int cull,
                                                 int64_t* p2f, float* zbuf, float* bary, float* dists, int32_t* cover_and_list,
                                                 void* workspace, size_t workspace_bytes, p3d_stream_t stream) {
  return raster_meshes_impl(face_verts, mesh_first, mesh_count, neighbor, F, N, H, W, blur_radius, K, bin_size, max_faces_per_bin,
                            persp, clip, cull, p2f, zbuf, bary, dists, cover_and_list, workspace, workspace_bytes, stream, false,
                            cover_and_list != nullptr);
}

P3D_API int p3d_rasterize_meshes_cuda_order(const float* face_verts, const int64_t* mesh_first, const int64_t* mesh_count,
                                            const int64_t* neighbor, int64_t F, int N, int H, int W, float blur_radius,
                                            int K, int bin_size, int max_faces_per_bin, int persp, int clip, int cull,
                                            int64_t* p2f, float* zbuf, float* bary, float* dists, int32_t* cover,
                                            void* workspace, size_t workspace_bytes, p3d_stream_t stream) {
  return raster_meshes_impl(face_verts, mesh_first, mesh_count, neighbor, F, N, H, W, blur_radius, K, bin_size, max_faces_per_bin,
                            persp, clip, cull, p2f, zbuf, bary, dists, cover, workspace, workspace_bytes, stream, true);
}

P3D_API int p3d_rasterize_meshes(const float* face_verts, const int64_t* mesh_first, const int64_t* mesh_count,
                                 const int64_t* neighbor, int64_t F, int N, int H, int W, float blur_radius, int K,
                                 int bin_size, int max_faces_per_bin, int persp, int clip, int cull, int64_t* p2f,
                                 float* zbuf, float* bary, float* dists, void* workspace, size_t workspace_bytes,
                                 p3d_stream_t stream) {
  return p3d_rasterize_meshes_with_cover(face_verts, mesh_first, mesh_count, neighbor, F, N, H, W, blur_radius, K, bin_size,
                                         max_faces_per_bin, persp, clip, cull, p2f, zbuf, bary, dists, nullptr, workspace,
                                         workspace_bytes, stream);
}

P3D_API int p3d_rasterize_meshes_coarse(const float* face_verts, const int64_t* mesh_first, const int64_t* mesh_count,
                                        int64_t F, int N, int H, int W, float blur_radius, int bin_size,
                                        int max_faces_per_bin, int32_t* bin_faces, void* workspace,
                                        size_t workspace_bytes, p3d_stream_t stream) {
  if (N < 0 || H <= 0 || W <= 0 || bin_size <= 0 || max_faces_per_bin < 0) return P3D_ERR_INVALID_ARG;
  const BinGeom g = make_geom(H, W, bin_size);
  if (g.BH > P3D_MAX_BINS_PER_SIDE || g.BW > P3D_MAX_BINS_PER_SIDE) return P3D_ERR_TOO_MANY_BINS;
  if ((int64_t)N * g.nbins * max_faces_per_bin == 0) return P3D_OK;
  if (!mesh_first || !mesh_count || !bin_faces || (!face_verts && F > 0)) return P3D_ERR_INVALID_ARG;
  Arena arena(workspace, workspace_bytes);
  BinWorkspace ws;
  if (!workspace || !bin_carve(arena, F, N, g, max_faces_per_bin, &ws)) return P3D_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int st = bin_build(kTriangles, face_verts, nullptr, mesh_first, mesh_count, F, N, g, max_faces_per_bin,
                     sqrtf(blur_radius), ws, s);
  if (st != P3D_OK) return st;
  return bin_expand_padded(ws, N, g, max_faces_per_bin, bin_faces, s);
}

P3D_API int p3d_rasterize_meshes_fine(const float* face_verts, const int32_t* bin_faces, const int64_t* neighbor,
                                      int64_t F, int N, int BH, int BW, int M, int H, int W, float blur_radius,
                                      int bin_size, int K, int persp, int clip, int cull, int64_t* p2f, float* zbuf,
                                      float* bary, float* dists, void* workspace, size_t workspace_bytes,
                                      p3d_stream_t stream) {
  const int rc = check_common(N, H, W, K);
  if (rc != P3D_OK) return rc;
  if ((int64_t)N * H * W * K == 0) return P3D_OK;
  if (bin_size <= 0 || BH <= 0 || BW <= 0 || M < 0) return P3D_ERR_INVALID_ARG;
  if (((!face_verts || !neighbor) && F > 0) || !p2f || !zbuf || !bary || !dists) return P3D_ERR_INVALID_ARG;
  if (BH > P3D_MAX_BINS_PER_SIDE || BW > P3D_MAX_BINS_PER_SIDE) return P3D_ERR_TOO_MANY_BINS;
  // the bins handed in must tile the image the way the coarse stage would have
  if ((int64_t)BH * bin_size < H || (int64_t)BW * bin_size < W) return P3D_ERR_INVALID_ARG;
  if (workspace_bytes < p3d_rasterize_fine_workspace_bytes(N, BH, BW, M) || !workspace) return P3D_ERR_WORKSPACE;
  if (M > 0 && !bin_faces) return P3D_ERR_INVALID_ARG;
  const int64_t rows = (int64_t)N * BH * BW;
  Arena arena(workspace, workspace_bytes);
  int* list = arena.take<int>((size_t)rows * M);
  int* total = arena.take<int>((size_t)rows);
  int64_t* offset = arena.take<int64_t>((size_t)rows + 1);
  hipStream_t s = (hipStream_t)stream;
  int st = bin_compact_padded(bin_faces, rows, M, list, total, offset, s);
  if (st != P3D_OK) return st;
  BinGeom g;
  g.H = H;
  g.W = W;
  g.bin_size = bin_size;
  g.BH = BH;
  g.BW = BW;
  g.nbins = BH * BW;
  BinCSR csr{offset, total, list, TilePlan{nullptr, nullptr, nullptr, nullptr}};
  return mesh_fine_from_csr(face_verts, neighbor, csr, N, H, W, g, blur_radius, K, persp, clip, cull, p2f, zbuf, bary,
                            dists, s);
}
