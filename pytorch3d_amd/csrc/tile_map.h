// tile_map.h -- workgroup -> (batch element, bin, tile) assignment shared by the fine rasterizers.
//
// A bin (bin_size x bin_size pixels, one CSR list) is cut into 16x16-pixel tiles, one workgroup
// each.  The dispatcher places workgroup b on XCD b % 8 (observed, used for speed only):
//   * the tiles of one bin get consecutive slots of ONE XCD, so the bin's list and vertex records
//     are fetched into a single L2;
//   * each XCD owns an eighth of the bin indices, and the bin index is run through an affine
//     permutation (multiplier coprime to the bin count): every XCD then gets an even sample of all
//     batch elements and image regions (a contiguous eighth of the batch per XCD left whole XCDs
//     idle behind the heaviest meshes), and busy bins (the projected primitive) and empty bins
//     (background, pure -1 stores), which come in long runs in (n, by, bx) order, are mixed on every
//     CU, so the store-bound and the ALU-bound tiles overlap.
#pragma once

#include "p3d_common.h"

namespace p3d {

constexpr int kTilePx = 16;

struct TileMap {
  int N, BH, BW, Ty, Tx;      // batch, bins per image, tiles per bin
  int bin_size;
  long long bins;             // N * BH * BW
  long long bin_mult;         // odd multiplier coprime to `bins` (1 = identity)
  long long groups_per_xcd;   // ceil(bins / 8)
  long long xcd_rot;          // per-XCD rotation of the walk through its own range (decorrelates the XCDs in time)
};

inline TileMap make_tile_map(int N, int H, int W, int bin_size, int BH, int BW, bool permute) {
  TileMap m;
  m.N = N;
  m.BH = BH;
  m.BW = BW;
  m.bin_size = bin_size;
  const int span_y = bin_size < H ? bin_size : H;
  const int span_x = bin_size < W ? bin_size : W;
  m.Ty = (int)ceil_div(span_y, kTilePx);
  m.Tx = (int)ceil_div(span_x, kTilePx);
  m.bins = (long long)N * BH * BW;
  m.groups_per_xcd = ceil_div(m.bins, 8);
  long long mult = 1;
  if (permute && m.bins > 8) {
    mult = (long long)((double)m.bins * 0.6180339887) | 1;  // near bins / golden ratio
    auto gcd = [](long long x, long long y) {
      while (y) {
        const long long t = x % y;
        x = y;
        y = t;
      }
      return x;
    };
    while (gcd(mult, m.bins) != 1) mult += 2;
  }
  m.bin_mult = mult;
  m.xcd_rot = permute ? (((long long)((double)m.groups_per_xcd * 0.3819660113)) | 1) : 0;
  return m;
}

inline unsigned tile_grid(const TileMap& m) { return (unsigned)(m.groups_per_xcd * 8 * m.Ty * m.Tx); }

#if defined(__HIPCC__)
struct TileCoord {
  int n, by, bx, ty, tx;
};

// false: this workgroup has no tile (grid padding)
__device__ __forceinline__ bool tile_of_block(const TileMap& m, unsigned block, TileCoord* c) {
  const int tpb = m.Ty * m.Tx;
  const long long slot = block / 8;
  // XCD x = block % 8 owns the contiguous range [x * G, (x + 1) * G) of pre-permutation bin indices; the
  // multiplicative permutation scatters every such range evenly over the batch.  (Dealing bins
  // round-robin -- index % 8 -- BEFORE the permutation ties the XCD to a residue class of the
  // permuted index whenever 8 divides the bin count, i.e. to fixed image columns.)
  const long long grp = slot / tpb;
  if (grp >= m.groups_per_xcd) return false;
  // every XCD starts its walk at a different phase: without this all eight XCDs sit on the same image
  // position of eight different batch elements at any moment (their ranges are bins/8 apart and the
  // permutation is linear), so the whole chip alternates between store-bound and ALU-bound phases
  const long long x = block % 8;
  long long bin = x * m.groups_per_xcd + (grp + x * m.xcd_rot) % m.groups_per_xcd;
  if (bin >= m.bins) return false;
  bin = (long long)(((unsigned long long)bin * (unsigned long long)m.bin_mult) % (unsigned long long)m.bins);
  const int t = (int)(slot % tpb);
  c->tx = t % m.Tx;
  c->ty = t / m.Tx;
  c->bx = (int)(bin % m.BW);
  bin /= m.BW;
  c->by = (int)(bin % m.BH);
  c->n = (int)(bin / m.BH);
  return true;
}
#endif

}  // namespace p3d
