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
  unsigned long long bins_magic;  // floor((2^64 - 1) / bins): x % bins without a 64-bit division on the device
  unsigned long long grp_magic;   // the same for groups_per_xcd
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
  m.bins_magic = m.bins > 0 ? ~0ull / (unsigned long long)m.bins : 0;
  m.grp_magic = m.groups_per_xcd > 0 ? ~0ull / (unsigned long long)m.groups_per_xcd : 0;
  m.xcd_rot = permute ? (((long long)((double)m.groups_per_xcd * 0.3819660113)) | 1) : 0;
  return m;
}

inline unsigned tile_grid(const TileMap& m) { return (unsigned)(m.groups_per_xcd * 8 * m.Ty * m.Tx); }

#if defined(__HIPCC__)
struct TileCoord {
  int n, by, bx, ty, tx;
};

// x % d for x < 2^63 given magic = floor((2^64 - 1) / d): the quotient estimate mulhi(x, magic) is at most 2 low.
__device__ __forceinline__ unsigned long long fast_mod(unsigned long long x, unsigned long long d,
                                                       unsigned long long magic) {
  unsigned long long r = x - __umul64hi(x, magic) * d;
  if (r >= d) r -= d;
  if (r >= d) r -= d;
  return r;
}

// false: this workgroup has no tile (grid padding).  32-bit arithmetic wherever the ranges allow: a runtime
// 64-bit division is ~200 instructions on gfx950, and every one of 65536 workgroups decodes its tile.
__device__ __forceinline__ bool tile_of_block(const TileMap& m, unsigned block, TileCoord* c) {
  const unsigned tpb = (unsigned)(m.Ty * m.Tx);
  const unsigned slot = block >> 3, x = block & 7u;
  const unsigned grp = slot / tpb, t = slot - grp * tpb;
  if ((long long)grp >= m.groups_per_xcd) return false;
  // XCD x owns the contiguous range [x * G, (x + 1) * G) of pre-permutation bin indices; the multiplicative
  // permutation scatters every such range evenly over the batch.  (Dealing bins round-robin -- index % 8 --
  // BEFORE the permutation ties the XCD to a residue class of the permuted index whenever 8 divides the bin
  // count, i.e. to fixed image columns.)  Every XCD starts its walk at a different phase: without it all eight
  // XCDs sit on the same image position of eight different batch elements at any moment (their ranges are
  // bins/8 apart and the permutation is linear), so the whole chip alternates between store-bound and
  // ALU-bound phases.
  const unsigned long long G = (unsigned long long)m.groups_per_xcd;
  unsigned long long walk = (unsigned long long)grp + (unsigned long long)x * (unsigned long long)m.xcd_rot;
  if (walk >= G) walk = fast_mod(walk, G, m.grp_magic);
  unsigned long long bin = (unsigned long long)x * G + walk;
  if (bin >= (unsigned long long)m.bins) return false;
  if (m.bin_mult != 1) bin = fast_mod(bin * (unsigned long long)m.bin_mult, (unsigned long long)m.bins, m.bins_magic);
  c->tx = (int)(t % (unsigned)m.Tx);
  c->ty = (int)(t / (unsigned)m.Tx);
  const unsigned per_image = (unsigned)(m.BH * m.BW);
  const unsigned b32 = (unsigned)bin;  // the grid is 32-bit, so bins < 2^32
  const unsigned n = b32 / per_image;
  const unsigned rem = b32 - n * per_image;
  c->bx = (int)(rem % (unsigned)m.BW);
  c->by = (int)(rem / (unsigned)m.BW);
  c->n = (int)n;
  return true;
}
#endif

}  // namespace p3d
