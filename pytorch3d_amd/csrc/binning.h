// binning.h -- coarse stage: per-(batch element, bin) ascending CSR lists of primitives.
//
// Replaces TriangleBoundingBoxKernel / PointBoundingBoxKernel / RasterizeCoarseCudaKernel
// (pytorch3d/csrc/rasterize_coarse/rasterize_coarse.cu:20-219) with a count -> scan -> fill
// pipeline: deterministic, sorted, exactly sized, O(E) instead of O(N * E).
#pragma once

#include "p3d_common.h"

namespace p3d {

constexpr int kBinChunk = 1024;  // primitives per workgroup in the count / fill passes
// The binning kernels handle up to 32 x 32 bins per image: more than the operator interface admits
// (P3D_MAX_BINS_PER_SIDE = 21, the reference's limit), because the fused rasterizers bin internally
// at tile granularity (make_internal_geom).
constexpr int kMaxBinsSide = 32;
constexpr int kMaxBins = kMaxBinsSide * kMaxBinsSide;

enum BinKind { kTriangles = 0, kPoints = 1 };

struct BinGeom {
  int H, W, bin_size, BH, BW, nbins;
};

inline BinGeom make_geom(int H, int W, int bin_size) {
  BinGeom g;
  g.H = H;
  g.W = W;
  g.bin_size = bin_size;
  g.BH = 1 + (H - 1) / bin_size;
  g.BW = 1 + (W - 1) / bin_size;
  g.nbins = g.BH * g.BW;
  return g;
}

// Geometry the fused operators (rasterize_meshes / rasterize_points with bin_size > 0) bin with.
// Which bins the coarse stage uses is invisible in their results (the K nearest per pixel do not
// depend on it), so they use bins of one 16x16 tile -- every workgroup of the fine stage then
// streams only primitives that touch its own tile, instead of its whole 32x32 (or larger) bin --
// doubling the bin side until at most 32 bins span the image, and never coarser than the caller's
// bin_size.  The test-visible _rasterize_*_coarse / _fine operators keep the caller's geometry.
// Overflow: the caller's max_faces_per_bin caps every INTERNAL bin's list.  An internal bin that lies inside one user bin
// (always the case when the user bin_size is a multiple of the internal one, e.g. the reference's defaults 16..128 vs
// 16) holds a subset of that user bin's faces, so it overflows only where the reference's bin would have ("Bin size
// was too small", rasterize_coarse.cu:186-201: faces dropped there too); with a non-multiple user bin_size an internal bin
// can straddle two user bins and reach the cap a little earlier than either of them.
inline BinGeom make_internal_geom(int H, int W, int user_bin_size) {
  int b = 16;
  const int m = H > W ? H : W;
  while ((m + b - 1) / b > kMaxBinsSide) b *= 2;
  if (user_bin_size < b) {
    // the caller asked for finer bins than a tile: honour them when the kernels can (<= 32 per side)
    if ((m + user_bin_size - 1) / user_bin_size <= kMaxBinsSide) b = user_bin_size;
  }
  return make_geom(H, W, b);
}

// Which bins hold primitives ("active") and which are background, written by the offsets scan at no extra launch:
//   arank[row]  number of active rows before `row`          (valid for every row)
//   bg_list[j]  the j-th background row, ascending           (j < hdr[1])
//   order[e]    the active rows by descending list length    (e < hdr[0]; kPlanClasses classes of 8 primitives,
//               binning.hip: plan_class), valid when hdr[3] != 0 (not written by the single-workgroup scan of small launches)
//   hdr         {A = active rows, B = background rows, lists overflow the workspace (short workspaces, below), order valid},
//               then the class counts and cursors of the sort
// The fine rasterizers use it to walk the tiles in a balanced order (raster_mesh.hip: "Which tile") and to let the
// workgroups of active tiles write the -1 fill of the background tiles ("piggyback fill"): active row number r fills
// background rows [r * q, (r + 1) * q), q = ceil(B / A).
constexpr int kPlanClasses = 64;  // list-length classes of the tile order
constexpr int kPlanHdr = 4;       // first class counter in plan_hdr
constexpr int kPlanTicket = kPlanHdr + 2 * kPlanClasses;  // arrival counter of the row scan's workgroups (bin_scan_rows_tail_kernel)
constexpr int kPlanHdrInts = kPlanTicket + 4;
struct TilePlan {
  const int* arank;
  const int* bg_list;
  const int* hdr;
  const int* order;
};

// Device-side CSR view consumed by the fine kernels.
struct BinCSR {
  const int64_t* offset;  // (N*nbins) start of each bin's list inside `list`
  const int* total;       // (N*nbins) entries in each bin's list
  const int* list;        // primitive ids (packed, global), ascending per bin
  TilePlan plan;          // null pointers when the lists did not come from bin_build (user-supplied bins)
  int stride = 1;         // ints per list entry.  2 (points, round 5): entry i = (id, depth bits) at list[2 i], list[2 i + 1] --
                          // bin_fill writes both with one 8-byte store, and the fine kernel's depth sort reads the depths with
                          // the ids (one coalesced round trip) instead of gathering them from the points afterwards
};

struct BinWorkspace {
  int* chunk_start;  // (N+1)
  int* counts;       // (max_chunks * nbins)
  int* total;        // (N*nbins)
  int64_t* offset;   // (N*nbins + 1)
  long long* blocksum;  // (ceil(N*nbins / 1024) + 1) scratch of the offsets scan
  int* arank;        // (N*nbins)  TilePlan
  int* bg_list;      // (N*nbins)
  int* plan_hdr;     // (kPlanHdrInts: 4 + 2 * kPlanClasses + the ticket)
  int* order;        // (N*nbins)
  int* list;         // (capacity) -- last, so that a short workspace shortens only this
  int stride;        // ints per list entry: 1, or 2 = (id, depth bits) (bin_carve with_z: points)
  int64_t max_chunks;
  int64_t capacity;  // ids `list` holds
  int64_t worst;     // bin_capacity(): the most the lists can ever need
  size_t need_at;    // byte offset of offset[rows] inside the arena (what a caller of a short workspace reads back)
};

// Short workspaces.  The worst case of the lists (every primitive in every bin, capped by M per bin) is 100-1000 x what a
// real batch needs (bench batch: 1.3 GB against 8.4 MB), and the exact size is only known on the device, after the scan.
// A caller that has a fallback for the overflow case may therefore carve with list_entries >= 0: the list gets
// max(list_entries, what is left of the arena) ids, capped at the worst case.  The offsets scan compares the lists' total
// with the capacity and raises plan_hdr[2]; bin_fill then writes nothing, and the caller's consumers test the flag ON THE
// DEVICE (raster_mesh.hip: p3d_rasterize_meshes_with_cover launches the binned kernel, which returns at once when the
// flag is up, and the naive kernel, which returns at once when it is not) -- no host sync, exact either way.  offset[rows]
// holds the total the call needed; a caller reads it back later to size its next workspace.
int64_t bin_capacity(int64_t E, int N, const BinGeom& g, int M);
size_t bin_workspace_bytes(int64_t E, int N, const BinGeom& g, int M, int64_t list_entries = -1, bool with_z = false);
// Carve `arena`; returns false when it is too small.  list_entries < 0: the worst case or nothing.
// with_z: list entries of two ints, (id, depth bits) (BinWorkspace::stride).
bool bin_carve(Arena& arena, int64_t E, int N, const BinGeom& g, int M, BinWorkspace* ws, int64_t list_entries = -1, bool with_z = false);

// Build the CSR lists.  elems: face_verts (E,3,3) or points (E,3); aux: radius (E) for points.
// ordered = false (points only): ids inside a 1024-primitive chunk land in arrival order (integer LDS
// atomics) instead of ascending order -- for consumers whose result is order-free; which ids survive
// an overflowing bin (> M) is then unspecified, as in the reference (rasterize_coarse.cu:186-201).
int bin_build(BinKind kind, const float* elems, const float* aux, const int64_t* first, const int64_t* count, int64_t E,
              int N, const BinGeom& g, int M, float sqrt_blur, const BinWorkspace& ws, hipStream_t stream,
              bool ordered = true);

// out[i] = sum(in[0..i)) for i in 0..n (n + 1 outputs, int64); blocksum: scratch of ceil(n / 1024) + 1 entries.
int exclusive_scan_i32(const int* in, int64_t n, long long* blocksum, int64_t* out, hipStream_t stream);

// CSR -> the reference's padded (N,BH,BW,M) int32 layout, -1 filled.
int bin_expand_padded(const BinWorkspace& ws, int N, const BinGeom& g, int M, int32_t* out, hipStream_t stream);

// Padded (N,BH,BW,M) with -1 sentinels anywhere -> compacted lists at row*M inside ws_list, totals, offsets.
int bin_compact_padded(const int32_t* padded, int64_t rows, int M, int* ws_list, int* ws_total, int64_t* ws_offset,
                       hipStream_t stream);

}  // namespace p3d
