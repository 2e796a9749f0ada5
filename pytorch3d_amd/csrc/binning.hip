// binning.hip -- coarse rasterization for gfx950: count -> scan -> fill.
//
// What it computes is RasterizeCoarseCudaKernel's bin membership
// (pytorch3d/csrc/rasterize_coarse/rasterize_coarse.cu:143-167), with the same float
// expressions for the bin edges so that membership is bit-identical; how it computes it is
// different: a workgroup owns 1024 consecutive primitives of ONE batch element, every
// primitive derives its bin rectangle from two monotone edge tables held in LDS (the
// reference brute-forces all bins), waves count members per bin with a 64-bit ballot +
// popcount, and a second pass re-derives the ballots and places ids at
// offset + mbcnt-prefix.  Lists come out ascending and exactly sized; nothing is pre-filled.
#include "binning.h"
#include "p3d_geom.h"

namespace p3d {

namespace {

constexpr int kWavesPerChunk = kBinChunk / kWave;  // 16

// ---------------------------------------------------------------------------------------
// plan: chunk_start[n] = sum_{m<n} ceil(count[m] / 1024); chunk_start[N] = total chunks.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bin_plan_kernel(const int64_t* __restrict__ count, int N,
                                                        int* __restrict__ chunk_start, int* __restrict__ plan_hdr) {
  __shared__ int scan[1024];
  __shared__ int carry_s;
  const int tid = threadIdx.x;
  if (tid < 2 * kPlanClasses) plan_hdr[kPlanHdr + tid] = 0;  // class histogram and cursors of the tile order (plan_class)
  if (tid == 0) plan_hdr[kPlanTicket] = 0;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + tid;
    int v = 0;
    if (i < N) {
      const int64_t c = count[i];
      v = c > 0 ? (int)((c + kBinChunk - 1) / kBinChunk) : 0;
    }
    scan[tid] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int add = tid >= d ? scan[tid - d] : 0;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    const int carry = carry_s;
    if (i < N) chunk_start[i] = carry + scan[tid] - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + scan[1023];
    __syncthreads();
  }
  if (tid == 0) chunk_start[N] = carry_s;
}

// Per-primitive bin rectangle.  lo > hi means "covers nothing".
struct BinRect {
  int x0, x1, y0, y1;
};

struct ChunkCtx {
  int n;        // batch element
  int64_t e;    // this thread's primitive (global packed id), -1 if none
  BinRect r;    // its rectangle
  BinRect u;    // union over the wave
  float z;      // points: the primitive's depth (bin_fill writes it beside the id when BinWorkspace::stride == 2)
};

// Small launches (N <= kSelfPlanMax, see bin_build) skip the plan kernel: every workgroup derives the chunk table
// itself, one wave scanning ceil(count / 1024) into LDS.  cs: kSelfPlanMax + 1 ints.  Ends with a barrier.
constexpr int kSelfPlanMax = 64;

__device__ __forceinline__ void plan_in_lds(const int64_t* __restrict__ count, int N, int* cs) {
  const int tid = threadIdx.x;
  if (tid < kWave) {
    int v = 0;
    if (tid < N) {
      const int64_t c = count[tid];
      v = c > 0 ? (int)((c + kBinChunk - 1) / kBinChunk) : 0;
    }
    int x = v;
    for (int d = 1; d < kWave; d <<= 1) {
      const int y = __shfl_up(x, d);
      if (tid >= d) x += y;
    }
    cs[tid + 1] = x;
    if (tid == 0) cs[0] = 0;
  }
  __syncthreads();
}

// Number of leading entries b = 0 .. n-1 (n <= 32) for which a predicate that holds on a prefix holds.
template <typename Pred>
__device__ __forceinline__ int prefix_count(int n, Pred holds) {
  int lo = 0;
#pragma unroll
  for (int step = 32; step >= 1; step >>= 1) {
    const int t = lo + step;
    const int i = (t < n ? t : n) - 1;  // (n >= 1: a launch has at least one bin per side)
    if (t <= n && holds(i)) lo = t;
  }
  return lo;
}

// Shared prologue of the count and fill kernels.  chunk_start == nullptr: plan in LDS (cs_l) instead.
template <int KIND>
__device__ __forceinline__ bool chunk_prologue(const float* __restrict__ elems, const float* __restrict__ aux,
                                               const int64_t* __restrict__ first, const int64_t* __restrict__ count,
                                               const int* chunk_start, int* cs_l, int N, int H, int W, int bin_size,
                                               int BH, int BW, float sqrt_blur, float* xlo_t, float* xhi_t,
                                               float* ylo_t, float* yhi_t, ChunkCtx* c, int* publish_plan = nullptr,
                                               int* plan_hdr = nullptr) {
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x;
  if (chunk_start == nullptr) {
    plan_in_lds(count, N, cs_l);
    chunk_start = cs_l;
    // batches of up to kSelfPlanMax elements have no plan kernel (bin_build): the count pass's first workgroup leaves the
    // chunk table for the row scan and clears the tile order's class counters, as bin_plan_kernel would have
    if (publish_plan != nullptr && chunk == 0) {
      if (tid <= N) publish_plan[tid] = cs_l[tid];
      if (tid < 2 * kPlanClasses) plan_hdr[kPlanHdr + tid] = 0;
      if (tid == 0) plan_hdr[kPlanTicket] = 0;
    }
  }
  if (chunk >= chunk_start[N]) return false;
  // largest n with chunk_start[n] <= chunk
  int lo = 0, hi = N;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_start[mid] <= chunk)
      lo = mid;
    else
      hi = mid;
  }
  c->n = lo;
  // Bin edge tables (rasterize_coarse.cu:148-160), monotone in the bin index.
  if (tid < BW) {
    xlo_t[tid] = bin_lo(tid, bin_size, W, H);
    xhi_t[tid] = bin_hi(tid, bin_size, W, H);
  } else if (tid >= 64 && tid < 64 + BH) {
    const int b = tid - 64;
    ylo_t[b] = bin_lo(b, bin_size, H, W);
    yhi_t[b] = bin_hi(b, bin_size, H, W);
  }
  __syncthreads();

  const int64_t local = (int64_t)(chunk - chunk_start[lo]) * kBinChunk + tid;
  const int64_t cnt = count[lo];
  BinRect r;
  r.x0 = r.y0 = 1 << 20;
  r.x1 = r.y1 = -1;
  c->e = -1;
  c->z = 0.0f;
  if (local < cnt) {
    const int64_t e = first[lo] + local;
    c->e = e;
    float xmin, xmax, ymin, ymax;
    bool skip;
    if (KIND == kTriangles) {
      // TriangleBoundingBoxKernel, rasterize_coarse.cu:28-44
      const float* q = elems + e * 9;
      const float v0x = q[0], v0y = q[1], v0z = q[2];
      const float v1x = q[3], v1y = q[4], v1z = q[5];
      const float v2x = q[6], v2y = q[7], v2z = q[8];
      xmin = min3(v0x, v1x, v2x) - sqrt_blur;
      xmax = max3(v0x, v1x, v2x) + sqrt_blur;
      ymin = min3(v0y, v1y, v2y) - sqrt_blur;
      ymax = max3(v0y, v1y, v2y) + sqrt_blur;
      skip = (double)min3(v0z, v1z, v2z) < P3D_KEPS;
    } else {
      // PointBoundingBoxKernel, rasterize_coarse.cu:61-72
      const float* q = elems + e * 3;
      const float rad = aux[e];
      xmin = q[0] - rad;
      xmax = q[0] + rad;
      ymin = q[1] - rad;
      ymax = q[1] + rad;
      skip = q[2] < 0.0f;
      c->z = q[2];
    }
    if (!skip) {
      // overlap(b) = (min <= hi_t[b]) && (lo_t[b] < max); both predicates are monotone in b (the tables ascend),
      // so the overlapping bins are the interval [#(hi_t < min), #(lo_t < max) - 1]: four bisections of <= 32 entries
      // (six steps each; round 4 -- a linear scan of both tables was 400 of the kernels' instructions per wave).
      const int x0 = prefix_count(BW, [&](int b) { return !(xmin <= xhi_t[b]); });
      int x1 = prefix_count(BW, [&](int b) { return xlo_t[b] < xmax; });
      const int y0 = prefix_count(BH, [&](int b) { return !(ymin <= yhi_t[b]); });
      int y1 = prefix_count(BH, [&](int b) { return ylo_t[b] < ymax; });
      x1 -= 1;
      y1 -= 1;
      if (x0 <= x1 && y0 <= y1) {
        r.x0 = x0;
        r.x1 = x1;
        r.y0 = y0;
        r.y1 = y1;
      }
    }
  }
  c->r = r;
  // union rectangle over the wave (butterfly)
  BinRect u = r;
  for (int d = 32; d >= 1; d >>= 1) {
    u.x0 = min(u.x0, __shfl_xor(u.x0, d));
    u.y0 = min(u.y0, __shfl_xor(u.y0, d));
    u.x1 = max(u.x1, __shfl_xor(u.x1, d));
    u.y1 = max(u.y1, __shfl_xor(u.y1, d));
  }
  c->u = u;
  return true;
}

// Which lanes of a wave touch bin row `by` / bin column `bx`: one ballot per row and per column of the wave's union
// rectangle, kept in LDS (rm, cm: the wave's own 32 + 32 words).  The members of bin (by, bx) are rm[by] & cm[bx]: the
// per-bin work needs no ballot of its own, so the LANES can take a bin each (for_union_bins) where until round 4 the whole
// wave walked the union bin by bin -- ~100 bins x 12 instructions per pass for the 64 consecutive faces of a torus strip,
// whose own rectangles hold ~9 bins each: 3300 instructions per wave, three quarters of them in those walks.
__device__ __forceinline__ void wave_masks(const ChunkCtx& c, int lane, unsigned long long* rm, unsigned long long* cm) {
  for (int by = c.u.y0; by <= c.u.y1; ++by) {
    const unsigned long long m = __ballot(by >= c.r.y0 && by <= c.r.y1);
    if (lane == 0) rm[by] = m;
  }
  for (int bx = c.u.x0; bx <= c.u.x1; ++bx) {
    const unsigned long long m = __ballot(bx >= c.r.x0 && bx <= c.r.x1);
    if (lane == 0) cm[bx] = m;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // same wave, in-order LDS: ordering for the compiler only
}

// f(by, bx) for every bin of the union rectangle, a lane per bin, 8 x 8 bins per step
template <typename F>
__device__ __forceinline__ void for_union_bins(const BinRect& u, int lane, F f) {
  const int ly = lane >> 3, lx = lane & 7;
  for (int ty = u.y0; ty <= u.y1; ty += 8) {
    for (int tx = u.x0; tx <= u.x1; tx += 8) {
      const int by = ty + ly, bx = tx + lx;
      if (by <= u.y1 && bx <= u.x1) f(by, bx);
    }
  }
}

// ---------------------------------------------------------------------------------------
// count: counts[chunk][bin] = members of `bin` among the chunk's 1024 primitives.
// ---------------------------------------------------------------------------------------
template <int KIND, bool ORDERED>
__global__ __launch_bounds__(kBinChunk) void bin_count_kernel(const float* __restrict__ elems,
                                                              const float* __restrict__ aux,
                                                              const int64_t* __restrict__ first,
                                                              const int64_t* __restrict__ count,
                                                              const int* chunk_start, int N, int H, int W,
                                                              int bin_size, int BH, int BW, float sqrt_blur,
                                                              int* __restrict__ counts, int* publish_plan, int* plan_hdr) {
  __shared__ float xlo_t[32], xhi_t[32], ylo_t[32], yhi_t[32];
  __shared__ int cs_l[kSelfPlanMax + 1];
  __shared__ int blk_cnt[kMaxBins];
  __shared__ unsigned long long rowm[ORDERED ? kWavesPerChunk : 1][kMaxBinsSide], colm[ORDERED ? kWavesPerChunk : 1][kMaxBinsSide];
  const int nbins = BH * BW;
  for (int b = threadIdx.x; b < nbins; b += kBinChunk) blk_cnt[b] = 0;
  ChunkCtx c;
  if (!chunk_prologue<KIND>(elems, aux, first, count, chunk_start, cs_l, N, H, W, bin_size, BH, BW, sqrt_blur, xlo_t,
                            xhi_t, ylo_t, yhi_t, &c, publish_plan, plan_hdr))
    return;
  const int lane = lane_id();
  if (ORDERED) {
    const int w = threadIdx.x / kWave;
    const unsigned long long* rm = rowm[ORDERED ? w : 0];
    const unsigned long long* cm = colm[ORDERED ? w : 0];
    wave_masks(c, lane, rowm[ORDERED ? w : 0], colm[ORDERED ? w : 0]);
    for_union_bins(c.u, lane, [&](int by, int bx) {
      const int members = __popcll(rm[by] & cm[bx]);
      if (members) atomicAdd(&blk_cnt[by * BW + bx], members);
    });
  } else {
    // primitives in arbitrary spatial order (point clouds): the wave's union rectangle is the whole
    // image, while each primitive touches a handful of bins -- one integer LDS atomic per (primitive, bin)
    for (int by = c.r.y0; by <= c.r.y1; ++by)
      for (int bx = c.r.x0; bx <= c.r.x1; ++bx) atomicAdd(&blk_cnt[by * BW + bx], 1);
  }
  __syncthreads();
  int* out = counts + (int64_t)blockIdx.x * nbins;
  for (int b = threadIdx.x; b < nbins; b += kBinChunk) out[b] = blk_cnt[b];
}

// ---------------------------------------------------------------------------------------
// row scan: for each (n, bin) an exclusive scan of counts over n's chunks, in place.  total[row] = min(sum, M).
// Two shapes, chosen by bin_build from the chunks per batch element:
//   few chunks, many rows (a batch of meshes: 6 chunks per image and 65 536 rows at the bench workload): one THREAD per
//     row -- neighbouring threads are neighbouring bins of one batch element, so every step reads and writes one
//     contiguous piece of a chunk's counts per wave, and the loads of eight chunks are issued together (a wave per row
//     kept 6 of its 64 lanes busy and gathered them at a stride of a whole counts row: 0.019 -> 0.007 ms) -- since round 4
//     in one launch with the block sums of the offsets scan (bin_scan_rows_sums_kernel, further down);
//   many chunks, few rows (a cloud of 1M points: 977 chunks, 1024 rows): one WAVE per row, 64 chunks per step (the
//     thread-per-row form walks the 977 chunks one after the other: 0.075 ms instead of 0.019).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bin_scan_rows_kernel(int* __restrict__ counts,
                                                            const int* __restrict__ chunk_start, int N, int nbins, int M,
                                                            int* __restrict__ total) {
  const int lane = lane_id();
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
  if (row >= (int64_t)N * nbins) return;
  const int n = (int)(row / nbins);
  const int b = (int)(row % nbins);
  const int c0 = chunk_start[n];
  const int nch = chunk_start[n + 1] - c0;
  int carry = 0;
  // four groups of 64 chunks per step: their loads are in flight together (a step per group waited out one memory round trip
  // per 64 chunks, 16 in a row for a cloud of 1M points)
  constexpr int U = 4;
  for (int base = 0; base < nch; base += U * kWave) {
    int v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * kWave + lane;
      v[u] = i < nch ? counts[((int64_t)(c0 + i)) * nbins + b] : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * kWave + lane;
      int x = v[u];
      for (int d = 1; d < kWave; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
      }
      if (i < nch) counts[((int64_t)(c0 + i)) * nbins + b] = carry + x - v[u];
      carry += __shfl(x, kWave - 1);
    }
  }
  if (lane == 0) total[row] = carry < M ? carry : M;
}

// ---------------------------------------------------------------------------------------
// offsets: exclusive scan of total[] -> offset[] (int64) in two small launches:
//   A  one workgroup per 1024 rows: block sum -> blocksum[b]
//   B  the same grid: base = sum of the preceding block sums (<= rows/1024 coalesced adds per
//      thread), then a local exclusive scan of the block's 1024 rows.
// The scans carry the tile plan (binning.h: TilePlan) along for free: every row contributes total + (active << 40), so
// the low 40 bits of the running sum are the list offset (list capacities are far below 2^40 entries) and the bits
// above count the active rows before it.
// ---------------------------------------------------------------------------------------
constexpr int kPlanShift = 40;
constexpr long long kPlanMask = (1ll << kPlanShift) - 1;
__device__ __forceinline__ long long plan_pack(int total) { return (long long)total + (total > 0 ? (1ll << kPlanShift) : 0ll); }
// Tile order.  Workgroups reach the CUs round robin by index, so the ORDER of the fine kernel's work items is its load
// balance (profiles/r03/bwd_timeline.txt).  The active rows are listed by descending list length -- a counting sort over
// kPlanClasses classes of 8 primitives, folded into the two kernels of the offsets scan: the block sums build the class
// histogram (LDS, then one global atomic per class and block), the offsets scan reserves each block's range per class and
// scatters its rows.  Dealt longest-first every CU draws the same mix, and the launch has no tail of long tiles.
__device__ __forceinline__ int plan_class(int v) { return kPlanClasses - 1 - min(v >> 3, kPlanClasses - 1); }  // 0 = longest

// row `i` with running exclusive sum `ex` (packed) and own count `v`: offset, active rank, background list entry
__device__ __forceinline__ void plan_emit(int64_t i, long long ex, int v, int64_t* offset, int* arank, int* bg_list) {
  offset[i] = ex & kPlanMask;
  const int r = (int)(ex >> kPlanShift);
  if (v <= 0) bg_list[i - r] = (int)i;
  arank[i] = r;
}

__device__ __forceinline__ long long block_exclusive_scan_1024(long long v, long long* wsum, long long* block_total) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  long long x = v;
  for (int d = 1; d < kWave; d <<= 1) {
    const long long y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[w] = x;
  __syncthreads();
  long long base = 0, all = 0;
  for (int j = 0; j < 16; ++j) {
    const long long c = wsum[j];
    if (j < w) base += c;
    all += c;
  }
  *block_total = all;
  return base + x - v;
}

__global__ __launch_bounds__(1024) void bin_block_sums_kernel(const int* __restrict__ total, int64_t rows,
                                                              long long* __restrict__ blocksum, int* __restrict__ plan_hdr) {
  __shared__ long long wsum[16];
  const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  long long all;
  const int v = i < rows ? total[i] : 0;
  block_exclusive_scan_1024(i < rows ? plan_pack(v) : 0, wsum, &all);
  if (threadIdx.x == 0) blocksum[blockIdx.x] = all;
  if (plan_hdr) {  // uniform
    __shared__ int hist[kPlanClasses];
    if (threadIdx.x < kPlanClasses) hist[threadIdx.x] = 0;
    __syncthreads();
    if (v > 0) atomicAdd(&hist[plan_class(v)], 1);
    __syncthreads();
    if (threadIdx.x < kPlanClasses && hist[threadIdx.x] > 0) atomicAdd(&plan_hdr[kPlanHdr + threadIdx.x], hist[threadIdx.x]);
  }
}

// The thread-per-row scan and the block sums of the offsets scan in one launch (workgroups of 1024 rows, the blocks of the
// offsets scan): a row's total goes from the register into the block's sum and the tile order's class histogram.
__global__ __launch_bounds__(1024) void bin_scan_rows_sums_kernel(int* __restrict__ counts, const int* __restrict__ chunk_start,
                                                                  int N, int nbins, int M, int* __restrict__ total,
                                                                  long long* __restrict__ blocksum, int* __restrict__ plan_hdr) {
  __shared__ long long wsum[16];
  __shared__ int hist[kPlanClasses];
  const int64_t rows = (int64_t)N * nbins;
  const int64_t row = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  if (threadIdx.x < kPlanClasses) hist[threadIdx.x] = 0;
  int t = 0;
  if (row < rows) {
    const int n = (int)(row / nbins);
    const int b = (int)(row % nbins);
    const int c0 = chunk_start[n];
    const int nch = chunk_start[n + 1] - c0;
    int* p = counts + (int64_t)c0 * nbins + b;
    int carry = 0;
    for (int base = 0; base < nch; base += 8) {
      int v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = base + j < nch ? p[(int64_t)(base + j) * nbins] : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (base + j < nch) p[(int64_t)(base + j) * nbins] = carry;
        carry += v[j];
      }
    }
    t = carry < M ? carry : M;
    total[row] = t;
  }
  long long all;
  block_exclusive_scan_1024(row < rows ? plan_pack(t) : 0, wsum, &all);  // (its barrier also orders the histogram's zero fill)
  if (threadIdx.x == 0) blocksum[blockIdx.x] = all;
  if (t > 0) atomicAdd(&hist[plan_class(t)], 1);
  __syncthreads();
  if (threadIdx.x < kPlanClasses && hist[threadIdx.x] > 0) atomicAdd(&plan_hdr[kPlanHdr + threadIdx.x], hist[threadIdx.x]);
}

// One block of the offsets scan (1024 threads).  COHERENT: `total` was written by other workgroups of THIS launch
// (bin_scan_rows_tail_kernel): read it past the L1.
template <bool COHERENT>
__device__ __forceinline__ void scan_offsets_block(const int* __restrict__ total, int64_t rows, const long long* __restrict__ blocksum,
                                                   int64_t* __restrict__ offset, int* __restrict__ arank, int* __restrict__ bg_list,
                                                   int* __restrict__ plan_hdr, int* __restrict__ order, int64_t capacity, int block,
                                                   int nblocks) {
  __shared__ long long wsum[16];
  __shared__ long long part[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // base = sum of blocksum[0 .. block)
  long long acc = 0;
  for (int j = tid; j < block; j += 1024) acc += blocksum[j];
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if (lane == 0) part[w] = acc;
  __syncthreads();
  long long base = 0;
  for (int j = 0; j < 16; ++j) base += part[j];
  const int64_t i = (int64_t)block * 1024 + tid;
  const int v = i < rows ? (COHERENT ? __hip_atomic_load(total + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : total[i]) : 0;
  long long all;
  const long long ex = block_exclusive_scan_1024(i < rows ? plan_pack(v) : 0, wsum, &all);
  if (i < rows) {
    if (arank)
      plan_emit(i, base + ex, v, offset, arank, bg_list);
    else
      offset[i] = (base + ex) & kPlanMask;
  }
  if (i == rows - 1) {
    const long long end = base + ex + plan_pack(v);
    offset[rows] = end & kPlanMask;
    if (plan_hdr) {
      plan_hdr[0] = (int)(end >> kPlanShift);
      plan_hdr[1] = (int)(rows - (end >> kPlanShift));
      plan_hdr[2] = (end & kPlanMask) > capacity ? 1 : 0;  // the lists do not fit the workspace (binning.h: short workspaces)
      plan_hdr[3] = order != nullptr ? 1 : 0;
    }
  }
  if (order != nullptr) {  // uniform: this block's active rows into their classes' ranges (see plan_class)
    __shared__ int hist[kPlanClasses], start[kPlanClasses];
    if (tid < kPlanClasses) hist[tid] = 0;
    __syncthreads();
    const int cls = v > 0 ? plan_class(v) : -1;
    if (cls >= 0) atomicAdd(&hist[cls], 1);
    __syncthreads();
    // a single block (<= 1024 rows: one image of points) is its own class histogram: no block-sums launch before this one
    const bool single = nblocks == 1;
    int before = 0, mine = 0;
    if (tid < kPlanClasses) {
      mine = hist[tid];
      for (int c = 0; c < tid; ++c) before += single ? hist[c] : plan_hdr[kPlanHdr + c];  // (final: written by the kernel before this one)
    }
    __syncthreads();
    if (tid < kPlanClasses) {
      start[tid] = before + ((!single && mine > 0) ? atomicAdd(&plan_hdr[kPlanHdr + kPlanClasses + tid], mine) : 0);
      hist[tid] = 0;
    }
    __syncthreads();
    if (cls >= 0) order[start[cls] + atomicAdd(&hist[cls], 1)] = (int)i;
  }
}

__global__ __launch_bounds__(1024) void bin_scan_offsets_kernel(const int* __restrict__ total, int64_t rows,
                                                                const long long* __restrict__ blocksum,
                                                                int64_t* __restrict__ offset, int* __restrict__ arank,
                                                                int* __restrict__ bg_list, int* __restrict__ plan_hdr,
                                                                int* __restrict__ order, int64_t capacity) {
  scan_offsets_block<false>(total, rows, blocksum, offset, arank, bg_list, plan_hdr, order, capacity, (int)blockIdx.x, (int)gridDim.x);
}

// The row scan of a single image with many chunks (one cloud of 1M points: 977 chunks, 1024 rows) and the offsets scan behind it as ONE
// launch (round 6): a wave per row as in bin_scan_rows_kernel, four rows per workgroup; a workgroup that has written its totals takes a
// ticket, and the one that draws the last ticket runs the one-block offsets scan (tile plan and tile order included) on the spot -- one
// launch and one launch boundary less on the chain of a small image.  rows <= 1024.
constexpr int kTailRows = 4;
__global__ __launch_bounds__(1024) void bin_scan_rows_tail_kernel(int* __restrict__ counts, const int* __restrict__ chunk_start, int N,
                                                                  int nbins, int M, int* __restrict__ total, int64_t* __restrict__ offset,
                                                                  int* __restrict__ arank, int* __restrict__ bg_list,
                                                                  int* __restrict__ plan_hdr, int* __restrict__ order, int64_t capacity) {
  __shared__ int s_last;
  const int lane = lane_id();
  const int64_t rows = (int64_t)N * nbins;
  // four rows per workgroup, as bin_scan_rows_kernel: the scan is bound by the strided loads of its waves, and sixteen scanning waves on
  // one CU share one texture path (measured: 0.086 ms with sixteen against 0.019); the other twelve waves of the workgroup only take
  // part in the tail
  const int wave = threadIdx.x / kWave;
  const int64_t row = (int64_t)blockIdx.x * kTailRows + wave;
  if (wave < kTailRows && row < rows) {  // wave-uniform
    const int n = (int)(row / nbins);
    const int b = (int)(row % nbins);
    const int c0 = chunk_start[n];
    const int nch = chunk_start[n + 1] - c0;
    int carry = 0;
    constexpr int U = 4;
    for (int base = 0; base < nch; base += U * kWave) {
      int v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * kWave + lane;
        v[u] = i < nch ? counts[((int64_t)(c0 + i)) * nbins + b] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * kWave + lane;
        int x = v[u];
        for (int d = 1; d < kWave; d <<= 1) {
          const int y = __shfl_up(x, d);
          if (lane >= d) x += y;
        }
        if (i < nch) counts[((int64_t)(c0 + i)) * nbins + b] = carry + x - v[u];
        carry += __shfl(x, kWave - 1);
      }
    }
    if (lane == 0) {
      // The total must be visible to the workgroup that draws the last ticket, which may run on another XCD (its own L2).  A release
      // fence would do it -- and write back EVERYTHING this L2 holds dirty, the 4 MB of counts included, once per workgroup (measured:
      // the launch at 0.071 ms instead of 0.019 + 0.010 for the two it replaces).  A device-scope read-modify-write is performed at the
      // point of coherence, and its returned value says it has been: exchange, wait for the return, then the ticket (relaxed, also there).
      const int old = __hip_atomic_exchange(total + row, carry < M ? carry : M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(old) : "memory");
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(plan_hdr + kPlanTicket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;  // uniform
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (one workgroup, once: an invalidate, no write-back)
  scan_offsets_block<true>(total, rows, nullptr, offset, arank, bg_list, plan_hdr, order, capacity, 0, 1);
}

// ---------------------------------------------------------------------------------------
// Small launches (bin_build: N <= 64, <= 4096 rows, <= 128 chunks -- a single image or a small batch, where launch
// latency is the cost): the row scan, the block sums and the offsets scan in ONE workgroup, planning in LDS.  A thread
// per row walks its batch element's few chunks, then the 1024 rows of the step are scanned together.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bin_scan_small_kernel(int* __restrict__ counts, const int64_t* __restrict__ count,
                                                              int N, int nbins, int M, int* __restrict__ total,
                                                              int64_t* __restrict__ offset, int* __restrict__ arank,
                                                              int* __restrict__ bg_list, int* __restrict__ plan_hdr,
                                                              int64_t capacity) {
  __shared__ int cs[kSelfPlanMax + 1];
  if (threadIdx.x == 0) plan_hdr[3] = 0;  // no sorted order from this kernel (small launches run the split kernels)
  __shared__ long long wsum[16];
  plan_in_lds(count, N, cs);
  const int tid = threadIdx.x;
  const int64_t rows = (int64_t)N * nbins;
  long long carry = 0;
  for (int64_t base = 0; base < rows; base += 1024) {
    const int64_t row = base + tid;
    int t = 0;
    if (row < rows) {
      const int n = (int)(row / nbins), b = (int)(row % nbins);
      const int c0 = cs[n], nch = cs[n + 1] - c0;
      int run = 0;
      for (int i = 0; i < nch; ++i) {
        int* p = counts + (int64_t)(c0 + i) * nbins + b;
        const int v = *p;
        *p = run;
        run += v;
      }
      t = run < M ? run : M;
      total[row] = t;
    }
    long long all;
    const long long ex = block_exclusive_scan_1024(row < rows ? plan_pack(t) : 0, wsum, &all);
    if (row < rows) plan_emit(row, carry + ex, t, offset, arank, bg_list);
    carry += all;
    __syncthreads();  // wsum is rewritten by the next step
  }
  if (tid == 0) {
    offset[rows] = carry & kPlanMask;
    plan_hdr[0] = (int)(carry >> kPlanShift);
    plan_hdr[1] = (int)(rows - (carry >> kPlanShift));
    plan_hdr[2] = (carry & kPlanMask) > capacity ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------
// fill: re-derive the ballots and write ids at offset + row prefix + wave prefix + lane rank.
// ---------------------------------------------------------------------------------------
template <int KIND, bool ORDERED>
__global__ __launch_bounds__(kBinChunk) void bin_fill_kernel(const float* __restrict__ elems,
                                                             const float* __restrict__ aux,
                                                             const int64_t* __restrict__ first,
                                                             const int64_t* __restrict__ count,
                                                             const int* chunk_start, int N, int H, int W,
                                                             int bin_size, int BH, int BW, float sqrt_blur, int M,
                                                             const int* __restrict__ counts,
                                                             const int64_t* __restrict__ offset,
                                                             int* __restrict__ list, const int* __restrict__ plan_hdr,
                                                             int stride) {
  if (plan_hdr[2] != 0) return;  // uniform: the lists do not fit `list` (a short workspace); the caller's fallback runs instead
  __shared__ float xlo_t[32], xhi_t[32], ylo_t[32], yhi_t[32];
  __shared__ int cs_l[kSelfPlanMax + 1];
  __shared__ unsigned short wpre[ORDERED ? kWavesPerChunk : 1][kMaxBins];  // per-wave counts, then prefixes (<= 1024)
  __shared__ int base_t[kMaxBins];                                         // this chunk's row prefix per bin
  // where this chunk's entries of a bin start in the list (row offset + row prefix): one coalesced read per workgroup.
  // Looked up per (primitive, bin) in the placement loop it was a dependent global round trip in every iteration.
  __shared__ int64_t dst_t[kMaxBins];
  __shared__ unsigned long long rowm[ORDERED ? kWavesPerChunk : 1][kMaxBinsSide], colm[ORDERED ? kWavesPerChunk : 1][kMaxBinsSide];
  const int nbins = BH * BW;
  if (ORDERED)
    for (int i = threadIdx.x; i < kWavesPerChunk * kMaxBins / 2; i += kBinChunk)
      reinterpret_cast<unsigned*>(&wpre[0][0])[i] = 0u;
  ChunkCtx c;
  if (!chunk_prologue<KIND>(elems, aux, first, count, chunk_start, cs_l, N, H, W, bin_size, BH, BW, sqrt_blur, xlo_t,
                            xhi_t, ylo_t, yhi_t, &c))
    return;
  const int lane = lane_id();
  const int w = threadIdx.x / kWave;
  if (!ORDERED) {
    // unordered placement: slot = chunk's row prefix + arrival order inside the chunk (integer LDS
    // atomic with return).  The list order inside a chunk is then arbitrary -- only for consumers
    // whose result does not depend on it (point rasterization: top-K under a total order).
    int* pos_t = base_t;
    const int64_t row0u = (int64_t)c.n * nbins;
    for (int b = threadIdx.x; b < nbins; b += kBinChunk) {
      pos_t[b] = counts[(int64_t)blockIdx.x * nbins + b];
      dst_t[b] = offset[row0u + b];
    }
    __syncthreads();
    for (int by = c.r.y0; by <= c.r.y1; ++by) {
      for (int bx = c.r.x0; bx <= c.r.x1; ++bx) {
        const int b = by * BW + bx;
        const int pos = atomicAdd(&pos_t[b], 1);
        if (pos < M) {
          if (KIND == kPoints && stride == 2)
            reinterpret_cast<int2*>(list)[dst_t[b] + pos] = make_int2((int)c.e, __float_as_int(c.z));  // one request, as the id alone
          else
            list[dst_t[b] + pos] = (int)c.e;
        }
      }
    }
    return;
  }
  // pass A: per-wave member counts (the prologue's barrier ordered the zero fill), a lane per bin of the wave's union
  const unsigned long long* rm = rowm[ORDERED ? w : 0];
  const unsigned long long* cm = colm[ORDERED ? w : 0];
  wave_masks(c, lane, rowm[ORDERED ? w : 0], colm[ORDERED ? w : 0]);
  for_union_bins(c.u, lane, [&](int by, int bx) { wpre[w][by * BW + bx] = (unsigned short)__popcll(rm[by] & cm[bx]); });
  __syncthreads();
  // pass B: exclusive prefix over the 16 waves, seeded with this chunk's row prefix
  const int64_t row0 = (int64_t)c.n * nbins;
  for (int b = threadIdx.x; b < nbins; b += kBinChunk) {
    base_t[b] = counts[(int64_t)blockIdx.x * nbins + b];
    dst_t[b] = offset[row0 + b];
    int run = 0;
    for (int j = 0; j < kWavesPerChunk; ++j) {
      const int v = wpre[ORDERED ? j : 0][b];
      wpre[ORDERED ? j : 0][b] = (unsigned short)run;
      run += v;
    }
  }
  __syncthreads();
  // pass C: place -- every lane walks its OWN rectangle; its rank in a bin is the number of lower lanes among the members
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int by = c.r.y0; by <= c.r.y1; ++by) {
    const unsigned long long mr = rm[by] & below;
    for (int bx = c.r.x0; bx <= c.r.x1; ++bx) {
      const int b = by * BW + bx;
      const int pos = base_t[b] + wpre[ORDERED ? w : 0][b] + __popcll(mr & cm[bx]);
      if (pos < M) {
        if (KIND == kPoints && stride == 2)
          reinterpret_cast<int2*>(list)[dst_t[b] + pos] = make_int2((int)c.e, __float_as_int(c.z));
        else
          list[dst_t[b] + pos] = (int)c.e;
      }
    }
  }
}

__global__ void bin_expand_kernel(const int64_t* __restrict__ offset, const int* __restrict__ total,
                                  const int* __restrict__ list, int64_t rows, int M, int32_t* __restrict__ out) {
  const int64_t n = rows * M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / M;
    const int m = (int)(i - row * M);
    out[i] = m < total[row] ? list[offset[row] + m] : -1;
  }
}

// padded row -> compacted (order kept) at row*M; one wave per row.
__global__ __launch_bounds__(256) void bin_compact_kernel(const int32_t* __restrict__ padded, int64_t rows, int M,
                                                          int* __restrict__ list, int* __restrict__ total,
                                                          int64_t* __restrict__ offset) {
  const int lane = lane_id();
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
  if (row >= rows) return;
  const int32_t* src = padded + row * M;
  int* dst = list + row * M;
  int carry = 0;
  for (int base = 0; base < M; base += kWave) {
    const int i = base + lane;
    const int v = i < M ? src[i] : -1;
    const unsigned long long m = __ballot(v >= 0);
    if (v >= 0) dst[carry + mask_rank(m)] = v;
    carry += __popcll(m);
  }
  if (lane == 0) {
    total[row] = carry;
    offset[row] = row * M;
  }
}

}  // namespace

int64_t bin_capacity(int64_t E, int N, const BinGeom& g, int M) {
  // Every bin holds at most M ids (the reference's own cap) and at most E_n ids.
  const int64_t a = E * (int64_t)g.nbins;
  const int64_t b = (int64_t)N * g.nbins * (int64_t)M;
  int64_t c = a < b ? a : b;
  if (c < 1) c = 1;
  return c;
}

bool bin_carve(Arena& arena, int64_t E, int N, const BinGeom& g, int M, BinWorkspace* ws, int64_t list_entries, bool with_z) {
  ws->max_chunks = ceil_div(E, kBinChunk) + N;
  ws->worst = bin_capacity(E, N, g, M);
  ws->chunk_start = arena.take<int>((size_t)N + 1);
  ws->counts = arena.take<int>((size_t)ws->max_chunks * g.nbins);
  ws->total = arena.take<int>((size_t)N * g.nbins);
  ws->need_at = arena.off + (size_t)N * g.nbins * sizeof(int64_t);  // offset[rows]: the list total the build needed
  ws->offset = arena.take<int64_t>((size_t)N * g.nbins + 1);
  ws->blocksum = arena.take<long long>((size_t)ceil_div((int64_t)N * g.nbins, 1024) + 1);
  ws->arank = arena.take<int>((size_t)N * g.nbins);
  ws->bg_list = arena.take<int>((size_t)N * g.nbins);
  ws->plan_hdr = arena.take<int>(kPlanHdrInts);
  ws->order = arena.take<int>((size_t)N * g.nbins);
  // the list comes last: a caller that allows short workspaces (list_entries >= 0) gets whatever is left of the arena, at
  // least list_entries ids and never more than the worst case
  int64_t want = ws->worst;
  if (list_entries >= 0) {
    want = list_entries < 1 ? 1 : (list_entries < ws->worst ? list_entries : ws->worst);
    if (arena.base != nullptr && arena.cap > arena.off) {
      const int64_t room = (int64_t)((arena.cap - arena.off) / sizeof(int)) / (with_z ? 2 : 1);
      if (room > want) want = room < ws->worst ? room : ws->worst;
    }
  }
  ws->capacity = want;
  ws->stride = with_z ? 2 : 1;
  ws->list = arena.take<int>((size_t)want * (size_t)ws->stride);
  return arena.ok();
}

size_t bin_workspace_bytes(int64_t E, int N, const BinGeom& g, int M, int64_t list_entries, bool with_z) {
  Arena probe(nullptr, 0);
  BinWorkspace ws;
  bin_carve(probe, E, N, g, M, &ws, list_entries, with_z);
  return probe.off;
}

int bin_build(BinKind kind, const float* elems, const float* aux, const int64_t* first, const int64_t* count, int64_t E,
              int N, const BinGeom& g, int M, float sqrt_blur, const BinWorkspace& ws, hipStream_t stream, bool ordered) {
  if (N <= 0) return P3D_OK;
  const int64_t rows = (int64_t)N * g.nbins;
  // small launches: three kernels instead of six (no plan kernel, one scan kernel) -- see bin_scan_small_kernel
  const bool small = N <= kSelfPlanMax && rows <= 4096 && ws.max_chunks <= 128;
  // up to kSelfPlanMax batch elements every count / fill workgroup derives the chunk table itself (plan_in_lds) and the
  // count pass's first workgroup publishes it for the row scan: no plan launch (round 4; the bench batch has 64 elements)
  const bool self_plan = N <= kSelfPlanMax;
  const int* cs = self_plan ? nullptr : ws.chunk_start;
  int* publish = self_plan && !small ? ws.chunk_start : nullptr;
  if (!self_plan) {
    LaunchScope ls("bin_plan", stream);
    bin_plan_kernel<<<1, 1024, 0, stream>>>(count, N, ws.chunk_start, ws.plan_hdr);
  }
  const unsigned chunks = (unsigned)ws.max_chunks;
  {
    LaunchScope ls("bin_count", stream);
    if (kind == kTriangles)
      bin_count_kernel<kTriangles, true><<<chunks, kBinChunk, 0, stream>>>(elems, aux, first, count, cs, N, g.H, g.W,
                                                                          g.bin_size, g.BH, g.BW, sqrt_blur, ws.counts, publish, ws.plan_hdr);
    else if (ordered)
      bin_count_kernel<kPoints, true><<<chunks, kBinChunk, 0, stream>>>(elems, aux, first, count, cs, N, g.H, g.W,
                                                                       g.bin_size, g.BH, g.BW, sqrt_blur, ws.counts, publish, ws.plan_hdr);
    else
      bin_count_kernel<kPoints, false><<<chunks, kBinChunk, 0, stream>>>(elems, aux, first, count, cs, N, g.H, g.W,
                                                                        g.bin_size, g.BH, g.BW, sqrt_blur, ws.counts, publish, ws.plan_hdr);
  }
  if (small) {
    LaunchScope ls("bin_scan_small", stream);
    bin_scan_small_kernel<<<1, 1024, 0, stream>>>(ws.counts, count, N, g.nbins, M, ws.total, ws.offset, ws.arank, ws.bg_list,
                                                  ws.plan_hdr, ws.capacity);
  } else {
    const unsigned nb = (unsigned)ceil_div(rows, 1024);
    // chunks per batch element (max_chunks = ceil(E / chunk) + N bounds their sum): see the kernels
    const bool thread_rows = ws.max_chunks <= 32 * (int64_t)N;
    {
      LaunchScope ls("bin_scan_rows", stream);
      if (thread_rows)
        bin_scan_rows_sums_kernel<<<nb, 1024, 0, stream>>>(ws.counts, ws.chunk_start, N, g.nbins, M, ws.total, ws.blocksum,
                                                           ws.plan_hdr);
      else if (nb == 1)  // one image: the offsets scan is the tail of the row scan (the last workgroup to arrive runs it)
        bin_scan_rows_tail_kernel<<<(unsigned)ceil_div(rows, kTailRows), 1024, 0, stream>>>(
            ws.counts, ws.chunk_start, N, g.nbins, M, ws.total, ws.offset, ws.arank, ws.bg_list, ws.plan_hdr, ws.order, ws.capacity);
      else
        bin_scan_rows_kernel<<<(unsigned)ceil_div(rows, 4), 256, 0, stream>>>(ws.counts, ws.chunk_start, N, g.nbins, M,
                                                                             ws.total);
    }
    if (thread_rows || nb > 1) {
      LaunchScope ls("bin_scan_offsets", stream);
      if (!thread_rows && nb > 1) bin_block_sums_kernel<<<nb, 1024, 0, stream>>>(ws.total, rows, ws.blocksum, ws.plan_hdr);
      bin_scan_offsets_kernel<<<nb, 1024, 0, stream>>>(ws.total, rows, ws.blocksum, ws.offset, ws.arank, ws.bg_list, ws.plan_hdr,
                                                       ws.order, ws.capacity);
    }
  }
  {
    LaunchScope ls("bin_fill", stream);
    if (kind == kTriangles)
      bin_fill_kernel<kTriangles, true><<<chunks, kBinChunk, 0, stream>>>(elems, aux, first, count, cs, N, g.H, g.W,
                                                                         g.bin_size, g.BH, g.BW, sqrt_blur, M, ws.counts,
                                                                         ws.offset, ws.list, ws.plan_hdr, ws.stride);
    else if (ordered)
      bin_fill_kernel<kPoints, true><<<chunks, kBinChunk, 0, stream>>>(elems, aux, first, count, cs, N, g.H, g.W,
                                                                      g.bin_size, g.BH, g.BW, sqrt_blur, M, ws.counts,
                                                                      ws.offset, ws.list, ws.plan_hdr, ws.stride);
    else
      bin_fill_kernel<kPoints, false><<<chunks, kBinChunk, 0, stream>>>(elems, aux, first, count, cs, N, g.H, g.W,
                                                                       g.bin_size, g.BH, g.BW, sqrt_blur, M, ws.counts,
                                                                       ws.offset, ws.list, ws.plan_hdr, ws.stride);
  }
  return launch_status();
}

int exclusive_scan_i32(const int* in, int64_t n, long long* blocksum, int64_t* out, hipStream_t stream) {
  if (n <= 0) return P3D_OK;
  const unsigned nb = (unsigned)ceil_div(n, 1024);
  if (nb > 1) bin_block_sums_kernel<<<nb, 1024, 0, stream>>>(in, n, blocksum, nullptr);  // (one block reads no block sums)
  bin_scan_offsets_kernel<<<nb, 1024, 0, stream>>>(in, n, blocksum, out, nullptr, nullptr, nullptr, nullptr, 0);
  return launch_status();
}

int bin_expand_padded(const BinWorkspace& ws, int N, const BinGeom& g, int M, int32_t* out, hipStream_t stream) {
  const int64_t rows = (int64_t)N * g.nbins;
  const int64_t n = rows * M;
  if (n <= 0) return P3D_OK;
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 8192) blocks = 8192;
  LaunchScope ls("bin_expand", stream);
  bin_expand_kernel<<<(unsigned)blocks, 256, 0, stream>>>(ws.offset, ws.total, ws.list, rows, M, out);
  return launch_status();
}

int bin_compact_padded(const int32_t* padded, int64_t rows, int M, int* ws_list, int* ws_total, int64_t* ws_offset,
                       hipStream_t stream) {
  if (rows <= 0) return P3D_OK;
  LaunchScope ls("bin_compact", stream);
  bin_compact_kernel<<<(unsigned)ceil_div(rows, 4), 256, 0, stream>>>(padded, rows, M, ws_list, ws_total, ws_offset);
  return launch_status();
}

}  // namespace p3d
