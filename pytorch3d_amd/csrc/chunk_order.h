// chunk_order.h -- front-to-back visiting order for one staged chunk (<= 256 primitives) in LDS.
//
// The K nearest under a total order do not depend on the order candidates are offered in, so the
// fine rasterizers visit a chunk by ascending depth key: once a pixel's queue is full, everything
// behind its K-th entry is discarded by one compare, and a wave stops scanning the chunk when the
// nearest remaining primitive is too deep for all of its pixels.  An exact sort is not needed:
// primitives are dealt into 64 depth buckets (linear between the chunk's min and max key) with
// integer LDS atomics (ds_add_rtn_u32 / ds_min_i32 -- fast on gfx950, unlike float atomics), the
// buckets are laid out in order, and `qlow[i]` = the smallest key of the bucket that position i
// belongs to is a lower bound for every key at positions >= i.  Order inside a bucket is arrival
// order (not deterministic); results do not depend on it.
#pragma once

#include "p3d_common.h"

namespace p3d {

struct ChunkOrderScratch {
  int hist[64];
  int start[64];
  int bmin[64];
  float red[8];
};

// key[0..n): depth keys, each >= 0 or -inf.  Writes order[0..n) (a permutation of 0..n-1) and
// qlow[0..n).  All 256 threads of the workgroup must call; ends with a barrier.
__device__ __forceinline__ void chunk_bucket_order(const float* key, int n, int* order, float* qlow,
                                                   ChunkOrderScratch& s, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  const bool have = tid < n;
  const float k = have ? key[tid] : 0.0f;
  const bool finite = have && k > -INFINITY;
  float lo = finite ? k : INFINITY, hi = finite ? k : -INFINITY;
  for (int d = 32; d >= 1; d >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, d));
    hi = fmaxf(hi, __shfl_xor(hi, d));
  }
  if (lane == 0) {
    s.red[w] = lo;
    s.red[4 + w] = hi;
  }
  if (tid < 64) {
    s.hist[tid] = 0;
    s.bmin[tid] = 0x7fffffff;
  }
  __syncthreads();
  lo = fminf(fminf(s.red[0], s.red[1]), fminf(s.red[2], s.red[3]));
  hi = fmaxf(fmaxf(s.red[4], s.red[5]), fmaxf(s.red[6], s.red[7]));
  const float scale = hi > lo ? 62.5f / (hi - lo) : 0.0f;
  int b = 0, r = 0;
  if (have) {
    if (finite) {
      const int q = (int)((k - lo) * scale);  // monotone in k
      b = 1 + (q < 0 ? 0 : (q > 62 ? 62 : q));
    }
    r = atomicAdd(&s.hist[b], 1);
    atomicMin(&s.bmin[b], __float_as_int(k));  // keys >= 0 order like their bit patterns; -inf only meets -inf
  }
  __syncthreads();
  if (tid < 64) {
    const int c = s.hist[tid];
    int x = c;
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    s.start[tid] = x - c;
  }
  __syncthreads();
  if (have) {
    const int pos = s.start[b] + r;
    order[pos] = tid;
    qlow[pos] = __int_as_float(s.bmin[b]);
  }
  __syncthreads();
}

}  // namespace p3d
