// topk.h -- per-pixel "K nearest" queues for the fine / naive rasterizers.
//
// Target semantics ("Top-K"): the K smallest candidates under the total
// order (z ascending, primitive index ascending), emitted in that order -- what the reference's
// CPU queue (rasterize_meshes_cpu.cpp:281-285) and Python implementation do and what
// test_order_of_ties pins.  The reference's CUDA queue (rasterize_meshes.cu:216-237, an unsorted
// array with a tracked maximum + final BubbleSort) yields the same set except when several queued
// entries tie exactly at the maximum z while the queue overflows.
//
// TopKReg keeps the queue sorted in VGPRs (compile-time capacity KT, fully unrolled
// compare-exchange insertion: no scratch, no divergence inside an insertion).  TopKMem is the
// K <= 150 fallback in private memory.  NP = number of float payload words per entry.
// Host-compilable like p3d_geom.h, so tests/ can exercise the queue logic without a GPU.
#pragma once

#include <float.h>
#include <math.h>
#include <stdint.h>

#include "p3d_geom.h"  // P3D_HDM

namespace p3d {

constexpr int kEmptyIdx = 0x7fffffff;

template <int KT, int NP>
struct TopKReg {
  static constexpr int NPS = NP > 0 ? NP : 1;  // storage rows; NP = 0: no payload (the one row is never touched and costs no register)
  float z[KT];
  int idx[KT];
  float pl[NPS][KT];

  // (kz, ki): the K-th entry of the live queue -- what a candidate has to beat -- cached so that the
  // admission test is one compare also when K < KT (+inf / kEmptyIdx while the queue has room).
  float kz;
  int ki;

  P3D_HDM void init() {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      z[k] = INFINITY;
      idx[k] = kEmptyIdx;
#pragma unroll
      for (int p = 0; p < NP; ++p) pl[p][k] = -1.0f;
    }
    kz = INFINITY;
    ki = kEmptyIdx;
  }

  P3D_HDM void refresh_kth(int K) {
    if (K >= KT) {
      kz = z[KT - 1];
      ki = idx[KT - 1];
    } else {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k == K - 1) {
          kz = z[k];
          ki = idx[k];
        }
      }
    }
  }

  // Sorted insertion; the largest entry falls off the end.  `K` <= KT is the live capacity.
  // Shift formulation: lt[k] = "the candidate sorts before entry k" is monotone in k (the queue is
  // sorted), so entry k becomes  lt[k] ? (lt[k-1] ? old entry k-1 : candidate) : itself.  All KT
  // slots update independently of each other (select depth 2) -- a compare-exchange chain that
  // carries the displaced entry from slot to slot has the same instruction count but a dependency
  // chain 2*KT long, which a SIMD with 3-4 resident waves cannot hide.
  P3D_HDM void insert(int K, float cz, int cidx, const float (&cpl)[NPS]) {
    bool lt[KT];
#pragma unroll
    // bitwise, not short-circuit: `||` / `&&` compile to exec-masked branches (12 instructions per slot instead of 5)
    for (int k = 0; k < KT; ++k) lt[k] = (cz < z[k]) | ((cz == z[k]) & (cidx < idx[k]));
#pragma unroll
    for (int k = KT - 1; k >= 1; --k) {
      const float tz = lt[k - 1] ? z[k - 1] : cz;
      const int ti = lt[k - 1] ? idx[k - 1] : cidx;
      z[k] = lt[k] ? tz : z[k];
      idx[k] = lt[k] ? ti : idx[k];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float tp = lt[k - 1] ? pl[p][k - 1] : cpl[p];
        pl[p][k] = lt[k] ? tp : pl[p][k];
      }
    }
    z[0] = lt[0] ? cz : z[0];
    idx[0] = lt[0] ? cidx : idx[0];
#pragma unroll
    for (int p = 0; p < NP; ++p) pl[p][0] = lt[0] ? cpl[p] : pl[p][0];
    if (K < KT) {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k >= K) {
          z[k] = INFINITY;
          idx[k] = kEmptyIdx;
        }
      }
    }
    refresh_kth(K);
  }

  // Exact pre-test: can (cz, cidx) enter the queue at all?
  P3D_HDM bool admits(int /*K*/, float cz, int cidx) const { return (cz < kz) | ((cz == kz) & (cidx < ki)); }

  // Depth a candidate must not exceed to enter the queue: the K-th entry's z, +inf while the queue has room.
  P3D_HDM float kth_z(int /*K*/) const { return kz; }

  // Position of primitive `want` in the queue, or -1.
  P3D_HDM int find(int want) const {
    int at = -1;
#pragma unroll
    for (int k = KT - 1; k >= 0; --k) {
      if (idx[k] == want) at = k;
    }
    return at;
  }

  P3D_HDM float payload_at(int p, int at) const {
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k == at) v = pl[p][k];
    }
    return v;
  }

  // Remove entry `at` (keeps order).
  P3D_HDM void erase(int at) {
#pragma unroll
    for (int k = 0; k < KT - 1; ++k) {
      if (k >= at) {
        z[k] = z[k + 1];
        idx[k] = idx[k + 1];
#pragma unroll
        for (int p = 0; p < NP; ++p) pl[p][k] = pl[p][k + 1];
      }
    }
    z[KT - 1] = INFINITY;
    idx[KT - 1] = kEmptyIdx;
    kz = INFINITY;  // the queue has room again
    ki = kEmptyIdx;
  }

  P3D_HDM bool valid(int k) const { return idx[k] != kEmptyIdx; }
};

template <int KMAX, int NP>
struct TopKMem {
  float z[KMAX];
  int idx[KMAX];
  float pl[NP][KMAX];
  int n;

  P3D_HDM void init() { n = 0; }

  P3D_HDM void insert(int K, float cz, int cidx, const float (&cpl)[NP]) {
    int j = n;
    while (j > 0 && ((cz < z[j - 1]) || (cz == z[j - 1] && cidx < idx[j - 1]))) --j;
    if (j >= K) return;
    const int last = n < K ? n : K - 1;
    for (int m = last; m > j; --m) {
      z[m] = z[m - 1];
      idx[m] = idx[m - 1];
      for (int p = 0; p < NP; ++p) pl[p][m] = pl[p][m - 1];
    }
    z[j] = cz;
    idx[j] = cidx;
    for (int p = 0; p < NP; ++p) pl[p][j] = cpl[p];
    if (n < K) ++n;
  }

  P3D_HDM bool admits(int K, float cz, int cidx) const {
    if (n < K) return true;
    return (cz < z[K - 1]) || (cz == z[K - 1] && cidx < idx[K - 1]);
  }

  P3D_HDM float kth_z(int K) const { return n < K ? INFINITY : z[K - 1]; }

  P3D_HDM int find(int want) const {
    for (int k = 0; k < n; ++k)
      if (idx[k] == want) return k;
    return -1;
  }

  P3D_HDM float payload_at(int p, int at) const { return pl[p][at]; }

  P3D_HDM void erase(int at) {
    for (int k = at; k < n - 1; ++k) {
      z[k] = z[k + 1];
      idx[k] = idx[k + 1];
      for (int p = 0; p < NP; ++p) pl[p][k] = pl[p][k + 1];
    }
    --n;
  }

  P3D_HDM bool valid(int k) const { return k < n; }
};

}  // namespace p3d
