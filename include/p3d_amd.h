/*
 * p3d_amd.h -- C ABI of libp3d_amd.so: the MI355X (gfx950) implementation of PyTorch3D's
 * differentiable-rasterization hot path.
 *
 * One entry point per operator of the reference's pybind surface `pytorch3d._C`
 * (pytorch3d/csrc/ext.cpp:38-73).  Every function
 *   - takes plain device pointers and sizes (no torch types),
 *   - is asynchronous on the HIP stream it is handed (no allocation, no host sync),
 *   - writes every element of its outputs (padding value -1 included: callers pass
 *     uninitialised memory, there is no pre-fill pass),
 *   - returns P3D_OK or a negative error code (p3d_error_string()).
 * Scratch memory comes from the caller: ask p3d_*_workspace_bytes() first.
 * The current HIP device must be the one that owns the pointers and the stream.
 *
 * Layouts are the reference's: packed AoS face_verts (F,3,3) f32; per-mesh
 * first-index/count vectors (N) i64; outputs (N,H,W,K[,3]).  Output pixel (y, x) looks
 * along flipped axes (+Y up, +X left), rasterize_meshes.cu:271-277.
 */
#ifndef P3D_AMD_H_
#define P3D_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3D_ABI_VERSION 1

#define P3D_OK 0
#define P3D_ERR_INVALID_ARG (-1)   /* null pointer / negative size / bad mode                    */
#define P3D_ERR_K_TOO_LARGE (-2)   /* K > 150 (rasterization_utils.cuh:49, rasterize_meshes.cu:361) */
#define P3D_ERR_TOO_MANY_BINS (-3) /* bins per side >= 22 (rasterize_coarse.cu:244-249)           */
#define P3D_ERR_WORKSPACE (-4)     /* workspace smaller than p3d_*_workspace_bytes()              */
#define P3D_ERR_LAUNCH (-5)        /* HIP reported a launch failure                              */
#define P3D_ERR_UNSUPPORTED (-6)

#define P3D_MAX_K 150
#define P3D_MAX_BINS_PER_SIDE 21

typedef void* p3d_stream_t; /* hipStream_t */

int p3d_abi_version(void);
const char* p3d_error_string(int code);

/* ---- meshes -------------------------------------------------------------------------- */

/* Scratch bytes for p3d_rasterize_meshes / p3d_rasterize_meshes_coarse (0 when bin_size == 0).
 * Sized for the WORST case, because sizing it exactly would need a host sync: the bin lists are reserved as
 * min(F * bins, N * bins * max_faces_per_bin) int32 entries (every face in every bin, or every bin at its cap), plus
 * per-(mesh, bin) counters, offsets and the tile plan (~20 B per bin) and per-(1024-face chunk, bin) partial counts.
 * Only the used prefix is ever touched.  Examples at 512 x 512 (1024 internal bins per image), max_faces_per_bin =
 * max(10000, F / 5) as the reference's wrapper picks it: one 5.8k-face mesh 24 MB; the bench batch (N = 64, F = 321k)
 * 1.3 GB; an un-sharded batch of 512 such meshes 42 GB -- hand over a short workspace (below), shard the batch
 * (pytorch3d_amd/sharding.py) or lower max_faces_per_bin when that matters.  Reuse the workspace across calls; it carries no
 * state between them. */
size_t p3d_rasterize_meshes_workspace_bytes(int64_t F, int N, int H, int W, int bin_size, int max_faces_per_bin);

/* Short workspaces (p3d_rasterize_meshes and p3d_rasterize_meshes_with_cover only; the reference has no counterpart: its
 * coarse stage allocates the padded (N, BH, BW, max_faces_per_bin) tensor, rasterize_coarse.cu:353-354).
 * Those two calls accept ANY workspace of at least p3d_rasterize_meshes_short_workspace_bytes(..., list_entries = 0) bytes:
 * the bin lists get whatever room is left after the fixed arrays.  Whether the lists fit is decided on the device, after
 * the scan, with no host sync: when they do, the binned kernel runs as usual and a second, naive launch returns at once;
 * when they do not, the binned kernel returns at once and the naive kernel (every tile tests every face of its mesh)
 * writes the same outputs -- slower, bit-identical.  The int64 at byte p3d_rasterize_meshes_workspace_need_offset(...) of
 * the workspace holds, once the call has run, the number of list entries it needed: read it back whenever convenient and
 * size the next workspace with p3d_rasterize_meshes_short_workspace_bytes(..., that number plus headroom).  Bench batch:
 * 2.1 M entries = 8.4 MB of lists against the 1.3 GB worst case.  The cost of a short workspace is the second launch
 * (its workgroups exit on a scalar load): 0.026 ms for the 65 536 tiles of the bench batch, whose lists that do not fit cost
 * 6.7 ms instead of 1.5 (profiles/r04/r04c8/). */
size_t p3d_rasterize_meshes_short_workspace_bytes(int64_t F, int N, int H, int W, int bin_size, int max_faces_per_bin,
                                                  int64_t list_entries);
size_t p3d_rasterize_meshes_workspace_need_offset(int64_t F, int N, int H, int W, int bin_size, int max_faces_per_bin);

/* replaces RasterizeMeshes, pytorch3d/csrc/rasterize_meshes/rasterize_meshes.h:513-562
 * (_C.rasterize_meshes).  bin_size == 0 or max_faces_per_bin == 0 -> naive path, else
 * coarse binning + fine rasterization.  Outputs: pix_to_face (N,H,W,K) i64, zbuf (N,H,W,K) f32,
 * bary (N,H,W,K,3) f32, dists (N,H,W,K) f32. */
int p3d_rasterize_meshes(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                         const int64_t* num_faces_per_mesh, const int64_t* clipped_faces_neighbor_idx, int64_t F, int N,
                         int H, int W, float blur_radius, int faces_per_pixel, int bin_size, int max_faces_per_bin,
                         int perspective_correct, int clip_barycentric_coords, int cull_backfaces, int64_t* pix_to_face,
                         float* zbuf, float* bary, float* dists, void* workspace, size_t workspace_bytes,
                         p3d_stream_t stream);

/* replaces RasterizeMeshesNaive, rasterize_meshes.h:108-156 (_C._rasterize_meshes_naive). */
int p3d_rasterize_meshes_naive(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                               const int64_t* num_faces_per_mesh, const int64_t* clipped_faces_neighbor_idx, int64_t F,
                               int N, int H, int W, float blur_radius, int faces_per_pixel, int perspective_correct,
                               int clip_barycentric_coords, int cull_backfaces, int64_t* pix_to_face, float* zbuf,
                               float* bary, float* dists, p3d_stream_t stream);

/* replaces RasterizeMeshesCoarse, rasterize_meshes.h:292-329 (_C._rasterize_meshes_coarse).
 * bin_faces (N,BH,BW,M) i32, -1 padded, each bin's list ascending; a bin with more than M faces keeps
 * its first M (the reference drops an unspecified subset, rasterize_coarse.cu:186-201). */
int p3d_rasterize_meshes_coarse(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                                const int64_t* num_faces_per_mesh, int64_t F, int N, int H, int W, float blur_radius,
                                int bin_size, int max_faces_per_bin, int32_t* bin_faces, void* workspace,
                                size_t workspace_bytes, p3d_stream_t stream);

/* Scratch bytes for p3d_rasterize_meshes_fine / p3d_rasterize_points_fine. */
size_t p3d_rasterize_fine_workspace_bytes(int N, int BH, int BW, int M);

/* replaces RasterizeMeshesFine, rasterize_meshes.h:406-441 (_C._rasterize_meshes_fine).
 * bin_faces (N,BH,BW,M) i32 with -1 sentinels anywhere. */
int p3d_rasterize_meshes_fine(const float* face_verts, const int32_t* bin_faces,
                              const int64_t* clipped_faces_neighbor_idx, int64_t F, int N, int BH, int BW, int M, int H,
                              int W, float blur_radius, int bin_size, int faces_per_pixel, int perspective_correct,
                              int clip_barycentric_coords, int cull_backfaces, int64_t* pix_to_face, float* zbuf,
                              float* bary, float* dists, void* workspace, size_t workspace_bytes, p3d_stream_t stream);

/* replaces RasterizeMeshesBackward, rasterize_meshes.h:211-252 (_C.rasterize_meshes_backward).
 * grad_face_verts (F,3,3) f32 is zeroed and accumulated here. */
int p3d_rasterize_meshes_backward(const float* face_verts, const int64_t* pix_to_face, const float* grad_zbuf,
                                  const float* grad_bary, const float* grad_dists, int64_t F, int N, int H, int W, int K,
                                  int perspective_correct, int clip_barycentric_coords, float* grad_face_verts,
                                  p3d_stream_t stream);
/* The same backward with the gradient of `face_verts = verts_packed[faces_packed]` (rasterize_meshes.py:146, torch
 * indexing + its index_put backward) fused in: the per-face partials are flushed straight to grad_verts (V,3) through
 * faces (F,3) -- no (F,3,3) intermediate, no separate scatter.  grad_verts zeroed and accumulated. */
int p3d_rasterize_meshes_backward_verts(const float* face_verts, const int64_t* faces, const int64_t* pix_to_face,
                                        const float* grad_zbuf, const float* grad_bary, const float* grad_dists, int64_t F,
                                        int64_t V, int N, int H, int W, int K, int perspective_correct,
                                        int clip_barycentric_coords, float* grad_verts, p3d_stream_t stream);

/* ---- row cover: what the forward knows about empty image regions, handed to the backward (round 3) ----
 * The reference's autograd node saves pix_to_face for the backward (renderer/mesh/rasterize_meshes.py:291-296) and the
 * backward kernel reads all N*H*W*K entries of it to find the samples that hold a face (rasterize_meshes.cu:593-603).  At
 * the bench workload 68 % of those reads find nothing.  The forward can say so for free: `cover` is (N, ceil(H/16),
 * ceil(W/16)) int32; bit r of word (n, cy, cx) is set iff some pixel of output row 16*cy + r, columns 16*cx .. 16*cx+15 of
 * image n holds a face (pix_to_face[n, y, x, 0] >= 0).  The autograd nodes of pytorch3d_amd/rasterize_meshes.py save it
 * next to pix_to_face; a backward without cover (the `_C.rasterize_meshes_backward` drop-in) reads everything, as before.
 * The backward TRUSTS the cover: a clear bit skips the row without looking. */
size_t p3d_rasterize_meshes_cover_bytes(int N, int H, int W);

/* Is `cover` still the cover of `pix_to_face`?  *stale (one int32 on the device, written here) becomes non-zero iff some 16-pixel
 * row segment holds a face the cover does not know of -- the one way a cover can make the backward wrong (a set bit over an
 * empty segment only costs time).  For callers that cannot rule out writes into pix_to_face between the forward and the
 * backward (pytorch3d_amd._C with P3D_CHECK=1: tensors edited through `.data`); reads slot 0 of every pixel, ~0.2 ms at the
 * bench size -- about half of what the cover saves.  The reference has no counterpart (its backward reads every entry). */
int p3d_rasterize_meshes_cover_check(const int64_t* pix_to_face, const int32_t* cover, int N, int H, int W, int K, int32_t* stale,
                                     p3d_stream_t stream);

/* p3d_rasterize_meshes + the row cover of its output (cover may be null: then identical to p3d_rasterize_meshes). */
int p3d_rasterize_meshes_with_cover(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                                    const int64_t* num_faces_per_mesh, const int64_t* clipped_faces_neighbor_idx, int64_t F,
                                    int N, int H, int W, float blur_radius, int faces_per_pixel, int bin_size,
                                    int max_faces_per_bin, int perspective_correct, int clip_barycentric_coords,
                                    int cull_backfaces, int64_t* pix_to_face, float* zbuf, float* bary, float* dists,
                                    int32_t* cover, void* workspace, size_t workspace_bytes, p3d_stream_t stream);

/* The cover AND the list of its non-empty words (round 6): `cover_and_list` is one buffer of p3d_rasterize_meshes_cover_list_bytes --
 * the (N, ceil(H/16), ceil(W/16)) words of the cover as above, an int32 counter (+ 15 spare), then room for one int32 per word.
 * The wave that sets the first bit of a word appends the word's index to the list (one atomic per 16 x 16 pixel block that holds a
 * face), so the backward finds its work without a pass over the cover: p3d_rasterize_meshes_backward[_verts]_with_cover_list take
 * the same buffer and no workspace, and launch two kernels fewer than the _with_cover forms (the list builder and the memset of
 * its counter; 0.02 ms of the 2.3 ms bench step).  The words in front are a plain cover: the buffer may be handed to every
 * function that takes `cover`.  The list's order is the order in which the forward's tiles finished. */
size_t p3d_rasterize_meshes_cover_list_bytes(int N, int H, int W);
int p3d_rasterize_meshes_with_cover_list(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                                         const int64_t* num_faces_per_mesh, const int64_t* clipped_faces_neighbor_idx, int64_t F,
                                         int N, int H, int W, float blur_radius, int faces_per_pixel, int bin_size,
                                         int max_faces_per_bin, int perspective_correct, int clip_barycentric_coords,
                                         int cull_backfaces, int64_t* pix_to_face, float* zbuf, float* bary, float* dists,
                                         int32_t* cover_and_list, void* workspace, size_t workspace_bytes, p3d_stream_t stream);
int p3d_rasterize_meshes_backward_with_cover_list(const float* face_verts, const int64_t* pix_to_face, const float* grad_zbuf,
                                                  const float* grad_bary, const float* grad_dists, const int32_t* cover_and_list,
                                                  int64_t F, int N, int H, int W, int K, int perspective_correct,
                                                  int clip_barycentric_coords, float* grad_face_verts, p3d_stream_t stream);
int p3d_rasterize_meshes_backward_verts_with_cover_list(const float* face_verts, const int64_t* faces, const int64_t* pix_to_face,
                                                        const float* grad_zbuf, const float* grad_bary, const float* grad_dists,
                                                        const int32_t* cover_and_list, int64_t F, int64_t V, int N, int H, int W,
                                                        int K, int perspective_correct, int clip_barycentric_coords,
                                                        float* grad_verts, p3d_stream_t stream);

/* Per-face reciprocals for the backward (round 6; the reference has no counterpart: its backward re-derives them per sample,
 * geometry_utils.cuh:101-161, 365-385).  p3d_gather_face_verts_pre is p3d_gather_face_verts with a thread per face that also writes
 * face_pre (F, 4) f32, 16-byte aligned: 1 / (barycentric area), 1 / |v1 - v0|^2, 1 / |v2 - v0|^2, 1 / |v2 - v1|^2 (-1 where the squared
 * length is <= 1e-8: the degenerate-edge rule).  p3d_rasterize_meshes_backward_verts_pre is
 * p3d_rasterize_meshes_backward_verts_with_cover_list (cover_and_list may be null: then every row is read) reading those instead of
 * forming them per sample; face_pre null: identical to the _with_cover_list form.  face_pre must belong to THESE face_verts. */
int p3d_gather_face_verts_pre(const float* verts, const int64_t* faces, int64_t V, int64_t F, float* face_verts, float* face_pre,
                              p3d_stream_t stream);
int p3d_rasterize_meshes_backward_verts_pre(const float* face_verts, const float* face_pre, const int64_t* faces,
                                            const int64_t* pix_to_face, const float* grad_zbuf, const float* grad_bary,
                                            const float* grad_dists, const int32_t* cover_and_list, int64_t F, int64_t V, int N, int H,
                                            int W, int K, int perspective_correct, int clip_barycentric_coords, float* grad_verts,
                                            p3d_stream_t stream);

/* The same for callers that arrive with face_verts already made (the reference's own signature, `_C.rasterize_meshes_backward`):
 * p3d_rasterize_meshes_backward_with_cover / _with_cover_list (cover null: every row is read; cover_has_list != 0: the list behind the
 * cover is taken and the workspace ignored) that first writes the per-face reciprocals into face_pre_scratch (F x 4 f32, 16-byte aligned;
 * null: the per-sample form) with one small launch. */
int p3d_rasterize_meshes_backward_pre(const float* face_verts, const int64_t* pix_to_face, const float* grad_zbuf, const float* grad_bary,
                                      const float* grad_dists, const int32_t* cover, int cover_has_list, int64_t F, int N, int H, int W,
                                      int K, int perspective_correct, int clip_barycentric_coords, float* grad_face_verts,
                                      float* face_pre_scratch, void* workspace, size_t workspace_bytes, p3d_stream_t stream);

/* CUDA tie order -- p3d_rasterize_meshes_with_cover, then a replay that makes pix_to_face (and the rows that go with it) what
 * the reference's CUDA kernels return where faces tie EXACTLY in depth at a pixel's K-th place.  The kernels of this library keep
 * the K nearest under the total order (depth, face index), as the reference's CPU and Python implementations do
 * (rasterize_meshes_cpu.cpp:263-288, rasterize_meshes.py); its CUDA kernels keep an unsorted array and replace "the" farthest entry
 * only by a strictly nearer candidate (RasterizeMeshesFineCudaKernel / CheckPixelInsideFace, rasterize_meshes.cu:112-237): the
 * same depths, but among faces of exactly the K-th depth possibly other survivors, depending on array positions, i.e. on the
 * pixel's whole history (2 in 10^4 entries of the bench launch; zbuf is bit-equal either way).  The fine kernel marks the
 * pixels in which the two procedures can differ (an entry dropped at the depth of the last survivor while a nearer survivor has
 * a larger face index; or the clipped-face neighbour rule in play: 2 in 10^3 pixels of the bench launch) and the replay re-runs
 * the reference's procedure, faces in ascending index, for those: 1.2 x the time of p3d_rasterize_meshes on the bench batch
 * (round 4: ~10 x).  The marks take the LAST N * ceil(H/8) * ceil(W/8) * 8 bytes (rounded up to 256) of the workspace when it
 * is at least that much larger than the binning needs (p3d_rasterize_meshes_workspace_bytes counts them in; a caller of the
 * short-workspace size adds them); without that room the replay finds the marks in the output itself (one pix_to_face entry of
 * every pixel is read).  Diagnostic: with P3D_TIE_SKIP_REPLAY set in the environment the replay is skipped and the marks (-2)
 * stay in pix_to_face (profiles/tie_order_timing.py --count-marks). */
int p3d_rasterize_meshes_cuda_order(const float* face_verts, const int64_t* mesh_to_face_first_idx,
                                    const int64_t* num_faces_per_mesh, const int64_t* clipped_faces_neighbor_idx, int64_t F,
                                    int N, int H, int W, float blur_radius, int faces_per_pixel, int bin_size,
                                    int max_faces_per_bin, int perspective_correct, int clip_barycentric_coords,
                                    int cull_backfaces, int64_t* pix_to_face, float* zbuf, float* bary, float* dists,
                                    int32_t* cover, void* workspace, size_t workspace_bytes, p3d_stream_t stream);

/* the two backward entry points with the cover of THAT pix_to_face (null: all rows are read).  workspace (optional, may be
 * null; p3d_rasterize_meshes_backward_workspace_bytes): room for the list of covered 16 x 16 areas, so that the launch holds
 * only workgroups with work -- workgroups reach the CUs round robin, and a mix of empty and full ones leaves CUs idle. */
size_t p3d_rasterize_meshes_backward_workspace_bytes(int N, int H, int W);
int p3d_rasterize_meshes_backward_with_cover(const float* face_verts, const int64_t* pix_to_face, const float* grad_zbuf,
                                             const float* grad_bary, const float* grad_dists, const int32_t* cover, int64_t F,
                                             int N, int H, int W, int K, int perspective_correct,
                                             int clip_barycentric_coords, float* grad_face_verts, void* workspace,
                                             size_t workspace_bytes, p3d_stream_t stream);
int p3d_rasterize_meshes_backward_verts_with_cover(const float* face_verts, const int64_t* faces, const int64_t* pix_to_face,
                                                   const float* grad_zbuf, const float* grad_bary, const float* grad_dists,
                                                   const int32_t* cover, int64_t F, int64_t V, int N, int H, int W, int K,
                                                   int perspective_correct, int clip_barycentric_coords, float* grad_verts,
                                                   void* workspace, size_t workspace_bytes, p3d_stream_t stream);

/* ---- packed vertices <-> per-face vertices (optional fast path of the L2 function) ------ */

/* replaces the Python-side gather `face_verts = verts_packed[faces_packed]`
 * (pytorch3d/renderer/mesh/rasterize_meshes.py:144-148): verts (V,3) f32, faces (F,3) i64 -> face_verts (F,3,3). */
int p3d_gather_face_verts(const float* verts, const int64_t* faces, int64_t V, int64_t F, float* face_verts,
                          p3d_stream_t stream);
/* its autograd backward (torch: index_put_ accumulate, a sort on ROCm): grad_verts (V,3) is zeroed and
 * accumulated here with f32 atomics (order not deterministic). */
int p3d_scatter_face_grads(const float* grad_face_verts, const int64_t* faces, int64_t V, int64_t F, float* grad_verts,
                           p3d_stream_t stream);

/* ---- world -> NDC vertex transform fused into the gather ---------------- */

/* replaces MeshRasterizer.transform (pytorch3d/renderer/mesh/rasterizer.py:171-216: two batched 4x4 transform_points
 * with homogeneous divides on padded vertices, z taken from view space) + the gather `verts_packed[faces_packed]`
 * (renderer/mesh/rasterize_meshes.py:144-148) by ONE launch: verts_world (V,3) f32, faces (F,3) i64 packed,
 * mesh_to_face_first_idx (N) i64, matrices (num_matrices,2,4,4) f32 row-major in the reference's row-vector convention
 * ([.][0] world->view, [.][1] view->NDC), num_matrices = N or 1 -> face_verts (F,3,3) in NDC (x, y) + view depth. */
int p3d_transform_gather_face_verts(const float* verts_world, const int64_t* faces, const int64_t* mesh_to_face_first_idx,
                                    const float* matrices, int64_t V, int64_t F, int N, int num_matrices, float* face_verts,
                                    p3d_stream_t stream);
/* the same transform per packed vertex (for callers that need the NDC vertices themselves) ... */
int p3d_transform_verts_forward(const float* verts_world, const int64_t* mesh_to_vert_first_idx, const float* matrices,
                                int64_t V, int N, int num_matrices, float* verts_ndc, p3d_stream_t stream);
/* ... and its backward: grad_verts_world (V,3) = J^T grad_verts_ndc (V,3), i.e. what torch autograd computes through the
 * two transform_points calls; applied to the output of p3d_rasterize_meshes_backward_verts it gives the gradient of the
 * rasterization wrt world-space vertices. */
int p3d_transform_verts_backward(const float* verts_world, const int64_t* mesh_to_vert_first_idx, const float* matrices,
                                 const float* grad_verts_ndc, int64_t V, int N, int num_matrices, float* grad_verts_world,
                                 p3d_stream_t stream);

/* ---- point clouds -------------------------------------------------------------------- */

size_t p3d_rasterize_points_workspace_bytes(int64_t P, int N, int H, int W, int bin_size, int max_points_per_bin);
/* Short workspaces for p3d_rasterize_points, exactly as for the meshes (above): any workspace of at least
 * p3d_rasterize_points_short_workspace_bytes(..., 0) bytes is accepted, the lists take what is left, the device decides whether
 * they fit and the naive kernel writes the same outputs when they do not.  (1M points, 512 x 512, max_points_per_bin = P / 5:
 * worst case 0.8 GB for one cloud, 1.3 M entries = 5 MB needed.) */
size_t p3d_rasterize_points_short_workspace_bytes(int64_t P, int N, int H, int W, int bin_size, int max_points_per_bin,
                                                  int64_t list_entries);
size_t p3d_rasterize_points_workspace_need_offset(int64_t P, int N, int H, int W, int bin_size, int max_points_per_bin);

/* replaces RasterizePoints, pytorch3d/csrc/rasterize_points/rasterize_points.h:343-374 (_C.rasterize_points).
 * Outputs idxs (N,H,W,K) i32, zbuf, dists (squared) f32. */
int p3d_rasterize_points(const float* points, const int64_t* cloud_to_packed_first_idx,
                         const int64_t* num_points_per_cloud, const float* radius, int64_t P, int N, int H, int W,
                         int points_per_pixel, int bin_size, int max_points_per_bin, int32_t* idxs, float* zbuf,
                         float* dists, void* workspace, size_t workspace_bytes, p3d_stream_t stream);

/* p3d_rasterize_points, then the CUDA tie order (see p3d_rasterize_meshes_cuda_order): the reference's point kernels keep the same
 * unsorted array (rasterize_points.cu:38-84) and sort it by depth ALONE (rasterize_points.cu:26-28, stable): where points tie
 * exactly in depth, the survivors at the K-th place and the order of the tied entries follow the array positions.  The replay
 * re-runs that procedure, points in ascending index, for every pixel whose K slots are full. */
int p3d_rasterize_points_cuda_order(const float* points, const int64_t* cloud_to_packed_first_idx,
                                    const int64_t* num_points_per_cloud, const float* radius, int64_t P, int N, int H, int W,
                                    int points_per_pixel, int bin_size, int max_points_per_bin, int32_t* idxs, float* zbuf,
                                    float* dists, void* workspace, size_t workspace_bytes, p3d_stream_t stream);

/* replaces RasterizePointsNaive, rasterize_points.h:70-99 (_C._rasterize_points_naive). */
int p3d_rasterize_points_naive(const float* points, const int64_t* cloud_to_packed_first_idx,
                               const int64_t* num_points_per_cloud, const float* radius, int64_t P, int N, int H, int W,
                               int points_per_pixel, int32_t* idxs, float* zbuf, float* dists, p3d_stream_t stream);

/* replaces RasterizePointsCoarse, rasterize_points.h:146-191 (_C._rasterize_points_coarse). */
int p3d_rasterize_points_coarse(const float* points, const int64_t* cloud_to_packed_first_idx,
                                const int64_t* num_points_per_cloud, const float* radius, int64_t P, int N, int H, int W,
                                int bin_size, int max_points_per_bin, int32_t* bin_points, void* workspace,
                                size_t workspace_bytes, p3d_stream_t stream);

/* replaces RasterizePointsFine, rasterize_points.h:222-247. */
int p3d_rasterize_points_fine(const float* points, const int32_t* bin_points, const float* radius, int64_t P, int N,
                              int BH, int BW, int M, int H, int W, int bin_size, int points_per_pixel, int32_t* idxs,
                              float* zbuf, float* dists, void* workspace, size_t workspace_bytes, p3d_stream_t stream);

/* PointsRenderer's chain as two launches (round 6; renderer/points/renderer.py:56-76: fragments = rasterize_points(...),
 * weights = 1 - dists / r^2, images = alpha_composite(idx, weights, features)).  The reference has no single operator for it; the
 * patched PointsRenderer (pytorch3d_amd.shim) and pytorch3d_amd.render_points call these.  mode: P3D_COMPOSITE_ALPHA (AlphaCompositor) or
 * P3D_COMPOSITE_NORM_SUM (NormWeightedCompositor: norm_weighted_sum.cu:24-154 in place of alpha_composite.cu below; the constants are
 * defined further down with the compositing operators).
 *   p3d_rasterize_points_composite: p3d_rasterize_points (same arguments, same workspace, same idxs / zbuf / dists) that also writes
 *     images (N, H, W, C) f32 = the alpha compositing (alpha_composite.cu:24-68) of features (P, C) f32 rows, C in 1..4, with
 *     alpha = 1 - dists * inv_r2, inv_r2 = float(1) / float(r * r) (how torch evaluates `dists / (r * r)`): the pixel is formed in the
 *     fine kernel's epilogue while its K entries are in LDS (K <= 28, binned; with a short workspace whose lists did not fit: by a pass
 *     behind the stand-by kernel, decided on the device), else by a pass behind the rasterizer.
 *     Bit-equal to the three operators run one after the other.
 *   p3d_rasterize_points_composite_backward: grad_points (P, 3) [z column zero: the chain does not expose zbuf] and grad_features
 *     (P, C), both fully written, from grad_images (N, H, W, C): alphaCompositeCudaBackwardKernel (alpha_composite.cu:72-141),
 *     grad_dists = -grad_alphas * inv_r2 and RasterizePointsBackwardCudaKernel (rasterize_points.cu:366-411) as ONE kernel whose two
 *     scatters share a wave-private table.  K <= 16, C in 1..4 (P3D_ERR_INVALID_ARG otherwise: run the three operators instead). */
int p3d_rasterize_points_composite(int mode, const float* points, const int64_t* cloud_to_packed_first_idx,
                                   const int64_t* num_points_per_cloud, const float* radius, const float* features, int64_t P, int C,
                                   int N, int H, int W, int points_per_pixel, int bin_size, int max_points_per_bin, float inv_r2,
                                   int32_t* idxs, float* zbuf, float* dists, float* images, void* workspace, size_t workspace_bytes,
                                   p3d_stream_t stream);
int p3d_rasterize_points_composite_backward(int mode, const float* points, const float* features, const int32_t* idxs,
                                            const float* dists, const float* grad_images, int64_t P, int C, int N, int H, int W,
                                            int points_per_pixel, float inv_r2, float* grad_points, float* grad_features,
                                            p3d_stream_t stream);

/* replaces RasterizePointsBackward, rasterize_points.h:281-305 (_C.rasterize_points_backward). */
int p3d_rasterize_points_backward(const float* points, const int32_t* idxs, const float* grad_zbuf,
                                  const float* grad_dists, int64_t P, int N, int H, int W, int K, float* grad_points,
                                  p3d_stream_t stream);

/* ---- compositors --------------------------------------------------------------------- */

#define P3D_COMPOSITE_ALPHA 0    /* alphaComposite*,  compositing/alpha_composite.h:59-115    */
#define P3D_COMPOSITE_NORM_SUM 1 /* weightedSumNorm*, compositing/norm_weighted_sum.h:57-115  */
#define P3D_COMPOSITE_SUM 2      /* weightedSum*,     compositing/weighted_sum.h:55-111       */

/* features (C,P) f32 contiguous; alphas / points_idx are logically (N,K,H,W) and addressed through
 * element strides (so the permuted (N,H,W,K) views the renderers pass need no copy);
 * result (N,C,H,W) f32 contiguous. */
int p3d_composite_forward(int mode, const float* features, const float* alphas, const int64_t* points_idx, int N, int C,
                          int64_t P, int K, int H, int W, const int64_t alphas_strides[4],
                          const int64_t idx_strides[4], float* result, p3d_stream_t stream);

/* grad_features (C,P) and grad_alphas (N,K,H,W) contiguous, both fully written. */
int p3d_composite_backward(int mode, const float* grad_outputs, const float* features, const float* alphas,
                           const int64_t* points_idx, int N, int C, int64_t P, int K, int H, int W,
                           const int64_t alphas_strides[4], const int64_t idx_strides[4], float* grad_features,
                           float* grad_alphas, p3d_stream_t stream);

/* The same operators with the FEATURES addressed through element strides (channel, point): (P, 1) is the contiguous (C,P) tensor
 * above, (1, C) the transposed view of a (P, C) tensor -- what PointsRenderer passes (renderer/points/renderer.py:67:
 * `features_packed().permute(1, 0)`; the reference's launchers copy it to (C,P) first, compositing/alpha_composite.h:63-65).
 * With (1, C) a point's channels share a cache line: the gathers of a pixel cost one memory request per entry instead of C.
 * grad_features: C * P floats of one allocation in the layout grad_feature_strides names, fully written. */
int p3d_composite_forward_strided(int mode, const float* features, const int64_t feature_strides[2], const float* alphas,
                                  const int64_t* points_idx, int N, int C, int64_t P, int K, int H, int W,
                                  const int64_t alphas_strides[4], const int64_t idx_strides[4], float* result,
                                  p3d_stream_t stream);
int p3d_composite_backward_strided(int mode, const float* grad_outputs, const float* features, const int64_t feature_strides[2],
                                   const float* alphas, const int64_t* points_idx, int N, int C, int64_t P, int K, int H, int W,
                                   const int64_t alphas_strides[4], const int64_t idx_strides[4], float* grad_features,
                                   const int64_t grad_feature_strides[2], float* grad_alphas, p3d_stream_t stream);

/* ---- interpolate_face_attributes ----------------------------------------------------- */

/* replaces InterpFaceAttrsForward/Backward, pytorch3d/csrc/interp_face_attrs/interp_face_attrs.h:46-116.
 * dtype: 0 = f32, 1 = f64 (the reference dispatches both).  pix_attrs (P,D) fully written. */
int p3d_interp_face_attrs_forward(int dtype, const int64_t* pix_to_face, const void* barycentric_coords,
                                  const void* face_attrs, int64_t P, int64_t F, int64_t D, void* pix_attrs,
                                  p3d_stream_t stream);
int p3d_interp_face_attrs_backward(int dtype, const int64_t* pix_to_face, const void* barycentric_coords,
                                   const void* face_attrs, const void* grad_pix_attrs, int64_t P, int64_t F, int64_t D,
                                   void* grad_barycentric_coords, void* grad_face_attrs, p3d_stream_t stream);

/* ---- frustum culling / z-plane clipping before rasterization, un-clipping after -------- */

/* Together these replace clip_faces and convert_clipped_rasterization_to_original_faces,
 * pytorch3d/renderer/mesh/clip.py:324-615, 618-734 (pure torch in the reference).
 *
 * plan:  classify every face (1 kept, 2 removed, 3 clipped to one triangle, 4 clipped to two) and scan the
 *        destination indices.  `plan` is caller-allocated scratch of p3d_clip_faces_plan_bytes(F) that must stay
 *        alive until emit / backward; its first four int64 receive {F_clipped, T3, T4, F} -- read them (one sync,
 *        as in the reference) to size the outputs.  planes = {left, right, top, bottom, znear, zfar}, bit i of
 *        plane_mask set when plane i is used; cull as ClipFrustum.cull.
 * emit:  face_verts_clipped (F_clipped,3,3), mesh_to_face_first_idx / num_faces_per_mesh (N),
 *        faces_clipped_to_unclipped_idx (F_clipped); when T3 + T4 > 0 also barycentric_conversion (T3 + 2*T4,3,3),
 *        faces_clipped_to_conversion_idx and clipped_faces_neighbor_idx (F_clipped).
 * backward: gradient of (face_verts_clipped, barycentric_conversion) w.r.t. face_verts, with the reference's
 *        autograd semantics (w3 detached, clip.py:291). */
size_t p3d_clip_faces_plan_bytes(int64_t F);
int p3d_clip_faces_plan(const float* face_verts, int64_t F, const float planes[6], int plane_mask, int cull,
                        int has_z_clip, float z_clip_value, void* plan, size_t plan_bytes, p3d_stream_t stream);
int p3d_clip_faces_emit(const float* face_verts, int64_t F, const int64_t* mesh_to_face_first_idx, int N,
                        const void* plan, size_t plan_bytes, int64_t F_clipped, int64_t T3, int64_t T4,
                        float z_clip_value, int perspective_correct, float* face_verts_clipped,
                        int64_t* mesh_to_face_first_idx_clipped, int64_t* num_faces_per_mesh_clipped,
                        int64_t* faces_clipped_to_unclipped_idx, float* barycentric_conversion,
                        int64_t* faces_clipped_to_conversion_idx, int64_t* clipped_faces_neighbor_idx,
                        p3d_stream_t stream);
int p3d_clip_faces_backward(const float* face_verts, int64_t F, const void* plan, size_t plan_bytes, int64_t T3,
                            int64_t T4, float z_clip_value, int perspective_correct,
                            const float* grad_face_verts_clipped, const float* grad_barycentric_conversion,
                            float* grad_face_verts, p3d_stream_t stream);
/* pix_to_face (S) / bary (S,3) of the clipped faces -> of the original faces; S = N*H*W*K samples.
 * barycentric_conversion may be null (only culling happened).  Backward: grad_bary_clipped (S,3) fully written,
 * grad_barycentric_conversion (T,3,3) zeroed and accumulated (may be null). */
int p3d_convert_clipped_forward(const int64_t* pix_to_face_clipped, const float* bary_coords_clipped,
                                const int64_t* faces_clipped_to_unclipped_idx, const float* barycentric_conversion,
                                const int64_t* faces_clipped_to_conversion_idx, int64_t num_samples,
                                int64_t* pix_to_face_unclipped, float* bary_coords_unclipped, p3d_stream_t stream);
int p3d_convert_clipped_backward(const int64_t* pix_to_face_clipped, const float* bary_coords_clipped,
                                 const float* barycentric_conversion, const int64_t* faces_clipped_to_conversion_idx,
                                 const float* grad_bary_unclipped, int64_t num_samples, int64_t T,
                                 float* grad_bary_clipped, float* grad_barycentric_conversion, p3d_stream_t stream);

/* ---- fragment blending (the step right after rasterization) --------------------------- */

/* replaces SigmoidAlphaBlend / SigmoidAlphaBlendBackward, pytorch3d/csrc/blending/sigmoid_alpha_blend.h:73-103
 * (_C.sigmoid_alpha_blend[_backward]).  dists, pix_to_face (npix,K); alphas, grad_alphas (npix); grad_dists
 * (npix,K) fully written.  npix = N*H*W. */
int p3d_sigmoid_alpha_blend_forward(const float* dists, const int64_t* pix_to_face, float sigma, int64_t npix, int K,
                                    float* alphas, p3d_stream_t stream);
int p3d_sigmoid_alpha_blend_backward(const float* grad_alphas, const float* alphas, const float* dists,
                                     const int64_t* pix_to_face, float sigma, int64_t npix, int K, float* grad_dists,
                                     p3d_stream_t stream);

/* replaces the Python function softmax_rgb_blend (pytorch3d/renderer/blending.py:147-244: ~20 elementwise torch
 * ops over (N,H,W,K)) and its autograd graph.  colors (N*P,K,3), pix_to_face / dists / zbuf (N*P,K) with
 * P = pix_per_image; znear / zfar either the scalars or, when the pointers are non-null, per-image device arrays
 * (N); out (N*P,4) RGBA.  Backward: grad_out (N*P,4) -> grad_colors (N*P,K,3), grad_dists, grad_zbuf (N*P,K). */
int p3d_softmax_rgb_blend_forward(const float* colors, const int64_t* pix_to_face, const float* dists,
                                  const float* zbuf, float sigma, float gamma, const float background[3], float znear,
                                  float zfar, const float* znear_per_image, const float* zfar_per_image, int64_t N,
                                  int64_t pix_per_image, int K, float* out, p3d_stream_t stream);
int p3d_softmax_rgb_blend_backward(const float* grad_out, const float* colors, const int64_t* pix_to_face,
                                   const float* dists, const float* zbuf, float sigma, float gamma,
                                   const float background[3], float znear, float zfar, const float* znear_per_image,
                                   const float* zfar_per_image, int64_t N, int64_t pix_per_image, int K,
                                   float* grad_colors, float* grad_dists, float* grad_zbuf, p3d_stream_t stream);

/* The same backward when the caller knows that the P samples are image-shaped fragments (N,H,W,K) -- what
 * interpolate_face_attributes has before it flattens them (pytorch3d/ops/interp_face_attrs.py:57-63): lanes map to
 * 8x8 pixel tiles instead of 64 consecutive samples.  f32, D <= 4. */
int p3d_interp_face_attrs_backward_nhwk(const int64_t* pix_to_face, const float* barycentric_coords,
                                        const float* face_attrs, const float* grad_pix_attrs, int N, int H, int W, int K,
                                        int64_t F, int D, float* grad_barycentric_coords, float* grad_face_attrs,
                                        p3d_stream_t stream);

/* ---- per-pixel Phong shading of the fragments ------------------------ */

/* replaces the Python function phong_shading (pytorch3d/renderer/mesh/shading.py:59-112: two
 * interpolate_face_attributes calls + lights.diffuse / lights.specular, renderer/lighting.py:17-159, + the colour
 * mix; ~45 torch kernels over (N,H,W,K,3) tensors and their autograd graph) with one kernel each way.
 *   face_attrs (F,3,D): D = 6 -> [vertex xyz | vertex normal] per face corner and `texels` (N,H,W,K,3) given per
 *                       sample (the reference's signature);  D = 9 -> [.. | vertex colour], texels interpolated
 *                       in the kernel (TexturesVertex), `texels` / `grad_texels` unused (may be null).
 *   params (N, P3D_SHADE_PARAM_FLOATS): light ambient, diffuse, specular colour (3 each), light location
 *                       (P3D_LIGHT_POINT) or direction (P3D_LIGHT_DIRECTIONAL), material ambient, diffuse,
 *                       specular colour, shininess, camera centre -- already broadcast to the batch.
 *                       AmbientLights = directional with zero diffuse and specular colour.
 *   colors (N,H,W,K,3) fully written.  Backward: grad_bary (N,H,W,K,3) and grad_texels fully written,
 *   grad_face_attrs (F,3,D) zeroed and accumulated; grad_params (N, P3D_SHADE_PARAM_FLOATS), the gradient of the
 *   lights / materials / camera centre, zeroed and accumulated when non-null (null: not computed). */
#define P3D_SHADE_PARAM_FLOATS 25
#define P3D_LIGHT_DIRECTIONAL 0
#define P3D_LIGHT_POINT 1
int p3d_phong_shade_forward(const int64_t* pix_to_face, const float* bary_coords, const float* face_attrs, int D,
                            const float* texels, const float* params, int light_kind, int N, int H, int W, int K,
                            int64_t F, float* colors, p3d_stream_t stream);
int p3d_phong_shade_backward(const float* grad_colors, const int64_t* pix_to_face, const float* bary_coords,
                             const float* face_attrs, int D, const float* texels, const float* params, int light_kind,
                             int N, int H, int W, int K, int64_t F, float* grad_bary_coords, float* grad_face_attrs,
                             float* grad_texels, float* grad_params, p3d_stream_t stream);

/* ---- SoftPhongShader in one kernel each way: Phong shading fused with softmax_rgb_blend --------------------------------
 *
 * replaces SoftPhongShader.forward (pytorch3d/renderer/mesh/shader.py:113-147): phong_shading (renderer/mesh/shading.py:
 * 100-125) followed by softmax_rgb_blend (renderer/blending.py:147-244).  Same inputs as p3d_phong_shade_* plus the
 * fragments' dists / zbuf and the blend parameters of p3d_softmax_rgb_blend_*; the per-sample colours (N,H,W,K,3) and
 * their gradient never reach memory.  K must be 1, 2, 4, 8 or 16 (p3d_soft_phong_supported_k; other K: run the two
 * operators one after the other), otherwise P3D_ERR_INVALID_ARG.
 *   forward: out (N,H,W,4) RGBA fully written.
 *   backward: grad_out (N,H,W,4) -> grad_bary (N,H,W,K,3), grad_dists, grad_zbuf (N,H,W,K) [and grad_texels (N,H,W,K,3)
 *   with D = 6] fully written; grad_face_attrs (F,3,D) zeroed and accumulated; grad_params (N, 25) zeroed and
 *   accumulated when non-null. */
int p3d_soft_phong_supported_k(int K);
int p3d_soft_phong_forward(const int64_t* pix_to_face, const float* bary_coords, const float* dists, const float* zbuf,
                           const float* face_attrs, int D, const float* texels, const float* params, int light_kind,
                           float sigma, float gamma, const float background[3], float znear, float zfar,
                           const float* znear_per_image, const float* zfar_per_image, int N, int H, int W, int K, int64_t F,
                           float* out, p3d_stream_t stream);
int p3d_soft_phong_backward(const float* grad_out, const int64_t* pix_to_face, const float* bary_coords, const float* dists,
                            const float* zbuf, const float* face_attrs, int D, const float* texels, const float* params,
                            int light_kind, float sigma, float gamma, const float background[3], float znear, float zfar,
                            const float* znear_per_image, const float* zfar_per_image, int N, int H, int W, int K, int64_t F,
                            float* grad_bary, float* grad_dists, float* grad_zbuf, float* grad_face_attrs, float* grad_texels,
                            float* grad_params, p3d_stream_t stream);

/* replaces TexturesUV.sample_textures (pytorch3d/renderer/mesh/textures.py:1190-1268, one map per mesh):
 * interpolate_face_attributes of the per-face uvs + torch.lerp to grid coordinates + F.grid_sample on K expanded
 * NCHW copies of the maps + permutes, and their autograd graph, with one kernel each way reading the maps in their
 * own (N, Hm, Wm, C) layout.  face_uvs (F,3,2) = verts_uvs[faces_uvs]; texels (N,H,W,K,C) fully written.
 * Backward: grad_bary (N,H,W,K,3) fully written; grad_face_uvs (F,3,2) and grad_maps (N,Hm,Wm,C) zeroed and
 * accumulated.  padding "reflection" is not provided. */
#define P3D_PAD_ZEROS 0
#define P3D_PAD_BORDER 1
#define P3D_SAMPLE_BILINEAR 0
#define P3D_SAMPLE_NEAREST 1
int p3d_sample_uv_forward(const int64_t* pix_to_face, const float* bary_coords, const float* face_uvs, const float* maps,
                          int N, int H, int W, int K, int64_t F, int Hm, int Wm, int C, int align_corners,
                          int padding_mode, int sampling_mode, float* texels, p3d_stream_t stream);
int p3d_sample_uv_backward(const float* grad_texels, const int64_t* pix_to_face, const float* bary_coords,
                           const float* face_uvs, const float* maps, int N, int H, int W, int K, int64_t F, int Hm, int Wm,
                           int C, int align_corners, int padding_mode, int sampling_mode, float* grad_bary_coords,
                           float* grad_face_uvs, float* grad_maps, p3d_stream_t stream);

/* replaces the maps_ids branch of TexturesUV.sample_textures (pytorch3d/renderer/mesh/textures.py:1270-1313, several
 * texture maps per mesh): maps (N,M,Hm,Wm,C) with M >= 2; maps_ids = the flattened maps_ids_padded (L entries), indexed
 * by the packed face index as the reference's gather does; background samples use face 0's map at uv = (0,0).  The map
 * index is the z coordinate of the reference's 3-D grid_sample: "bilinear" blends neighbouring maps wherever the
 * un-normalised z is not an integer (always with align_corners = 0) -- restated in csrc/uvm_sample.h.  Outputs as
 * p3d_sample_uv_forward / _backward (grad_maps (N,M,Hm,Wm,C)); faces >= L or >= F read as zero. */
int p3d_sample_uv_multi_forward(const int64_t* pix_to_face, const float* bary_coords, const float* face_uvs,
                                const float* maps, const int64_t* maps_ids, int64_t L, int N, int H, int W, int K, int64_t F,
                                int M, int Hm, int Wm, int C, int align_corners, int padding_mode, int sampling_mode,
                                float* texels, p3d_stream_t stream);
int p3d_sample_uv_multi_backward(const float* grad_texels, const int64_t* pix_to_face, const float* bary_coords,
                                 const float* face_uvs, const float* maps, const int64_t* maps_ids, int64_t L, int N, int H,
                                 int W, int K, int64_t F, int M, int Hm, int Wm, int C, int align_corners, int padding_mode,
                                 int sampling_mode, float* grad_bary_coords, float* grad_face_uvs, float* grad_maps,
                                 p3d_stream_t stream);

/* replaces TexturesAtlas.sample_textures (pytorch3d/renderer/mesh/textures.py:565-612): the nearest-cell lookup of a
 * per-face R x R atlas (F,R,R,C) by the first two barycentrics of each of the P = N*H*W*K samples -> texels (P,C), fully
 * written (zero for pix_to_face < 0).  Cell arithmetic: csrc/atlas_cell.h.  Indices the reference fails on (torch raises
 * IndexError: face >= F, cell outside [-R, R-1]) read as zero.  Backward: grad_atlas (F,R,R,C) zeroed and accumulated;
 * the barycentrics have no gradient (nearest sampling), as in the reference. */
int p3d_sample_atlas_forward(const int64_t* pix_to_face, const float* bary_coords, const float* atlas, int64_t P, int64_t F,
                             int R, int C, float* texels, p3d_stream_t stream);
int p3d_sample_atlas_backward(const float* grad_texels, const int64_t* pix_to_face, const float* bary_coords, int64_t P,
                              int64_t F, int R, int C, float* grad_atlas, p3d_stream_t stream);

/* replaces hard_rgb_blend (pytorch3d/renderer/blending.py:54-88: mask, masked_scatter of the background colour, cat with
 * the alpha channel): colors (npix,K,3), pix_to_face (npix,K) -> out (npix,4), 16-byte aligned: RGB of slot 0 and
 * alpha 1 where pix_to_face[...,0] >= 0, else the background colour and alpha 0.  Backward: grad_colors (npix,K,3)
 * fully written (slot 0 of covered pixels = grad_out[..., :3], zero elsewhere). */
int p3d_hard_rgb_blend_forward(const float* colors, const int64_t* pix_to_face, const float background[3], int64_t npix,
                               int K, float* out, p3d_stream_t stream);
int p3d_hard_rgb_blend_backward(const float* grad_out, const int64_t* pix_to_face, int64_t npix, int K, float* grad_colors,
                                p3d_stream_t stream);

/* ---- built-in per-kernel timing (HIP events on the launch stream) --------------------- */

/* enable != 0: every kernel launch is bracketed by hipEventRecord on its stream. */
void p3d_profile_enable(int enable);
/* Synchronise the recorded events and fold them into per-kernel totals. */
void p3d_profile_collect(void);
/* Number of distinct kernel names seen; name/launch count/total milliseconds of entry i. */
int p3d_profile_num_entries(void);
const char* p3d_profile_entry(int i, int64_t* launches, double* total_ms);
void p3d_profile_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* P3D_AMD_H_ */
