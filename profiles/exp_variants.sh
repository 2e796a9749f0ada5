#!/bin/bash
# Build-flag experiments of the fine rasterizer, prepared at the end of round 2 (no GPU time was left to measure them):
#   -DP3D_QUEUE_PAIRS=1   K = 4 / 8 queues in 64-bit register pairs, insertion by v_pk_mov_b32 under the lane mask
#                         (csrc/topk.h: TopKPairs; inner loop of the K = 8 kernel 278 -> 233 VALU instructions)
#   -DP3D_GEOM_PACKED=1   the (x, y) arithmetic of the per-(pixel, face) test on two-float vectors
#                         (csrc/p3d_geom.h: face_hit_rec_pk; 278 -> 251, both together 206)
#   -DP3D_QUEUE_PAIRS=2   as 1, and the perspective + clip kernels order queue entries by ONE unsigned 64-bit compare of
#                         (z bits, idx) instead of three 32-bit compares (their depths are never -0.0): 189 with packed
#   -DP3D_BWD_PACKED=1    the per-sample backward on two-float vectors, the shared terms of the barycentric gradient summed
#                         first (csrc/p3d_geom.h: face_sample_bwd_pk; K = 8 backward kernel 4147 -> 3549 VALU instructions;
#                         gradients are tolerance-gated, host build within 1e-5 of the scalar form)
#   -DP3D_CONCURRENT_FILL=2 -DP3D_FILL_LIMITER_KB=16|0   the concurrent background fill of profiles/r02_concurrent_fill.txt
#                         with a smaller / no LDS occupancy limiter (was 40 KB: 1.47 ms against 1.49)
# Both are bit-exact by construction; the host builds of the same code are checked in tests/test_cpu_abi_and_host.py.
#
# 1. HERE (hipcc cross-compiles):      bash profiles/exp_variants.sh build
# 2. on the GPU box through gpurun:    bash profiles/exp_variants.sh run     (writes gpurun_out/exp/*.json, *.txt)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
# Point rasterizer (csrc/raster_points.hip): -DP3D_POINT_QUEUE_PAIRS=1|2 -- K = 8, 10, 16, 32, 40, 50, 64, 100 as payload-free
# pair queues (2: one 64-bit key compare per entry).  Static counts of the whole K = 100 kernel: 5958 -> 2291 VALU
# instructions, 256 + 241 AGPRs -> 256 registers (two waves per SIMD instead of one); K = 50: 3073 -> 1441, 199 -> 156.
declare -A FLAGS=( [pairs]="-DP3D_QUEUE_PAIRS=1" [packed]="-DP3D_GEOM_PACKED=1" [both]="-DP3D_QUEUE_PAIRS=1 -DP3D_GEOM_PACKED=1" [key64]="-DP3D_QUEUE_PAIRS=2 -DP3D_GEOM_PACKED=1" [bwdpk]="-DP3D_BWD_PACKED=1" [all]="-DP3D_QUEUE_PAIRS=2 -DP3D_GEOM_PACKED=1 -DP3D_BWD_PACKED=1" [fill16]="-DP3D_CONCURRENT_FILL=2 -DP3D_FILL_LIMITER_KB=16" [fill0]="-DP3D_CONCURRENT_FILL=2 -DP3D_FILL_LIMITER_KB=0" [ppairs]="-DP3D_POINT_QUEUE_PAIRS=1" [pkey64]="-DP3D_POINT_QUEUE_PAIRS=2" )
case "${1:-}" in
  build)
    for v in pairs packed both key64 bwdpk all fill16 fill0 ppairs pkey64; do
      P3D_LIB_PATH=$ROOT/pytorch3d_amd/libp3d_$v.so P3D_EXTRA_FLAGS="${FLAGS[$v]}" python -m pytorch3d_amd.build > /dev/null || exit 1
      echo "built libp3d_$v.so (${FLAGS[$v]})"
    done ;;
  run)
    mkdir -p gpurun_out/exp
    B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs"
    $B > gpurun_out/exp/bench_product.json 2> gpurun_out/exp/bench_product.err
    for v in pairs packed both key64 bwdpk all fill16 fill0; do
      export P3D_LIB_PATH=$ROOT/pytorch3d_amd/libp3d_$v.so P3D_EXTRA_FLAGS="${FLAGS[$v]}"
      $B > gpurun_out/exp/bench_$v.json 2> gpurun_out/exp/bench_$v.err
      # parity of the variant: the mesh suite (oracle, fixtures, large images) and the device-vs-device comparison
      timeout 300 python -m pytest tests/test_gpu_meshes.py tests/test_gpu_vs_reference_device_kernels.py -x -q > gpurun_out/exp/tests_$v.txt 2>&1
      tail -2 gpurun_out/exp/tests_$v.txt
      unset P3D_LIB_PATH P3D_EXTRA_FLAGS
    done
    # point rasterizer variants: K sweep on the 1M-point config + the point / compositor suite
    python profiles/points_k_sweep.py 8 10 16 32 40 50 64 100 > gpurun_out/exp/points_product.txt 2>&1
    for v in ppairs pkey64; do
      export P3D_LIB_PATH=$ROOT/pytorch3d_amd/libp3d_$v.so P3D_EXTRA_FLAGS="${FLAGS[$v]}"
      python profiles/points_k_sweep.py 8 10 16 32 40 50 64 100 > gpurun_out/exp/points_$v.txt 2>&1
      timeout 300 python -m pytest tests/test_gpu_points_composite_interp.py tests/test_gpu_vs_reference_device_kernels.py -x -q > gpurun_out/exp/tests_$v.txt 2>&1
      tail -2 gpurun_out/exp/tests_$v.txt
      unset P3D_LIB_PATH P3D_EXTRA_FLAGS
    done
    tail -n 12 gpurun_out/exp/points_product.txt gpurun_out/exp/points_ppairs.txt gpurun_out/exp/points_pkey64.txt
    python - <<'PY'
import json
for t in ("product", "pairs", "packed", "both", "key64", "bwdpk", "all", "fill16", "fill0"):
    try:
        j = json.loads(open("gpurun_out/exp/bench_%s.json" % t).read().strip().splitlines()[-1])
        print(t, round(j["value"]), "Mpix/s", j["ms_per_step"], "ms/step", {k: v for k, v in j["kernels_ms"].items() if k.startswith("mesh")})
    except Exception as e:
        print(t, "ERR", e)
PY
    ;;
  *) echo "usage: $0 build|run"; exit 2 ;;
esac
