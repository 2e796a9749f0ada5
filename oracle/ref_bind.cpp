// ref_bind.cpp -- TEST INFRASTRUCTURE.  pybind glue that exposes the REFERENCE's own CPU
// implementation of the hot path (compiled from /root/reference/pytorch3d/csrc where it
// lies, without WITH_CUDA) under the names pytorch3d/csrc/ext.cpp:38-73 gives them.
// Built by oracle/build.py:build_ref() into oracle/_ref/p3d_ref_cpu.so.  It is the
// "real reference" the C restatement (p3d_oracle.c) and the HIP kernels are pinned
// against, and the "reference" kind of bench.py's cpu_baseline.
#include <torch/extension.h>

#include "utils/pytorch3d_cutils.h"  // sigmoid_alpha_blend.h relies on ext.cpp having included it
#include "blending/sigmoid_alpha_blend.h"
#include "compositing/alpha_composite.h"
#include "compositing/norm_weighted_sum.h"
#include "compositing/weighted_sum.h"
#include "face_areas_normals/face_areas_normals.h"
#include "packed_to_padded_tensor/packed_to_padded_tensor.h"
#include "rasterize_meshes/rasterize_meshes.h"
#include "rasterize_points/rasterize_points.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("rasterize_points", &RasterizePoints);
  m.def("rasterize_points_backward", &RasterizePointsBackward);
  m.def("rasterize_meshes_backward", &RasterizeMeshesBackward);
  m.def("rasterize_meshes", &RasterizeMeshes);
  m.def("accum_weightedsumnorm", &weightedSumNormForward);
  m.def("accum_weightedsum", &weightedSumForward);
  m.def("accum_alphacomposite", &alphaCompositeForward);
  m.def("accum_weightedsumnorm_backward", &weightedSumNormBackward);
  m.def("accum_weightedsum_backward", &weightedSumBackward);
  m.def("accum_alphacomposite_backward", &alphaCompositeBackward);
  m.def("sigmoid_alpha_blend", &SigmoidAlphaBlend);
  m.def("sigmoid_alpha_blend_backward", &SigmoidAlphaBlendBackward);
  m.def("_rasterize_points_coarse", &RasterizePointsCoarse);
  m.def("_rasterize_points_naive", &RasterizePointsNaive);
  m.def("_rasterize_meshes_naive", &RasterizeMeshesNaive);
  m.def("_rasterize_meshes_coarse", &RasterizeMeshesCoarse);
  m.def("_rasterize_meshes_fine", &RasterizeMeshesFine);
  // outside the hot path; used only by tests/run_reference_suite.py to keep the reference's own tests running
  m.def("face_areas_normals_forward", &FaceAreasNormalsForward);
  m.def("face_areas_normals_backward", &FaceAreasNormalsBackward);
  m.def("packed_to_padded", &PackedToPadded);
  m.def("padded_to_packed", &PaddedToPacked);
  // constants pytorch3d/renderer/points/pulsar/renderer.py reads at import time (ext.cpp:180-185)
  m.attr("EPS") = py::float_(1e-6);
  m.attr("MAX_FLOAT") = py::float_(3.4e38);
  m.attr("MAX_INT") = py::int_(2147483647);
  m.attr("MAX_UINT") = py::int_(4294967295u);
  m.attr("MAX_USHORT") = py::int_(65535);
  m.attr("PULSAR_MAX_GRAD_SPHERES") = py::int_(128);
}
