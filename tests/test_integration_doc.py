"""INTEGRATION.md section 2 shows the reference-side binding a maintainer would add: `class Reconstruction`
(include/reconstruction_cuda2.cuh:92-341 of the reference) implemented on include/svr_hip.h.  This test compiles that text
(g++ -fsyntax-only) against the real header, with local uint3 / float3 / Matrix4 structs and a declaration of the class's
methods in place of the reference's header, so the documented binding cannot rot when the C-ABI changes."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what the adaptor needs from the reference's headers: the host structs of recon_volumeHelper.cuh:32-46 (layout-compatible
# with uint32_t[3] / float[3] / float[16]) and the declarations of the public methods the shims call
PREAMBLE = r"""
#include <stdint.h>
#include <vector>
struct uint3 { unsigned int x, y, z; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct Matrix4 { float4 data[4]; };
template <class T> struct Volume { uint3 size; float3 dim; T* data; };
class Reconstruction {
 public:
  Reconstruction(std::vector<int> dev, bool multiThreadedGPU);
  ~Reconstruction();
  void combineWeights(float* weights);
  void debugWeights(float* weights); void debugBias(float* bias); void debugSmoothMask(float* mask);
  void debugSimslices(float* s); void debugSimweights(float* s); void debugConfidenceMap(float* c); void debugAddon(float* a);
  void debugv_PSF_sums(float* p);
  void getVolWeights(float* weights);
  void updateStackSizes(std::vector<uint3> stack_sizes_);
  void InitializeEMValues();
  void syncCPU(float* reconstructed);
  void initStorageVolumes(uint3 size, float3 dim);
  void FillSlices(float* sdata, std::vector<int> sizesX, std::vector<int> sizesY);
  void generatePSFVolume(float* CPUPSF, uint3 PSFsize_, float3 sliceVoxelDim, float3 PSFdim, Matrix4 PSFI2W, Matrix4 PSFW2I, float _quality_factor);
  void setSliceDims(std::vector<float3> slice_dims, float quality_factor);
  void SetSliceMatrices(std::vector<Matrix4> matSliceTransforms, std::vector<Matrix4> matInvSliceTransforms, std::vector<Matrix4>& matsI2Winit,
                        std::vector<Matrix4>& matsW2Iinit, std::vector<Matrix4>& matsI2W, std::vector<Matrix4>& matsW2I, Matrix4 reconI2W, Matrix4 reconW2I);
  void UpdateSliceWeights(std::vector<float> slices_weights);
  void InitReconstructionVolume(uint3 s, float3 dim, float* data, float sigma_bias);
  void UpdateReconstructed(const uint3 vsize, float* data);
  void UpdateScaleVector(std::vector<float> scales, std::vector<float> slices_weights);
  void CalculateScaleVector(std::vector<float>& scale_vec);
  void setMask(uint3 s, float3 dim, float* data, float sigma_bias);
  void NormaliseBias(int iter, float sigma_bias);
  void EStep(float _m, float _sigma, float _mix, std::vector<float>& slice_potential);
  void MStep(int iter, float _step, float& _sigma, float& _mix, float& _m);
  void SimulateSlices(std::vector<bool>& slice_inside);
  void InitializeRobustStatistics(float& _sigma);
  void CorrectBias(float sigma_bias, bool _global_bias_correction);
  void Superresolution(int iter, std::vector<float> _slice_weight, bool _adaptive, float alpha, float _min_intensity, float _max_intensity,
                       float delta, float lambda, bool _global_bias_correction, float sigma_bias, float _low_intensity_cutoff);
  void maskVolume();
  void ScaleVolume();
  void RestoreSliceIntensities(std::vector<float> stack_factors_, std::vector<int> stack_index_);
  void GaussianReconstruction(std::vector<int>& voxel_num);
  void initRegStorageVolumes(uint3 size, float3 dim);
  void FillRegSlices(float* sdata, std::vector<Matrix4> slices_resampledI2W);
  void updateResampledSlicesI2W(std::vector<Matrix4> ofsSlice);
  void registerSlicesToVolume(std::vector<Matrix4>& transf);
  void prepareSliceToVolumeReg();
  std::vector<int> devicesToUse;
  std::vector<float> h_scales;
  Volume<float> regSlices;
};
"""


def _adaptor_text():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = txt[txt.index("## 2."):txt.index("## 3.")]
    blocks = re.findall(r"```cpp\n(.*?)```", sec, flags=re.S)
    assert len(blocks) >= 2, "the adaptor and the bias / registration block"
    return "\n".join(blocks)


def test_documented_adaptor_compiles_against_the_c_abi(tmp_path):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    code = _adaptor_text()
    assert "svr_update_stack_sizes" in code and "svr_combine_weights" in code and "svr_register_slices_to_volume" in code
    code = code.replace('#include "reconstruction_cuda2.cuh"', "// (the reference's class declaration: PREAMBLE of this test)")
    src = tmp_path / "reconstruction_hip_adaptor.cc"
    src.write_text(PREAMBLE + code)
    r = subprocess.run([cxx, "-std=c++11", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    # every svr_* call of the text is a symbol the header declares
    hdr = open(os.path.join(ROOT, "include", "svr_hip.h")).read()
    for name in sorted(set(re.findall(r"\b(svr_[a-z0-9_]+)\s*\(", code))):
        assert re.search(r"\b" + name + r"\s*\(", hdr), name
