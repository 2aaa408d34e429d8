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
    sec = txt[txt.index("## 2."):txt.index("### 2b.")]
    blocks = re.findall(r"```cpp\n(.*?)```", sec, flags=re.S)
    assert len(blocks) >= 2, "the adaptor and the bias / registration block"
    return "\n".join(blocks)


# what the patch-based adaptor needs from the reference's headers -- declarations only: the host structs, the host-side
# accessors of PatchBasedObject<T> (include/patchBasedObject.cuh:84-139), ImagePatch2D<T> (include/ImagePatch2D.cuh:33-52), Volume<T> /
# ReconVolume<T> (include/volume.cuh, include/reconVolume.cuh:41-287), PointSpreadFunction<T> (include/pointSpreadFunction.cuh:35-42), the two
# classes (include/patchBasedSuperresolution_gpu.cuh:33-55, include/patchBasedRobustStatistics_gpu.cuh:33-66) and the three free functions
# (irtkPatchBasedReconstruction.cpp:51-56)
PREAMBLE_PVR = r"""
#include <stdint.h>
#include <vector>
struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct float3 { float x, y, z; };
template <typename T> struct real4 { T x, y, z, w; };
template <typename T> struct Matrix4 { real4<T> data[4]; };
template <typename T> class irtkGenericImage { public: const T* GetPointerToVoxels() const; };
template <typename T> class ImagePatch2D {
 public:
  Matrix4<T> Mo, InvMo, RI2W, I2W, W2I, Transformation, InvTransformation;
  T scale, patchWeight;
  char spxMask[64 * 64];
};
template <typename T> class Volume {
 public:
  virtual ~Volume() {}
  uint3 getSize() { return m_size; }
  float3 getDim() { return m_dim; }
  virtual void copyFromHost(const T* data);
  virtual void copyToHost(T* data) const;
  uint3 m_size;
  float3 m_dim;
  T* m_d_data;
};
template <typename T> class PatchBasedVolume : public Volume<T> {
 public:
  virtual uint3 getXYZPatchGridSize();
  std::vector<ImagePatch2D<T> > getHostImagePatch2DVector();
  std::vector<irtkGenericImage<T> > getHostImagePatchDataVector();
};
template <typename T> class PointSpreadFunction { public: float3 m_PSFdim; uint3 m_PSFsize; Matrix4<T> m_PSFI2W, m_PSFW2I; T m_quality_factor; };
template <typename T> class ReconVolume : public Volume<T> {
 public:
  void init(int cuda_device, uint3 s, float3 d, const Matrix4<float>& reconWorld2Image, const Matrix4<float>& reconImage2World);
  void release();
  void reset();
  void resetAddonCmap();
  void equalize();
  void setMask(char* mask_data);
  void copyFromHost(const T* data);
  void copyToHost(T* data) const;
  using Volume<T>::m_size;
  using Volume<T>::m_dim;
};
template <typename T> void patchBasedPSFReconstruction_gpu(int cuda_device, PatchBasedVolume<T>& inputStack, ReconVolume<T>& reconstruction, bool useSpx);
template <typename T> void patchBasedSimulatePatches_gpu(int cuda_device, PatchBasedVolume<T>& inputStack, ReconVolume<T>& reconstruction);
template <typename T> void initPatchBasedRecon_gpu(int cuda_device, PatchBasedVolume<T>& inputStack, ReconVolume<T>& reconstruction, PointSpreadFunction<float>& _PSF, bool useSpx);
template <typename T> class patchBasedSuperresolution_gpu {
 public:
  patchBasedSuperresolution_gpu(T _min_intensity, T _max_intensity, bool _adaptive = false);
  ~patchBasedSuperresolution_gpu();
  virtual void run(int _cuda_device, PatchBasedVolume<T>* _inputStack, ReconVolume<T>* _reconstruction);
  virtual void regularize(int rdevice, ReconVolume<T>* _reconstruction);
  virtual void updatePatchWeights();
 private:
  PatchBasedVolume<T>* m_inputStack;
  ReconVolume<T>* m_reconstruction;
  int m_cuda_device;
  T m_alpha, m_lambda, m_delta;
  bool m_adaptive;
  T m_min_intensity, m_max_intensity;
};
template <typename T> class patchBasedRobustStatistics_gpu {
 public:
  patchBasedRobustStatistics_gpu(std::vector<PatchBasedVolume<T> >& _inputStacks);
  ~patchBasedRobustStatistics_gpu();
  void updateInputStacks(std::vector<PatchBasedVolume<T> >& _inputStacks);
  void initializeEMValues();
  void InitializeRobustStatistics(T _min_intensity, T _max_intensity, int cuda_device = 0);
  void EStep();
  void MStep(int iter);
  void Scale();
};
"""


def _pvr_adaptor_text():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = txt[txt.index("### 2b."):txt.index("## 3.")]
    blocks = re.findall(r"```cpp\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1, "one block: the patch-based adaptor"
    return blocks[0]


def test_documented_patch_based_adaptor_compiles_against_the_c_abi(tmp_path):
    """initPatchBasedRecon_gpu / patchBasedPSFReconstruction_gpu / patchBasedSimulatePatches_gpu (PBR.cpp:51-56),
    patchBasedSuperresolution_gpu<T>::{run, regularize}, patchBasedRobustStatistics_gpu<T> and ReconVolume<T> on svr_* / pvrh_*: the text is
    compiled (g++ -c, so that the explicit specialisations are instantiated) against the real headers, and every svr_* / pvrh_* call in it is
    a declared symbol that libsvr_hip.so exports."""
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    code = _pvr_adaptor_text()
    for name in ("initPatchBasedRecon_gpu<float>", "patchBasedPSFReconstruction_gpu<float>", "patchBasedSimulatePatches_gpu<float>",
                 "patchBasedSuperresolution_gpu<float>::run", "patchBasedSuperresolution_gpu<float>::regularize", "patchBasedRobustStatistics_gpu<float>::EStep",
                 "patchBasedRobustStatistics_gpu<float>::MStep", "patchBasedRobustStatistics_gpu<float>::Scale", "ReconVolume<float>::init",
                 "ReconVolume<float>::setMask", "ReconVolume<float>::equalize", "ReconVolume<float>::copyToHost", "ReconVolume<float>::copyFromHost"):
        assert name in code, name
    for inc in ("patchBasedVolume.cuh", "reconVolume.cuh", "patchBasedSuperresolution_gpu.cuh", "patchBasedRobustStatistics_gpu.cuh"):
        code = code.replace(f'#include "{inc}"', "// (the reference's declarations: PREAMBLE_PVR of this test)")
    src = tmp_path / "pvr_hip_adaptor.cc"
    src.write_text(PREAMBLE_PVR + code)
    r = subprocess.run([cxx, "-std=c++11", "-c", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-o", str(tmp_path / "a.o"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    hdrs = open(os.path.join(ROOT, "include", "svr_hip.h")).read() + open(os.path.join(ROOT, "include", "svr_host.h")).read()
    called = sorted(set(re.findall(r"\b((?:svr|pvrh)_[a-z0-9_]+)\s*\(", code)))
    assert len(called) >= 20
    for name in called:
        assert re.search(r"\b" + name + r"\s*\(", hdrs), name
    lib = os.path.join(ROOT, "fetalreconstruction_amd", "lib", "libsvr_hip.so")
    if os.path.exists(lib):
        import ctypes
        L = ctypes.CDLL(lib)
        for name in called:
            assert hasattr(L, name), name


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
