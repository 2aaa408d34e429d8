/*
 * svr_hip.h -- C-ABI of the MI355X-native SVR super-resolution engine (libsvr_hip.so).
 *
 * Drop-in boundary for the reference's `class Reconstruction`
 * (source/reconstructionGPU2/include/reconstruction_cuda2.cuh:92-341, "RC.cuh" below; bodies in
 * reconstruction_cuda2.cu, "RC.cu"): one entry point per public method that
 * irtkReconstruction actually calls (irtkReconstructionGPU.cc, "RG.cc"), with the C++ types
 * flattened to plain pointers and sizes:
 *     uint3 / float3            -> const uint32_t[3] / const float[3]
 *     Matrix4 (row-major 4x4)   -> const float[16]           (recon_volumeHelper.cuh:41-46)
 *     std::vector<Matrix4>      -> const float* (n*16)
 *     std::vector<float|int>    -> const float* / const int* (length = number of slices)
 *     std::vector<bool>&        -> uint8_t*
 * Every call is blocking and takes/returns HOST memory exactly like the reference (which
 * copies synchronously); one call in flight per context.  Instead of the reference's
 * print-and-exit (RC.cuh:79-83) every function returns 0 on success or a non-zero status
 * (a hipError_t value, or SVR_E_*), and svr_last_error() returns the message.
 *
 * One context drives ONE GPU.  Multi-GPU follows the one-process-per-GPU model: each rank
 * creates a context for its shard of the slices and the caller all-reduces the volume
 * accumulators between the *_local and *_finish halves (section "sharded operation").
 */
#ifndef SVR_HIP_H
#define SVR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svr_ctx svr_ctx;

#define SVR_OK 0
#define SVR_E_ARG 10001     /* bad argument / call order */
#define SVR_E_STATE 10002   /* required state (volume, slices, matrices...) not set */

/* ---- lifecycle --------------------------------------------------------------------- */
/* Reconstruction::Reconstruction(std::vector<int> dev, bool multiThreadedGPU)  RC.cuh:94, RC.cu:616-706 */
int svr_create(int device, svr_ctx **out);
/* Reconstruction::~Reconstruction  RC.cuh:95, RC.cu:1232-1273 */
void svr_destroy(svr_ctx *ctx);
const char *svr_last_error(const svr_ctx *ctx);
/* flags  _disableBiasC / _debugGPU  (RC.cuh public members; RG.cc:227-238).  The default is the
 * reference CLI's: bias correction disabled (reconstruction.cc:121,202).  Call with
 * disable_bias_correction = 0 before the first compute call to enable the bias path. */
int svr_set_flags(svr_ctx *ctx, int disable_bias_correction, int debug_gpu);
/* engine options (no reference equivalent).
 * "coeff_table" (default 0): 1 = keep the evaluated PSF taps of every live (pixel, plane) unit in HBM, written once per
 *   slice geometry, and stream them in the scatter and the gather instead of evaluating them in every SR iteration --
 *   irtkReconstruction::CoeffInit's _volcoeffs (irtkReconstructionGPU.cc:2305-2673) on the GPU path; 16 KiB per PSF
 *   pixel; switches itself off when that does not fit the free memory; results are those of the on-the-fly kernels.
 * "back_mode": 5 = cell-owned planes, staged and combined in a fixed order, no float atomics (default; csrc/svr_cell.inc),
 *   4 = wave-owned LDS planes flushed with float atomics (the fallback when the cell lists cannot hold a geometry), 3 = the
 *   workgroup kernel for every tile, 1 = LDS tiles with ds_add_f32, 0 = direct device-scope atomics per tap.  "fwd_mode": 2 = the
 *   gather over the same (cell, plane) items (default), 1 = unit-based LDS gather per slice tile, 0 = wave-per-pixel kernel.
 *   "gauss_mode", "pvr_mode": the same choice for the Gaussian pass / the patch-to-volume kernels (tests).
 * "cell_w"/"cell_h", "cell_gw"/"cell_gh" (0 = from the pixel density), "cell_band": cell sizes of the scatter / the gather.
 * "cell_order" (default 1): the (cell, plane) items of a launch go out in order of falling work, in classes of 2^(value - 1)
 *   pixels; 0 = (cell, plane) order.  "cell_balance" (default 0 = never): an item heavier than the launch's work /
 *   (1024 x value) is cut into parts, each a workgroup of its own; "cell_split": at least that many parts per item.
 * "tile_w"/"tile_h", "wave_cap", "fwd_tile_w"/"fwd_tile_h", "fwd_unit_cap": tile shapes and LDS box sizes of the tile kernels.
 * "fwd_autotune" (default 0 since round 4: the shapes follow from the geometry, every run and every rank picks the same):
 *   1 = the first forward projection / back-projection after new slice geometry times the
 *   candidate shapes and box sizes on the data and keeps the fastest; an explicit shape switches that off.
 *   Long tile lists are timed on runs of consecutive tiles (one run out of every stride, about 131072 tiles per trial;
 *   "tune_tiles" sets another number); environment: SVR_TUNE_TILES (tiles per trial, 0 = always the whole list), SVR_TUNE_RUN,
 *   SVR_TUNE_DEBUG=1 (the candidates' times on stderr).
 * "pvr": 1 selects the patch-to-volume constants and kernels.
 * Other environment variables: SVR_COEFF_MAX_GB (ceiling of the coefficient table), SVR_FWD_PIECE (test hook: tiles per
 *   dispatch of the gather, at most 2^22). */
int svr_set_option(svr_ctx *ctx, const char *name, int value);
/* the current value of an option: what the tile-shape / box-size timing chose, or whether "coeff_table" stayed on (it
 * switches itself off when the table -- 16 KiB per PSF pixel -- does not fit the free device memory) */
int svr_get_option(svr_ctx *ctx, const char *name, int *value);
/* "pvr" = 1 switches the PSF kernels to the patch-to-volume constants of
 * PVRreconstructionGPU (patchBasedPSFReconstruction_gpu.cu, patchBasedSimulatePatches_gpu.cu,
 * patchBasedSuperresolution_gpu.cu): patches are handed over as the "slices" of the padded grid,
 * patch.scale / patch.patchWeight as the scale / slice-weight vectors.  Optional superpixel masks:
 * ImagePatch2D::spxMask, n_patches * 64*64 chars of '0'/'1' (NULL clears them). */
int svr_set_spx_masks(svr_ctx *ctx, const char *masks_or_null);

/* ---- geometry / state upload -------------------------------------------------------- */
/* InitReconstructionVolume(uint3 s, float3 dim, float* data, float sigma_bias)  RC.cuh:230, RC.cu:1160-1230 */
int svr_init_reconstruction_volume(svr_ctx *ctx, const uint32_t size[3], const float dim[3],
                                   const float *data_or_null, float sigma_bias);
/* setMask(uint3 s, float3 dim, float* data, float sigma_bias)  RC.cuh:240, RC.cu:1095-1158 */
int svr_set_mask(svr_ctx *ctx, const uint32_t size[3], const float dim[3], const float *data,
                 float sigma_bias);
/* initStorageVolumes(uint3 size, float3 dim): size = (maxX, maxY, nSlices)  RC.cuh:214, RC.cu:1408-1573 */
int svr_init_storage_volumes(svr_ctx *ctx, const uint32_t size[3], const float dim[3]);
/* FillSlices(float* sdata, std::vector<int> sizesX, std::vector<int> sizesY)  RC.cuh:216, RC.cu:1574-1656 */
int svr_fill_slices(svr_ctx *ctx, const float *sdata, const int *sizes_x, const int *sizes_y);
/* setSliceDims(std::vector<float3> slice_dims, float quality_factor)  RC.cuh:220, RC.cu:772-833 */
int svr_set_slice_dims(svr_ctx *ctx, const float *slice_dims /* n*3 */, float quality_factor);
/* SetSliceMatrices(matSliceTransforms, matInvSliceTransforms, matsI2Winit, matsW2Iinit, matsI2W,
 *                  matsW2I, reconI2W, reconW2I)  RC.cuh:222-224, RC.cu:835-907 */
int svr_set_slice_matrices(svr_ctx *ctx, const float *slice_transforms, const float *inv_slice_transforms,
                           const float *i2w_init, const float *w2i_init, const float *i2w,
                           const float *w2i, const float recon_i2w[16], const float recon_w2i[16]);
/* generatePSFVolume(float* CPUPSF, uint3 PSFsize, float3 sliceVoxelDim, float3 PSFdim,
 *                   Matrix4 PSFI2W, Matrix4 PSFW2I, float quality_factor)  RC.cuh:218, RC.cu:718-750
 * (the reference uploads only the constants; CPUPSF may be NULL) */
int svr_generate_psf_volume(svr_ctx *ctx, const float *cpu_psf_or_null, const uint32_t psf_size[3],
                            const float slice_voxel_dim[3], const float psf_dim[3],
                            const float psf_i2w[16], const float psf_w2i[16], float quality_factor);
/* UpdateScaleVector(std::vector<float> scales, std::vector<float> slices_weights)  RC.cuh:236, RC.cu:1342-1393 */
int svr_update_scale_vector(svr_ctx *ctx, const float *scales, const float *slice_weights);
/* UpdateSliceWeights(std::vector<float>)  RC.cuh:228, RC.cu:1312-1340 */
int svr_update_slice_weights(svr_ctx *ctx, const float *slice_weights);
/* UpdateReconstructed(const uint3 vsize, float* data)  RC.cuh:233, RC.cu:1658-1675 */
int svr_update_reconstructed(svr_ctx *ctx, const uint32_t size[3], const float *data);
/* syncCPU(float* reconstructed)  RC.cuh:213, RC.cu:2031-2037 */
int svr_sync_cpu(svr_ctx *ctx, float *reconstructed);
/* getVolWeights(float*)  RC.cuh:206 */
int svr_get_vol_weights(svr_ctx *ctx, float *weights);
/* Reconstruction::combineWeights(float*) RC.cuh:190, RC.cu:5091-5097: the bias path's volume weights of device 0 */
int svr_combine_weights(svr_ctx *ctx, float *out);
/* Reconstruction::updateStackSizes(std::vector<uint3>) RC.cuh:210: stored (the reference never reads them back); sizes3 = n_stacks x {x, y, z} */
int svr_update_stack_sizes(svr_ctx *ctx, const uint32_t *sizes3, int n_stacks);

/* ---- compute ------------------------------------------------------------------------ */
/* GaussianReconstruction(std::vector<int>& voxel_num): voxel_num[0] = #pixels that hit the ROI
 * RC.cuh:274, RC.cu:2329-2493 */
int svr_gaussian_reconstruction(svr_ctx *ctx, int *voxel_num);
/* SimulateSlices(std::vector<bool>& slice_inside)  RC.cuh:257, RC.cu:2654-2763.  slice_inside may be NULL: the flags stay
 * on the device and the call does not wait for it (svr_get_slice_inside / svr_mstep_estep) */
int svr_simulate_slices(svr_ctx *ctx, uint8_t *slice_inside);
/* InitializeEMValues()  RC.cuh:211, RC.cu:3241-3310 */
int svr_initialize_em_values(svr_ctx *ctx);
/* InitializeRobustStatistics(float& sigma)  RC.cuh:260, RC.cu:2243-2308 */
int svr_initialize_robust_statistics(svr_ctx *ctx, float *sigma);
/* EStep(float m, float sigma, float mix, std::vector<float>& slice_potential)  RC.cuh:253, RC.cu:2766-2925 */
int svr_estep(svr_ctx *ctx, float m, float sigma, float mix, float *slice_potential);
/* MStep(int iter, float step, float& sigma, float& mix, float& m)  RC.cuh:255, RC.cu:2927-3112 */
int svr_mstep(svr_ctx *ctx, int iter, float step, float *sigma, float *mix, float *m);
/* CalculateScaleVector(std::vector<float>& scale_vec)  RC.cuh:238, RC.cu:3114-3239.  scale_vec may be NULL: the vector stays
 * on the device (it is what the next kernels read anyway) and the call does not wait; svr_get_scale_vector or
 * svr_mstep_estep hand it over later. */
int svr_calculate_scale_vector(svr_ctx *ctx, float *scale_vec);
int svr_get_scale_vector(svr_ctx *ctx, float *scale_vec);
/* the scale vector last calculated becomes the one the kernels see: the patch-based loop's copyToScales
 * (patchBasedRobustStatistics_gpu.cu:672-745) without a trip through the host */
int svr_adopt_scale_vector(svr_ctx *ctx);
/* slice_inside of the last SimulateSlices when that was called with slice_inside == NULL (no wait for the device there) */
int svr_get_slice_inside(svr_ctx *ctx, uint8_t *slice_inside);
/* MStep then EStep (reconstruction.cc:1093-1108, irtkPatchBasedReconstruction.cpp:540-545) with ONE wait for the device instead
 * of two: the M-step's scalars (Reconstruction::MStep's host part RC.cu:3016-3071; for a patch-based context
 * patchBasedRobustStatistics_gpu.cu:570-640) are worked out on the device with the same float operations.
 * em3 = {sigma, mix, m} in/out; scale_vec / slice_inside: NULL or what the deferred calls above left on the device. */
int svr_mstep_estep(svr_ctx *ctx, int iter, float step, float em3[3], float *slice_potential, float *scale_vec, uint8_t *slice_inside);
/* the M-step's five sums (svr_mstep_sums: what a sharded host exchanges) together with the deferred vectors, one wait */
int svr_mstep_sums_fetch(svr_ctx *ctx, double out5[5], float *scale_vec, uint8_t *slice_inside);
/* ... and without the wait in between (round 4): svr_mstep_partial runs the M-step's sums of this rank's slices and names two device
 * buffers -- *send: its 8 doubles (16 floats), *recv: world x 8 doubles -- for the launcher's all-gather on the engine's stream;
 * svr_mstep_estep_ranks then adds the ranks' sums up on the device in rank order (what the hosts do after their exchange), works the
 * scalars out there, runs the E-step and fetches potentials, {sigma, mix, m} and the deferred vectors in one wait: svr_mstep_estep of
 * a sharded run.  The reference keeps all of this on one GPU (reconstruction_cuda2.cu:3016-3071, 2766-2960). */
int svr_mstep_partial(svr_ctx *ctx, int world, void **send, void **recv);
int svr_mstep_estep_ranks(svr_ctx *ctx, int world, int iter, float step, float em3[3], float *slice_potential, float *scale_vec, uint8_t *slice_inside);
/* Superresolution(int iter, std::vector<float> slice_weight, bool adaptive, float alpha, float min_intensity,
 *                 float max_intensity, float delta, float lambda, bool global_bias_correction,
 *                 float sigma_bias, float low_intensity_cutoff)  RC.cuh:263-265, RC.cu:2119-2241 */
int svr_superresolution(svr_ctx *ctx, int iter, const float *slice_weight, int adaptive, float alpha,
                        float min_intensity, float max_intensity, float delta, float lambda,
                        int global_bias_correction, float sigma_bias, float low_intensity_cutoff);
/* CorrectBias(float sigma_bias, bool global_bias_correction)  RC.cuh:261, RC.cu:1837-1942 */
int svr_correct_bias(svr_ctx *ctx, float sigma_bias, int global_bias_correction);
/* NormaliseBias(int iter, float sigma_bias)  RC.cuh:251, RC.cu:2519-2652 */
int svr_normalise_bias(svr_ctx *ctx, int iter, float sigma_bias);
/* maskVolume()  RC.cuh:270, RC.cu:3313-3347 */
int svr_mask_volume(svr_ctx *ctx);
/* ScaleVolume()  RC.cuh:271, RC.cu:3386-3470 */
int svr_scale_volume(svr_ctx *ctx);
/* RestoreSliceIntensities(std::vector<float> stack_factors, std::vector<int> stack_index)  RC.cuh:272, RC.cu:3349-3383 */
int svr_restore_slice_intensities(svr_ctx *ctx, const float *stack_factors, int n_stacks,
                                  const int *stack_index);

/* ---- debug getters (RC.cuh:192-207): copy a device buffer to host --------------------- */
enum svr_buffer {
  SVR_BUF_RECONSTRUCTED = 0, /* volume-shaped float */
  SVR_BUF_VOL_WEIGHTS = 1,
  SVR_BUF_ADDON = 2,         /* debugAddon */
  SVR_BUF_CONFIDENCE_MAP = 3,/* debugConfidenceMap */
  SVR_BUF_MASK = 4,
  SVR_BUF_BIAS_VOLUME = 5,   /* debugNormalizeBias */
  SVR_BUF_SMOOTH_MASK = 6,   /* debugSmoothMask */
  SVR_BUF_SLICES = 10,       /* slice-grid-shaped float */
  SVR_BUF_WEIGHTS = 11,      /* debugWeights */
  SVR_BUF_SIMSLICES = 12,    /* debugSimslices */
  SVR_BUF_SIMWEIGHTS = 13,   /* debugSimweights */
  SVR_BUF_PSF_SUMS = 14,     /* debugv_PSF_sums */
  SVR_BUF_BIAS = 15,         /* debugBias */
  SVR_BUF_SIMINSIDE = 20,    /* slice-grid-shaped char, debugSiminside */
  SVR_BUF_VOXEL_COUNT = 21   /* slice-grid-shaped int (sliceVoxel_count_) */
};
int svr_debug_get(svr_ctx *ctx, int which, void *host_out, size_t bytes);
/* test/driver hook: overwrite a device buffer from host (same enum) */
int svr_debug_set(svr_ctx *ctx, int which, const void *host_in, size_t bytes);
/* test probe: the 16^3 PSF taps of one slice pixel as the kernels evaluate them
 * (vals[x + 16*y + 256*z] = PSF value, or -1 where the epsilon-skip of RC.cu:238 drops the tap)
 * and the rounded centre voxel (RC.cu:225-226) */
int svr_debug_probe_pixel(svr_ctx *ctx, int slice, int px, int py, float *vals4096, int *centre3);

/* ---- sharded operation (no reference equivalent: replaces GPUWorker.cpp + the peer-copy
 * reductions RC.cu:2225-2239,2445-2460 with "local half / caller all-reduce / finish half") */
/* device pointer of a buffer; ADDON is followed contiguously by CONFIDENCE_MAP, and
 * RECONSTRUCTED by VOL_WEIGHTS, so each pair all-reduces as one float[2*Nvox] message */
void *svr_device_ptr(svr_ctx *ctx, int which);
/* Sharded runs: the scatter only ever writes voxels of the mask, so the ranks need not exchange the rest of a volume pair.
 * svr_pair_pack copies the mask's bounding box of the n_floats / Nv volumes at svr_device_ptr(ctx, which) into a contiguous device
 * buffer (*packed, *n_packed floats) on the engine's stream; the caller all-reduces that buffer in place and calls svr_pair_unpack.
 * *packed == NULL: nothing to gain (no mask box, or the box is more than 80 % of the volume) -- reduce the whole buffer. */
int svr_pair_pack(svr_ctx *ctx, int which, size_t n_floats, void **packed, size_t *n_packed);
int svr_pair_unpack(svr_ctx *ctx, int which, size_t n_floats);
size_t svr_volume_voxels(const svr_ctx *ctx);
/* run all kernels on this hipStream_t (e.g. the caller's torch stream); NULL = default */
int svr_set_stream(svr_ctx *ctx, void *hip_stream);
/* the hipStream_t every kernel of the context is enqueued on, and the context's device (for collectives that must be ordered
 * with the kernels: csrc/svr_rccl.cpp) */
void *svr_get_stream(svr_ctx *ctx);
int svr_stream_sync(svr_ctx *ctx);   /* wait for everything queued on the engine's stream */
int svr_device(svr_ctx *ctx);
/* number of HIP devices visible to this process (hipGetDeviceCount; 0 on error): what `-d` / --gpus can name */
int svr_device_count(void);
int svr_gaussian_reconstruction_local(svr_ctx *ctx);            /* scatter into recon|volw */
int svr_gaussian_reconstruction_finish(svr_ctx *ctx, int *voxel_num_local); /* equalize */
int svr_superresolution_backproject(svr_ctx *ctx, const float *slice_weight); /* addon|cmap */
int svr_superresolution_update(svr_ctx *ctx, int adaptive, float alpha, float min_intensity,
                               float max_intensity, float delta, float lambda);
/* The volume update of a SHARDED run by z-slabs (csrc/svr_slab.inc) instead of all-reduce + svr_superresolution_update on every
 * rank -- replaces the reduce on GPU 0 + AdaptiveRegularization there, reconstruction_cuda2.cu:2225-2239, 2138-2181:
 *   svr_slab_plan(world, rank)   once per (mask, world): slab boundaries, index lists; chunk sizes in floats PER RANK
 *   svr_slab_rs_pack             after svr_superresolution_backproject: *send = float[world][rs_floats_per_rank] (addon | cmap at the
 *                                mask's voxels of every rank's slab + halo planes), *recv = float[rs_floats_per_rank]
 *   -- the caller reduce-scatters (sum) send -> recv over the ranks, on the engine's stream or ordered behind it --
 *   svr_slab_update              the sums go into addon | cmap, the rank's planes are updated, *send = float[ag_floats_per_rank] (its part
 *                                of the new volume), *recv = float[world][ag_floats_per_rank]
 *   -- the caller all-gathers send -> recv --
 *   svr_slab_finish              every rank's part goes into the new volume, which becomes the reconstructed volume.
 * Same bits as the replicated update.  Needs option reg_mode 1 (the default) and a mask set through svr_set_mask. */
int svr_slab_plan(svr_ctx *ctx, int world, int rank, size_t *rs_floats_per_rank, size_t *ag_floats_per_rank);
int svr_slab_rs_pack(svr_ctx *ctx, void **send, void **recv);
int svr_slab_update(svr_ctx *ctx, int adaptive, float alpha, float min_intensity, float max_intensity, float delta, float lambda,
                    void **send, void **recv);
int svr_slab_finish(svr_ctx *ctx);
/* NormaliseBias halves: scatter into SVR_BUF_BIAS_VOLUME (all-reduce it), then normalise + apply */
int svr_normalise_bias_local(svr_ctx *ctx);
int svr_normalise_bias_finish(svr_ctx *ctx, float sigma_bias);
/* partial sums for the caller to all-reduce: {sum (s-sim)^2, count} */
int svr_robust_statistics_sums(svr_ctx *ctx, double out2[2]);
/* {sum e^2 w, sum w, count, min e, max e} (min/max reduce with min/max) */
int svr_mstep_sums(svr_ctx *ctx, double out5[5]);
/* {sum w sw s sim, sum w sw sim^2} of ScaleVolume */
int svr_scale_volume_sums(svr_ctx *ctx, double out2[2]);
int svr_scale_volume_apply(svr_ctx *ctx, float scale);

/* ---- slice-to-volume registration cost -------------------------------------
 * The metric of the reference's DEFAULT registration path: irtkImageRigidRegistrationWithPadding::
 * Evaluate (IRTKSimple2/packages/registration/src/irtkImageRigidRegistrationWithPadding.cc:534-610)
 * with irtkCrossCorrelationSimilarityMetric (.../include/irtkCrossCorrelationSimilarityMetric.h:66-165),
 * batched over (target slice, candidate transform) pairs so an optimiser step is one call.
 * targets: n resampled slices, short [n][ty][tx], padding < 0 (irtkReconstructionGPU.cc:2008-2016).
 * source: the volume as short; NULL = static_cast<short> of the current reconstruction (RG.cc:2031).
 * matrices: per evaluation a row-major double 4x4 = sourceW2I * T * targetI2W
 * (irtkHomogeneousTransformationIterator.h:96).  sums6 = {n, sum t, sum s, sum t^2, sum s^2, sum t*s}. */
int svr_ncc_set_targets(svr_ctx *ctx, int n, int tx, int ty, const int16_t *targets);
/* The image pyramid of the same registration on the device (irtkImageRegistrationWithPadding::Initialize(level),
 * irtkImageRegistrationWithPadding.cc:28-334, for one image: irtkGaussianBlurringWithPadding<short> -- three 1-D passes, each
 * truncated back to short --, irtkResamplingWithPadding<short>, the shift of the range to start at 0 and padding -1), in the
 * reference's double arithmetic operation for operation, so that the levels are the ones the host code of csrc/irtk_reg.cpp
 * makes.  svr_pyr_upload keeps the unprocessed images of a registration pass on the device (slot 0: the source volume,
 * slot 1: the targets, images of one grid back to back); svr_pyr_level makes one level of `n_images` images of one grid
 * starting at element `offset` of the slot and leaves them where svr_ncc_evaluate reads them: slot 0 -> the NCC source
 * (n_images must be 1), slot 1 -> the target planes from `first_plane` on (svr_ncc_alloc_targets first).
 * kernel_x/y/z: the sampled Gaussian of each pass (size 0 = no pass on that axis); resample 0 = the level keeps the grid.
 * min_out / max_out[n_images]: range of the values above the padding before the shift (max < min: none). */
typedef struct svr_pyr_image {
  double i2w_out[12];   /* rows 0..2 of the image-to-world matrix of the level's grid */
  double w2i_in[12];    /* rows 0..2 of the world-to-image matrix of the unprocessed grid */
  int pad;              /* the image's padding value */
} svr_pyr_image;
int svr_ncc_alloc_targets(svr_ctx *ctx, int n, int tx, int ty);
int svr_pyr_upload(svr_ctx *ctx, int slot, const int16_t *images, size_t count);
int svr_pyr_level(svr_ctx *ctx, int slot, size_t offset, int n_images, const int in_dims[3], const double *kernel_x, int size_x,
                  const double *kernel_y, int size_y, const double *kernel_z, int size_z, int resample, const int out_dims[3],
                  const svr_pyr_image *images, int first_plane, int *min_out, int *max_out);
int svr_ncc_set_source(svr_ctx *ctx, const uint32_t size[3], const int16_t *source_or_null);
int svr_ncc_evaluate(svr_ctx *ctx, int n_eval, const int *target_index, const double *matrices,
                     int64_t *sums6_or_null, double *ncc_or_null);

/* ---- GPU slice-to-volume registration (the reference's --useGPUReg path) -------
 * One entry point per public method of `class Reconstruction` used by irtkReconstruction::
 * PrepareRegistrationSlices / SliceToVolumeRegistrationGPU (irtkReconstructionGPU.cc:2104-2290).
 * Matrices are row-major float[16] (Matrix4), one per slice of this context.
 *
 * initRegStorageVolumes(uint3 size, float3 dim)                    RC.cuh:326, RC.cu:4893-5021
 *   size = {x, y, slices} of the slices resampled to the reconstruction's voxel size. */
int svr_init_reg_storage_volumes(svr_ctx *ctx, const uint32_t size[3], const float dim[3]);
/* FillRegSlices(float* sdata, vector<Matrix4> slices_resampledI2W)  RC.cuh:328, RC.cu:5023-5088
 *   sdata [slices][y][x], padding -1.  The I2W matrices are accepted for signature parity; no
 *   reference kernel reads them (RC.cu:3515). */
int svr_fill_reg_slices(svr_ctx *ctx, const float *sdata, const float *slices_resampledI2W_or_null);
/* updateResampledSlicesI2W(vector<Matrix4> ofsSlice)                RC.cuh:331, RC.cu:4707-4757 */
int svr_update_resampled_slices_i2w(svr_ctx *ctx, const float *ofsSlice);
/* prepareSliceToVolumeReg()                                         RC.cuh:338, RC.cu:3800-3959
 *   snapshots the current reconstruction as the sampled volume, sets 2 levels x 4 step sizes x
 *   <= 20 iterations, epsilon 1e-4, blurring = voxel/2 and voxel, step lengths 0.1 and 0.2. */
int svr_prepare_slice_to_volume_reg(svr_ctx *ctx);
/* registerSlicesToVolume(vector<Matrix4>& transf_)                  RC.cuh:333, RC.cu:4760-4867
 *   -> registerMultipleSlicesToVolume RC.cu:4001-4141; transf [slices][16] in/out. */
int svr_register_slices_to_volume(svr_ctx *ctx, float *transf);
/* test / tuning hooks (no reference counterpart): shorter schedule (0 = keep), one cost evaluation,
 * counters {cost evaluations, line-search steps, outer iterations, slice evaluations} of the last run */
int svr_reg_set_schedule(svr_ctx *ctx, int levels, int steps, int iterations);
int svr_reg_evaluate_costs(svr_ctx *ctx, const float *transf, int level, const int *active_or_null, int n_active,
                           float *similarities_out, float *reg_slices_out_or_null);
int svr_reg_counters(svr_ctx *ctx, long long out4[4]);

/* PVR patch-to-volume registration cost (second variant): computeCCpatch
 * (patchBased2D3DRegistration_gpu2.cu:130-190) for every patch of the slice grid with its own
 * candidate matrix: raw-moment NCC against the current reconstruction through the software trilinear
 * interpolation of include/interpFunctions.cuh:81-96, offsets z = -1, 0, 1, every (level+1)-th pixel.
 * buffer: the (blurred) patches [n][pY][pX] (`getBufferValue`), NULL = the uploaded patches;
 * RI2W[i] = patch.RI2W, Tmats[i] = the candidate transformation * Mo (patchBasedObject.cuh:297-304).
 * sums6 = {n, sum a, sum b, sum a^2, sum b^2, sum a*b} (double). */
int svr_pvr_cc_patches(svr_ctx *ctx, const float *buffer_or_null, const float *RI2W, const float *Tmats, int level,
                       float *ncc_out, double *sums6_or_null);
/* PatchBased2D3DRegistration_gpu2<T>::run (patchBased2D3DRegistration_gpu2.cu:450-566): the patch-to-volume registration of
 * the patch-based path, every patch of the slice grid against the current reconstruction; parallelPatchRegOptimization
 * (:198-291) with a workgroup per patch instead of a thread.  T [n][16] in/out = the patches' transformations, Tinv_out their
 * inverses; Mo / InvMo / RI2W = the origin-reset matrices of the patches (patchBasedObject.cuh:285-304).
 * counters3 = {kernel launches, cost evaluations, patches}. */
int svr_pvr_register_patches(svr_ctx *ctx, const float *RI2W, const float *Mo, const float *InvMo, float *T, float *Tinv_out,
                             long long *counters3_or_null);

/* ---- measurement -------------------------------------------------------------------- */
enum svr_timer {
  SVR_T_BACKPROJECT = 0, SVR_T_FORWARD = 1, SVR_T_GAUSS = 2, SVR_T_REGULARIZE = 3,
  SVR_T_ESTEP = 4, SVR_T_MSTEP = 5, SVR_T_SCALE = 6, SVR_T_REGISTER = 7,
  /* sharded runs, filled by the host objects (csrc/svr_host.cpp, csrc/pvr_host.cpp) through svr_timer_begin / _end / _add:
   * the all-reduce of a volume pair (HIP events on the engine's stream around the collective) and the small host-side
   * exchanges (wall clock: a stream synchronisation plus the collective) */
  SVR_T_ALLREDUCE = 8, SVR_T_EXCHANGE = 9,
  SVR_T_COEFF_BUILD = 10,   /* k_coeff_build: writing the coefficient table (option coeff_table), once per slice geometry */
  SVR_T_REDUCE_SCATTER = 11, SVR_T_ALLGATHER = 12,   /* the two collectives of the slab update (svr_slab_*), HIP events like SVR_T_ALLREDUCE */
  /* round 6: SVR_T_BACKPROJECT / SVR_T_FORWARD by the kind of pass (the same intervals, counted a second time): the passes that stream the
   * coefficient table, and the gather / the scatter that evaluates and writes it (coeff_lazy: whichever PSF pass comes first after a new geometry) */
  SVR_T_BACKPROJECT_TABLE = 13, SVR_T_FORWARD_TABLE = 14, SVR_T_FORWARD_STORE = 15, SVR_T_BACKPROJECT_STORE = 16,
  SVR_T_COUNT = 17
};
/* HIP events on the engine's stream around work a caller enqueues there itself (the volume all-reduce); no-ops while the
 * timers are off.  svr_timer_end waits for the stream. */
int svr_timer_begin(svr_ctx *ctx, int which);
int svr_timer_end(svr_ctx *ctx, int which);
/* adds host-measured milliseconds to a timer (counted as one launch) */
int svr_timer_add(svr_ctx *ctx, int which, double ms);
/* accumulated HIP-event time (ms) and launch count of a hot kernel since the last reset */
int svr_timer_get(svr_ctx *ctx, int which, double *ms_total, long *launches);
int svr_timer_reset(svr_ctx *ctx);
int svr_timer_enable(svr_ctx *ctx, int enable);
/* workload counters: [0] pixels in slice grid (Vs), [1] active pixels s!=-1,
 * [2] pixels with v_PSF_sums!=0 (Va), [3] volume voxels (Nv), [4] slices,
 * [5] pixel tiles of the scatter, [6] tiles that took the atomic fallback in the last scatter, [7] tiles the 5-wave scatter
 * instance handed to the 8-wave one */
int svr_counters(svr_ctx *ctx, uint64_t out8[8]);
/* (pixel, plane) units of the pixels with s != -1 whose footprint reaches the volume: [0] such pixels, [1] live units (every tap
 * evaluated), [2] dead units (every row provably below the epsilon of RC.cu:238: only its first tap is processed) -- what
 * bench.py's `flops_executed` counts */
int svr_unit_counts(svr_ctx *ctx, uint64_t out3[3]);
/* ---- the slice-level EM of an SR iteration on the device (csrc/svr_em.inc; round 5) -------------------------------------------------
 * The host half of irtkReconstruction::EStepGPU (irtkReconstructionGPU.cc:3282-3420: potentials down, a two-class EM over the slices on the
 * host, slice weights up) as one workgroup behind the E-step's kernels: no wait for the device and no host exchange inside an SR iteration.
 * An extension beside the methods of class Reconstruction (the adaptor of INTEGRATION.md 2 does not need it); csrc/svr_host.cpp uses it.
 *   setup      ns_global slices in the caller's numbering, rank_lo[world + 1] = the ranks' ranges of it (this engine holds rank `rank`'s),
 *              order_or_null[k] = the reference's index of slice k (sums over slices run in the reference's order), step = _step
 *   set_state  the caller's copy of the state, when the caller changed it: global slice weights, force-excluded flags, {mean_s, mean_s2,
 *              sigma_s, sigma_s2, mix_s}, {sigma, mix, m}
 *   svr_mstep_estep_device  iter > 0: the M-step first (one rank: this engine's sums; more: svr_mstep_partial + the launcher's all-gather
 *              before this call); the E-step; *send = {potential, scale, slice_inside} of this rank's slices, 3 x maxn floats, *recv = where
 *              the launcher's all-gather puts every rank's (world x 3 x maxn; one rank: *recv == *send, nothing to do)
 *   run        unpack, the EM, the new slice weights (this rank's part straight into the vector svr_superresolution_backproject(ctx, NULL) reads)
 *   fetch      any of the outputs (NULL = not wanted) in ONE wait: global vectors in the caller's numbering, the scalars */
int svr_slice_em_setup(svr_ctx *ctx, int ns_global, int world, int rank, const int *rank_lo, const int *order_or_null, double step);
int svr_slice_em_set_state(svr_ctx *ctx, const float *slice_weights_global, const unsigned char *excluded_global, const double scalars5[5],
                           const float em3[3]);
int svr_mstep_estep_device(svr_ctx *ctx, int iter, float step, void **send, void **recv, size_t *n_floats_per_rank);
int svr_slice_em_run(svr_ctx *ctx);
int svr_slice_em_apply_weights(svr_ctx *ctx);   /* the EM's weights of this rank's slices -> the scatter's vector again (device to device) */
int svr_slice_em_fetch(svr_ctx *ctx, float *scale_global, float *slice_weight_global, float *slice_potential_global,
                       unsigned char *slice_inside_global, double scalars5[5], float em3[3]);
/* irtkPatchBasedReconstruction's form of the same EM (patchBasedRobustStatistics_gpu.cu:224-556; csrc/pvr_host.cpp uses it): the Gaussian in
 * float with the literal 0.00001f (:97-101), and a patch reads the potential of the patch source_ref[t] (both in the reference's numbering;
 * -1 = none, potential 0; NULL = its own) -- the reference's copy of the stacks' potentials without the stack offset (:256-276).  After setup. */
int svr_slice_em_set_patch_form(svr_ctx *ctx, const int *source_ref_or_null);
/* PSF launches since svr_create that were asked for on the cell path (back_mode 5 / fwd_mode 2: no float atomics, the same bits from run
 * to run) and LEFT it because the cell lists cannot hold the geometry (centre coordinates beyond int16, more than 2^18 cells or 1024
 * planes per class, slices of more than 2^20 pixels, more than 2^17 slices, a staging buffer that does not fit): out4 = {scatters that
 * ran as back_mode 4 (float atomics: last bits depend on the run), gathers and Gaussian pass-1 launches that ran on the tile kernels
 * (same bits, slower), tiles of tiled scatters re-run by a workgroup kernel}.  The first of each kind is also reported on stderr;
 * bench.py puts the four numbers into config.tuned.fallbacks and the tests of the bench workloads require zeros. */
int svr_fallbacks(svr_ctx *ctx, uint64_t out4[4]);
/* a measurement aid, not part of the reconstruction: the time (ms, shortest of three) of `chain` packed f32 fmas per lane (eight independent chains: the full issue rate) on four wavefronts per SIMD
 * of the whole chip: a fixed number of shader cycles, so the time is the inverse of the clock the device sustains under a full vector load.  bench.py prints it (config.device)
 * so that two lines from two boxes of a pool can be told apart from two lines of two binaries. */
int svr_clock_probe(svr_ctx *ctx, int chain, double *ms);
/* the cell lists of the current slice geometry (csrc/svr_cell.inc): out8 = {scatter items, runs, sorted pixels, staging bytes per
 * launch, gather items, gather runs, gather partial-sum bytes per launch, scatter cell size w << 32 | h} */
int svr_cell_stats(svr_ctx *ctx, uint64_t out8[8]);

#ifdef __cplusplus
}
#endif
#endif /* SVR_HIP_H */
