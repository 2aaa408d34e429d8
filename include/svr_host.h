/*
 * svr_host.h -- the host algorithm object above the engine, in the reference's own language.
 *
 * C++ class `svr::irtkReconstruction`: the GPU-path operator surface of the reference's
 * irtkReconstruction (source/reconstructionGPU2/irtkReconstructionGPU.cc, "RG.cc";
 * include/irtkReconstructionGPU.h, "RG.h") with the same method and member names, driving one
 * engine context (include/svr_hip.h) per process:
 *   InitializeEMValuesGPU RG.cc:2905-2919      InitializeEMGPU RG.cc:2921-2953
 *   GaussianReconstructionGPU RG.cc:2695-2762  SimulateSlicesGPU RG.cc:1163-1175
 *   InitializeRobustStatisticsGPU RG.cc:2988-3019   EStepGPU RG.cc:3184-3440
 *   ScaleGPU RG.cc:3751-3757   SuperresolutionGPU RG.cc:4024-4036   MStepGPU RG.cc:4214-4223
 *   MaskVolumeGPU RG.cc:5319-5323   ScaleVolumeGPU / RestoreSliceIntensitiesGPU RG.cc:4216,4653
 *   SetSmoothingParameters RG.h:605-612
 * and the reconstruction part of main()'s loop (reconstruction.cc:930-1140) as
 * reconstruct_iteration() / sr_iteration().
 *
 * Multi-GPU: one process per GPU; the slices are sharded, the collectives are supplied by the
 * launcher through `svr_collectives` (bench.py plugs torch.distributed/RCCL in; NULL = one rank).
 * A plain C-ABI (svrh_*) over the class is exported for bindings and tests.
 */
#ifndef SVR_HOST_H
#define SVR_HOST_H

#include <stddef.h>

#include "svr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* collectives supplied by the launcher; every callback returns 0 on success.
 * struct_size (first member, since round 5): sizeof(svr_collectives) of the header the LAUNCHER was built against.  The host objects copy
 * that many bytes and take every member beyond them as NULL / 0 -- a launcher built against an older (shorter) version of this struct
 * keeps working and simply gets the all-reduce + replicated update; 0 or anything smaller than the members up to allgather_slices is
 * refused by svrh_create / pvrh_create_sharded.  (Rounds 1-4 had no such member: launchers of that age must be rebuilt, INTEGRATION.md.) */
typedef struct svr_collectives {
  size_t struct_size;
  void *user;
  int rank, world;
  /* in-place sum of the float[2*Nv] device buffer at `device_ptr` over all ranks */
  int (*allreduce_volume_pair)(void *user, void *device_ptr, size_t n_floats);
  /* in-place reduction of a small host vector; op: 0 sum, 1 min, 2 max */
  int (*allreduce_host)(void *user, double *data, int n, int op);
  /* gather the per-slice vectors of all ranks (rank order = slice order) into global[n_global] */
  int (*allgather_slices)(void *user, const float *local, int n_local, float *global_out, int n_global);
  /* device buffers, on the engine's stream or ordered behind it (NULL: the host objects all-reduce the volume pair and run the volume
   * update replicated instead of by z-slabs, csrc/svr_slab.inc):
   * recv[n] = sum over ranks r' of send_{r'}[rank * n .. rank * n + n)   (send = float[world * n]) */
  int (*reduce_scatter_device)(void *user, const void *send, void *recv, size_t n_floats_per_rank);
  /* recv[r * n .. r * n + n) = send_r[0 .. n) for every rank r   (recv = float[world * n]) */
  int (*allgather_device)(void *user, const void *send, void *recv, size_t n_floats_per_rank);
  /* 1: the three device-buffer callbacks enqueue on the engine's stream (svr_get_stream) -- the host objects then launch scatter,
   * collective and volume update back to back without waiting for the device.  0: the host objects synchronise the engine's stream
   * before every device-buffer callback (its input may still be in flight there), and the callback returns with its result complete. */
  int on_engine_stream;
} svr_collectives;
/* the smallest struct_size the host objects accept: everything up to and including allgather_slices */
#define SVR_COLLECTIVES_MIN_SIZE (offsetof(svr_collectives, allgather_slices) + sizeof(void *))

/* ---- the collectives on RCCL, bound directly (csrc/svr_rccl.cpp) ---------------------------------------------------
 * One communicator per rank = per GPU: a process of its own (bench.py) or a thread of the command line (`-d 0 1 ..`).
 * Replaces the reference's per-device worker threads with their reduce on GPU 0 (reconstruction_cuda2.cu:1413-1457,
 * 2225-2239).  The volume pair is all-reduced in place on the engine's stream.  id128 = ncclUniqueId made by ONE rank
 * (svr_comm_unique_id) and handed to all; svr_comm_create blocks until all `world` ranks have called it. */
typedef struct svr_comm svr_comm;
int svr_comm_unique_id(char id128[128]);
svr_comm *svr_comm_create(int rank, int world, const char id128[128], svr_ctx *engine);
int svr_comm_rebind(svr_comm *c, svr_ctx *engine);                  /* the same communicator for another engine context on the same device */
const svr_collectives *svr_comm_collectives(svr_comm *c);
int svr_comm_world(svr_comm *c);                                     /* ncclCommCount */
int svr_comm_allreduce_host(svr_comm *c, double *data, int n, int op); /* op: 0 sum, 1 min, 2 max */
const char *svr_comm_last_error(const svr_comm *c);
void svr_comm_destroy(svr_comm *c);
/* the ranks of ONE process (the command line's `-d 0 1 2 ..`, one thread and one engine context per rank): RCCL
 * communicators when the devices are distinct, an exchange through host memory under a thread barrier when a device is
 * named more than once (RCCL refuses that; a test mode for one-GPU boxes).  svr_group_join is called by every rank from its
 * own thread and blocks until all have joined; NULL for world 1. */
typedef struct svr_group svr_group;
svr_group *svr_group_create(int world, const int *devices);
int svr_group_uses_rccl(const svr_group *g);
const svr_collectives *svr_group_join(svr_group *g, int rank, svr_ctx *engine);
void svr_group_destroy(svr_group *g);

typedef struct svrh_recon svrh_recon;

/* irtkReconstruction(std::vector<int> dev, bool useCPUReg)  RG.cc:159-221; the engine context is
 * created and filled (SyncGPU, RG.cc:249-328) by the caller.  [slice_lo, slice_hi) = this rank's
 * slices out of n_slices_global. */
svrh_recon *svrh_create(svr_ctx *engine, int n_slices_global, int slice_lo, int slice_hi,
                        const svr_collectives *coll_or_null);
void svrh_destroy(svrh_recon *r);
const char *svrh_last_error(const svrh_recon *r);

void svrh_set_intensity_range(svrh_recon *r, double min_intensity, double max_intensity); /* RG.cc:2937-2951 */
void svrh_set_smoothing_parameters(svrh_recon *r, double delta, double lambda);             /* RG.h:605-612 */
void svrh_set_force_excluded(svrh_recon *r, const int *idx, int n);                         /* RG.h:614-617 */
/* `intensity_matching` of reconstruction.cc:114,183 (--no_intensity_matching 0): off = no Bias / Scale / NormaliseBias in the SR
 * iterations (reconstruction.cc:1018-1045, 1062-1076); the scales stay 1 */
void svrh_set_intensity_matching(svrh_recon *r, int on);

/* disableBiasCorrection() RG.cc:234-238 / SetSigma RG.h:376; BiasGPU RG.cc:3904-3913, NormaliseBiasGPU RG.cc:4653 */
int svrh_set_bias_correction(svrh_recon *r, int enable, double sigma_bias);
int svrh_bias_gpu(svrh_recon *r);
int svrh_normalise_bias_gpu(svrh_recon *r, int iter);

int svrh_initialize_em_values_gpu(svrh_recon *r);
int svrh_gaussian_reconstruction_gpu(svrh_recon *r);
int svrh_simulate_slices_gpu(svrh_recon *r);
int svrh_initialize_robust_statistics_gpu(svrh_recon *r);
int svrh_estep_gpu(svrh_recon *r);
int svrh_scale_gpu(svrh_recon *r);
int svrh_superresolution_gpu(svrh_recon *r, int iter);
int svrh_mstep_gpu(svrh_recon *r, int iter);
int svrh_mask_volume_gpu(svrh_recon *r);
int svrh_scale_volume_gpu(svrh_recon *r);
/* one SR iteration of reconstruction.cc:1013-1108 (bias off) */
int svrh_sr_iteration(svrh_recon *r, int i);
/* Gaussian init + robust-statistics init + rec_iterations SR iterations + MaskVolume
 * (reconstruction.cc:930-1140) */
int svrh_reconstruct_iteration(svrh_recon *r, int rec_iterations);

/* ---- GPU slice-to-volume registration, host side (irtkReconstructionGPU.cc:2104-2290) ------------
 * irtkImageAttributes subset of one slice (z size 1): voxel counts and sizes, axes, origin = world
 * position of the image centre (IRTKSimple2/image++/src/irtkBaseImage.cc:79-147). */
typedef struct svr_image_attr {
  int nx, ny, nz;
  double dx, dy, dz;
  double xaxis[3], yaxis[3], zaxis[3], origin[3];
} svr_image_attr;
/* PrepareRegistrationSlices RG.cc:2104-2181: resample this rank's slices to the reconstruction's voxel
 * size with irtkResamplingWithPadding (padding -1), pack plane 0 of each into the padded grid and hand it
 * to the engine (initRegStorageVolumes + FillRegSlices).  slices [n_local][sy][sx]. */
int svrh_prepare_registration_slices(svrh_recon *r, const float *slices, int sx, int sy, const svr_image_attr *attrs,
                                     double recon_voxel);
/* SliceToVolumeRegistrationGPU RG.cc:2214-2290: transformations = row-major double 4x4 per local slice
 * (`_transformations_gpu`), updated in place; the engine registers against its current reconstruction. */
int svrh_slice_to_volume_registration_gpu(svrh_recon *r, double *transformations);
/* resampled grid of the last svrh_prepare_registration_slices: {x, y, slices} and a copy of the packed data */
int svrh_get_registration_slices(svrh_recon *r, int size3[3], float *data_or_null);

/* ---- NIfTI-1 I/O with the reference's image conventions (csrc/svr_io.cpp) ------------
 * read : .nii / .nii.gz, any of uint8/int8/int16/uint16/int32/uint32/float32/float64, either byte order,
 *        scl_slope/scl_inter applied; geometry from the qform, else the sform, else the default matrix, as
 *        irtkFileNIFTIToImage::ReadHeader does (irtkFileNIFTIToImage.cc:168-345): axes = matrix columns /
 *        voxel size, origin = world position of the centre voxel.  *data is malloc'ed float[nx*ny*nz*nt]
 *        (x fastest), release with svr_free.
 * write: float32 single-file NIfTI-1, qform_code 1 from the image-to-world matrix, sform_code 0
 *        (irtkImageToFileNIFTI.cc:65-145, irtkNIFTI.h:84-160); ".gz" suffix = gzip. */
int svr_nifti_read(const char *path, svr_image_attr *attr, int *nt_or_null, float **data, char err[256]);
int svr_nifti_write(const char *path, const svr_image_attr *attr, const float *data, char err[256]);
void svr_free(void *p);
/* IRTK rigid transformation files (`dof`, the reference's -t option): big-endian {815007, 2, 6} + tx ty tz rx ry rz
 * as doubles (irtkRigidTransformation.cc:392-451); matrix16_or_null = UpdateMatrix's 4x4 (:26-53), row-major. */
int svr_dof_read(const char *path, double params6[6], double *matrix16_or_null, char err[256]);
int svr_dof_write(const char *path, const double params6[6], char err[256]);
/* The number of host threads worth starting: the CPUs of the affinity mask, cut to the cgroup's CPU quota (cpu.max, or
 * cpu.cfs_quota_us / cpu.cfs_period_us) -- a container with 256 visible CPUs and a quota of 16 loses whole scheduler
 * periods to throttling when a pool starts 128 threads.  SVR_HOST_THREADS overrides.  Used by the registration's work
 * pool, the pre-processing of the command lines and the NIfTI writer. */
int svr_host_threads(void);

/* The numbering of a sharded run (round 5).  A launcher may deal the units to the ranks in any order -- e.g. the r-th part of EVERY stack
 * to rank r, so that a rank's slices are neighbours in space and it stages 1/N of the volume's work items instead of those of whole
 * stacks (csrc/svr_shard.h spatial_order; fetalreconstruction_amd/sharding.py shard_units) -- by uploading them in that order and
 * telling the host object: order[k] = index in the REFERENCE's order (stack after stack, slice after slice) of unit k of the numbering
 * this object and its engine use (n_slices_global entries, a permutation; NULL = the same numbering).  Per-unit vectors handed to or
 * returned by svrh_* / pvrh_* (scale, weights, potentials, force-excluded indices, transformations) are in the object's numbering;
 * whatever the reference computes ACROSS units in unit order -- the sums of the slice-level EM (RG.cc:3282-3420), the patch-based
 * path's within-stack indexing of the potentials (patchBasedRobustStatistics_gpu.cu:256-276) -- is computed in the reference's order. */
int svrh_set_unit_order(svrh_recon *r, const int *order_or_null);
/* test hook: a world-1 run goes through the launcher's collectives like a sharded one (instead of an environment variable) */
void svrh_force_collectives(svrh_recon *r, int on);
/* sharded runs: 1 (default) = the volume update by z-slabs when the launcher supplies reduce_scatter_device / allgather_device
 * (csrc/svr_slab.inc), 0 = all-reduce of the pair + the update replicated on every rank.  Identical on every rank either way; the two
 * forms are bit-equal to each other when the collectives add the ranks in rank order (host-staged / gloo), equal to a float sum's last
 * bits over RCCL. */
void svrh_set_slab_update(svrh_recon *r, int on);

/* state read-back: global per-slice vectors (length n_slices_global) and the EM scalars
 * scalars8 = {sigma, mix, m, mean_s, mean_s2, sigma_s, sigma_s2, mix_s}.  Sharded runs: COLLECTIVE -- the scale vector and
 * slice_inside of the other ranks travel with a rank's next exchange, so every rank must call this together (it completes
 * them with one exchange if any are pending). */
int svrh_get_state(svrh_recon *r, float *scale, float *slice_weight, float *slice_potential,
                   unsigned char *slice_inside, double scalars8[8]);

/* ---- the reference's default (CPU / IRTK) registration schedule around the batched NCC cost (csrc/irtk_reg.cpp) ----
 * irtkImageRigidRegistrationWithPadding::Run (3 levels, Gaussian blurring + resampling with padding per level, gradient
 * descent with 4 step halvings and up to 20 iterations per step, cross correlation) restated on the host, with every
 * similarity evaluation going through svr_ncc_evaluate -- all targets that share a source advance in lock step, one
 * batched call per optimiser step.  `backend` NULL = the engine `ctx`; tests pass the CPU oracle's evaluator instead. */
typedef struct svr_ncc_backend {
  void *user;
  int (*set_targets)(void *user, int n, int tx, int ty, const int16_t *targets);
  int (*set_source)(void *user, const uint32_t size[3], const int16_t *source);
  int (*evaluate)(void *user, int n_eval, const int *target_index, const double *matrices, int64_t *sums6, double *ncc_or_null);
} svr_ncc_backend;
/* irtkReconstruction::StackRegistrations (irtkReconstructionGPU.cc:849-1001): every stack against the (masked) template
 * stack; transformations: row-major double [n][16], in/out.  stacks[i]: double [nz][ny][nx] of attrs[i].
 * flags: SVRH_STACKREG_KEEP_ORIGIN = irtkStack3D3DRegistration<T>::run of the patch-based command line
 * (irtkStack3D3DRegistration.cpp:164-228), whose ResetOrigin is a no-op (arguments by value). */
#define SVRH_STACKREG_KEEP_ORIGIN 1
int svrh_stack_registrations(svr_ctx *ctx, const svr_ncc_backend *backend, int n_stacks, const svr_image_attr *attrs,
                             const double *const *stacks, double *transformations, int template_number,
                             const svr_image_attr *mask_attr, const double *mask_or_null, int flags, long *n_evaluations_or_null,
                             char err[256]);
/* irtkReconstruction::SliceToVolumeRegistration (irtkReconstructionGPU.cc:1991-2059, 2291-2303): every slice against the
 * current reconstruction.  slices: the padded grid float [n][sy][sx] (-1 = padding), attrs[i]: slice i's attributes.
 * flags: SVRH_S2V_NO_RESAMPLE = the patch-to-volume registration the patch-based command line actually runs
 * (patchBased2D3DRegistration<T>::runHybrid + ParallelPatchToVolumeRegistration, patchBased2D3DRegistration.cpp:88-225): the
 * same schedule on the patches as they are. */
#define SVRH_S2V_NO_RESAMPLE 1
int svrh_slice_to_volume_registration(svr_ctx *ctx, const svr_ncc_backend *backend, int n_slices, const float *slices, int sx, int sy,
                                      const svr_image_attr *attrs, double *transformations, const svr_image_attr *recon_attr,
                                      const float *reconstructed, int flags, long *n_evaluations_or_null, char err[256]);
/* irtkReconstruction::PackageToVolume (irtkReconstructionGPU.cc:5096-5192) with SplitImage / SplitImageEvenOdd /
 * SplitImageEvenOddHalf / HalfImage (:4979-5094): the packages (interleaved sub-stacks) of every stack are registered to
 * the reconstruction as 3-D targets and their transformations handed to their slices.  pack_num[i]: packages of stack i;
 * transformations: one per slice, double [n_slices][16], in/out. */
int svrh_package_to_volume(svr_ctx *ctx, const svr_ncc_backend *backend, int n_stacks, const svr_image_attr *attrs, const double *const *stacks,
                           const int *pack_num, int evenodd, int half, int half_iter, double *transformations,
                           const svr_image_attr *recon_attr, const float *reconstructed, long *n_evaluations_or_null, char err[256]);
/* building blocks (irtkResamplingWithPadding<short>, irtkGaussianBlurringWithPadding<short>, irtkRigidTransformation::
 * Matrix2Parameters / UpdateMatrix), exported for the tests */
int svrh_irtk_resample_with_padding(const svr_image_attr *attr, const int16_t *data, double rx, double ry, double rz, int padding,
                                    svr_image_attr *out_attr, int16_t *out_or_null, long capacity);
int svrh_irtk_blur_with_padding(const svr_image_attr *attr, int16_t *data, double sigma, int padding);
void svrh_irtk_rigid_parameters(const double matrix16[16], double params6[6], double *rebuilt16_or_null);

/* ---- patch-to-volume reconstruction loop (csrc/pvr_host.cpp) ---------------------------
 * svr::irtkPatchBasedReconstruction: the reconstruction part of irtkPatchBasedReconstruction<T>::run
 * (irtkPatchBasedReconstruction.cpp:445-593) and the host halves of patchBasedRobustStatistics_gpu<T>
 * (patchBasedRobustStatistics_gpu.cu: initializeEMValues :78-95, EStep :224-556, MStep :570-640, Scale :672-745,
 * InitializeRobustStatistics :793-845).  The engine must have the option "pvr" set and the patches uploaded
 * as its slices; patches_per_stack is PatchBasedVolume::getXYZPatchGridSize().z per stack. */
typedef struct pvrh_recon pvrh_recon;
pvrh_recon *pvrh_create(svr_ctx *engine, const int *patches_per_stack, int n_stacks, float min_intensity, float max_intensity);
/* sharded over ranks ("identical with patches as the unit"; the reference's patch-based path is single-GPU,
 * patchBasedReconMain.cpp:78,177-179): the engine holds the patches [patch_lo, patch_hi) of the global numbering (stack
 * after stack, patches_per_stack = the GLOBAL counts), every rank the whole volume.  Exchanges per SR iteration: one
 * all-reduce of addon|cmap, the M-step's five scalars and the E-step's patch potentials (with the scale vector riding along);
 * the patch-level EM runs replicated on the global vectors -- including the reference's within-stack indexing of the patch
 * potentials (patchBasedRobustStatistics_gpu.cu:256-276). */
pvrh_recon *pvrh_create_sharded(svr_ctx *engine, const int *patches_per_stack, int n_stacks, float min_intensity, float max_intensity,
                                int patch_lo, int patch_hi, const svr_collectives *coll_or_null);
int pvrh_set_unit_order(pvrh_recon *r, const int *order_or_null);    /* as svrh_set_unit_order, with patches */
void pvrh_force_collectives(pvrh_recon *r, int on);
void pvrh_set_slab_update(pvrh_recon *r, int on);
void pvrh_destroy(pvrh_recon *r);
const char *pvrh_last_error(const pvrh_recon *r);
int pvrh_initialize_em_values(pvrh_recon *r);
int pvrh_initialize_robust_statistics(pvrh_recon *r);
int pvrh_estep(pvrh_recon *r);
int pvrh_mstep(pvrh_recon *r, int iter);
int pvrh_scale(pvrh_recon *r);
/* one outer iteration without the patch registration (irtkPatchBasedReconstruction.cpp:490-548) */
int pvrh_reconstruct_iteration(pvrh_recon *r, int rec_iterations);
int pvrh_sr_iteration(pvrh_recon *r, int i);   /* one SR iteration: Scale, scatter (+ all-reduce) + regulariser, simulate, M-step, E-step */
/* the patch-to-volume registration between the outer iterations (irtkPatchBasedReconstruction.cpp:452-489): svr_pvr_register_patches,
 * then the new transformations go to the engine.  T / Tinv [n][16] in/out; counters3 = {launches, evaluations, patches} */
int pvrh_register_patches(pvrh_recon *r, const float *ri2w, const float *mo, const float *invmo, float *T, float *Tinv, const float *i2w,
                          const float *w2i, const float *recon_i2w, const float *recon_w2i, long long counters3[3]);
/* scalars8 = {sigma, mix, m, mean_s, mean_s2, sigma_s, sigma_s2, mix_s}; vectors of length sum(patches_per_stack) */
int pvrh_get_state(pvrh_recon *r, float *scale, float *patch_weight, float *patch_potential, double scalars8[8]);

#ifdef __cplusplus
}
#endif
#endif /* SVR_HOST_H */
