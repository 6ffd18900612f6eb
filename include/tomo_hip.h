/*
 * tomo_hip.h -- C ABI of libtomo_hip.so, the MI355X (gfx950) engine for the hot path of
 * jtschwar/tomo_TV: parallel-beam forward projection, voxel-driven back-projection, SIRT/SART
 * updates and the 3-D TV kernels, behind the reference's `tomoengine` / `ctvlib` method surface.
 *
 * This is the drop-in boundary: every entry point names the reference interface it replaces
 * (paths relative to the reference tree).  Plain pointers and sizes only.  Host arrays use the
 * reference's layouts:
 *     volume    float32 [Nslice][Ny][Nz]                 (container/Matrix3D.cpp:20-23)
 *     sinogram  float32 [Nslice][Nproj*Nray], index angle*Nray+ray   (gpu/reconstructor.py:54-56)
 * Device-resident state lives inside the opaque engine (slab-interleaved layout).
 *
 * Every function returns 0 on success or a tomo_status code; tomo_last_error() gives the text.
 * One engine = one device slab of slices; calls on one engine must be serialised by the caller.
 */
#ifndef TOMO_HIP_H
#define TOMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tomo_engine tomo_engine;

enum tomo_status { TOMO_OK = 0, TOMO_ERR_ARG = 1, TOMO_ERR_HIP = 2, TOMO_ERR_STATE = 3, TOMO_ERR_GEOMETRY = 4 };

/* volumes held by the engine (tomoengine.hpp:44: recon, temp_recon, original_volume, yk, recon_old) */
enum tomo_volume { TOMO_VOL_RECON = 0, TOMO_VOL_TEMP = 1, TOMO_VOL_ORIGINAL = 2, TOMO_VOL_YK = 3, TOMO_VOL_RECON_OLD = 4,
                   TOMO_VOL_COUNT = 5,
                   TOMO_VOL_USER0 = 5,      /* caller-managed extra volumes (per-element tomograms of Matrix4D, multimodal.hpp) */
                   TOMO_VOL_SLOTS = 5 + 40 };
/* sinograms held by the engine (tomoengine.hpp:53: b = measured, g = re-projection) */
enum tomo_sinogram { TOMO_SINO_B = 0, TOMO_SINO_G = 1, TOMO_SINO_R = 2 /* residual scratch */,
                     TOMO_SINO_USER0 = 3,   /* caller-managed extra sinograms (per-element bChem, multimodal.hpp) */
                     TOMO_SINO_SLOTS = 3 + 40,
                     /* read-only pseudo-slot of tomo_get_sinogram: A * yk as tomo_fista_project_yk formed it (TOMO_ERR_STATE when
                      * no such projection is in hand); TOMO_SINO_G itself always stays A * recon (tomoengine.cpp:410-427,459) */
                     TOMO_SINO_YK_MODEL = 1000 };

/* slots of the device scalar buffer (doubles); each holds THIS slab's partial sum */
enum tomo_scalar { TOMO_S_DD = 0,      /* sum (A x - b)^2            data_distance  */
                   TOMO_S_DIFF = 1,    /* sum (recon - temp)^2       matrix_2norm   */
                   TOMO_S_TV = 2,      /* sum sqrt(eps + |grad|^2)   tv / tv_gd / tv_fgp return value */
                   TOMO_S_GNORM = 3,   /* sum g^2 of the TV gradient (one tv_gd inner iteration) */
                   TOMO_S_RMSE = 4,    /* sum (recon - original)^2   rmse           */
                   TOMO_S_COST = 5,    /* Poisson-ML cost            poisson_ML     */
                   TOMO_S_L1 = 6,      /* sum |recon|                l1_norm        */
                   TOMO_S_GNORM_ALL = 8, /* sum g^2 over the sub-slab engines of a slab group (tomo_scalar_sum_from) */
                   TOMO_S_DIFF2 = 7,   /* a second step-norm slot: ASD-POCS keeps the SART step norm here until the
                                          iteration's scalars are read together (one all-reduce, one read-back) */
                   TOMO_S_COUNT = 16 };

const char *tomo_last_error(void);

/* GPU count; replaces tomofusion/__init__.py:10-18 (pycuda device_count). */
int tomo_device_count(int *count);

/* Host-only system matrix: drop-in for parallelRay(Nside, angles) of tomofusion/cpu/utils/pytvlib.py:8-121.
 * angles_rad[i] must be angle_deg*pi/180 evaluated in double.  Writes up to cap entries of (row, col, val)
 * as float32 in parallelRay's generation order and the entry count to *nnz (call with cap = 0 to size). */
int tomo_system_matrix(int nray, int nproj, const double *angles_rad, int64_t cap,
                       float *rows, float *cols, float *vals, int64_t *nnz);

/* tomoengine(Nslice, Nray, angles) -- gpu/utils/tomoengine.cpp:48-84.  Builds the line-intersection system
 * matrix of parallelRay for these angles and uploads its ray (CSR) and per-angle voxel tables. */
int tomo_create(int nslice, int nray, int nproj, const double *angles_rad, int device, tomo_engine **out);

/* ctvlib(Nslice, Nray, Nproj) + load_A(A) -- cpu/utils/ctvlib.cpp:28-48, 309-315.  A given as the three
 * float32 rows of parallelRay's output.  The matrix must have at most two rays of one angle per pixel. */
int tomo_create_from_matrix(int nslice, int nray, int nproj, int64_t nnz, const float *rows, const float *cols,
                            const float *vals, int device, tomo_engine **out);

int tomo_destroy(tomo_engine *e);

/* Run all work of this engine on an existing HIP stream (hipStream_t), e.g. torch's current stream. */
int tomo_set_stream(tomo_engine *e, void *hip_stream);
int tomo_synchronize(tomo_engine *e);
int tomo_get_device(tomo_engine *e, int *device);                       /* tomoengine.cpp:92-95 get_gpu_id */
int tomo_get_dims(tomo_engine *e, int *nslice, int *nray, int *nproj, int64_t *nnz);

/* ---- data in / out ------------------------------------------------------------------------------- */
int tomo_set_tilt_series(tomo_engine *e, const float *b);               /* tomoengine.cpp:101 setTiltSeries */
int tomo_set_sinogram(tomo_engine *e, int which, const float *b);       /* multimodal.cpp:150-151 set_haadf/chem_tilt_series */
int tomo_get_sinogram(tomo_engine *e, int which, float *out);           /* :454-458 get_projections / get_model_projections */
int tomo_set_volume(tomo_engine *e, int vol, const float *data);        /* all slices at once */
int tomo_get_volume(tomo_engine *e, int vol, float *data);
int tomo_set_slice(tomo_engine *e, int vol, int s, const float *img);   /* :104-106 setOriginalVolume / setRecon */
int tomo_get_slice(tomo_engine *e, int vol, int s, float *img);         /* :452 getRecon */
int tomo_restart_recon(tomo_engine *e);                                 /* :462-468 restart_recon */
int tomo_copy_volume(tomo_engine *e, int dst, int src);                 /* :404 copy_recon = (TEMP <- RECON) */
/* volume of another engine with the same slab shape (rebuilding the tilt geometry keeps the reconstruction:
 * tomoengine::update_projection_angles, tomoengine.cpp:128-149) */
int tomo_copy_volume_from(tomo_engine *dst, int dst_vol, tomo_engine *src, int src_vol);
/* the same rebuild without a copy and with one set of tables in memory: release the old engine's tables and sinograms
 * (only tomo_adopt_volumes / tomo_destroy remain valid on it), create the new engine, move the volumes over
 * (ctvlib::update_proj_angles, ctvlib.cpp:317-333, keeps recon across a matrix change as well) */
int tomo_release_geometry(tomo_engine *e);
int tomo_adopt_volumes(tomo_engine *dst, tomo_engine *src);

/* ---- projector -------------------------------------------------------------------------------------- */
int tomo_forward_projection(tomo_engine *e, int vol, int sino);         /* :416-427 forwardProjection; :109-126 create_projections */
int tomo_back_projection(tomo_engine *e, int sino, int vol);            /* :279-291 back_projection: vol = A^T sino */
int tomo_lipschitz(tomo_engine *e, float *L);                           /* ctvlib.cpp:194-202 lipschits; tomoengine.cpp:369-371 */
int tomo_row_inner_product(tomo_engine *e);                             /* ctvlib.cpp:234-242 normalization */
int tomo_lipschitz_cimmino(tomo_engine *e, float *L);                   /* ctvlib.cpp:198-199: max(A^T M A 1), M = diag(|A_i|^2) */

/* ---- reconstruction steps --------------------------------------------------------------------------- */
/* ctvlib::SIRT(beta): x = max(0, x + beta A^T (b - A x))   ctvlib.cpp:205-221.  vol = RECON or YK. */
int tomo_sirt_landweber(tomo_engine *e, int vol, float beta, int niter);
/* ctvlib::SIRT(beta) after cimminos_method(): x = max(0, x + A^T M (b - A x) beta/Nrow), M_ii = |A_i|^2 -- the
 * reference multiplies by the row norms (quirk Q10); reproduced as written   ctvlib.cpp:212-216, 245-251 */
int tomo_sirt_cimmino(tomo_engine *e, int vol, float beta, int niter);
/* tomoengine::SIRT(nIter) (ASTRA SIRT, min-constraint 0): x = max(0, x + C A^T R (b - A x))  tomoengine.cpp:181-205 */
int tomo_sirt(tomo_engine *e, int vol, int niter);
/* the same with the measured data in any sinogram slot: multimodal::SIRT(e, s, nIter)  multimodal.cpp:339-358 */
int tomo_sirt_data(tomo_engine *e, int vol, int sino_b, int niter);
/* tomoengine::SART(beta, nIter): nIter sweeps of per-angle updates, order[] = angle permutation or NULL
 * (sequential)  tomoengine.cpp:151-179 */
int tomo_sart(tomo_engine *e, int vol, float beta, int niter, const int32_t *order);
int tomo_sart_data(tomo_engine *e, int vol, int sino_b, float beta, int niter, const int32_t *order); /* multimodal.cpp:377-396 */
/* The sweep of tomo_sart_data whose last back-projection also leaves ||x_new - track_vol||^2 in scalar `slot` and
 * copies x_new into track_vol: the `matrix_2norm()` + `copy_recon()` pair that follows the SART sweep in the ASD-POCS
 * loop (examples/sim_ASD.py:70-78, gpu/reconstructor.py:168-176) without two more passes over the slab. */
int tomo_sart_tracked(tomo_engine *e, int vol, int sino_b, float beta, int niter, const int32_t *order, int track_vol, int slot);
/* tomoengine::CGLS(nIter): CGLS restarted from the current volume, per slice, then positivity  tomoengine.cpp:207-229 */
int tomo_cgls(tomo_engine *e, int vol, int niter);
/* tomoengine::FBP(apply_positivity): recon = scale * A^T (taps * b), taps[0..Nray-1] = symmetric real-space filter
 * (built by the host for the named filter, pytvlib.py:33-36)  tomoengine.cpp:317-347 */
int tomo_fbp(tomo_engine *e, const float *taps_host, float scale, int apply_positivity);
/* ctvlib::ART(beta): row-action Kaczmarz sweep + positivity  ctvlib.cpp:137-155 */
int tomo_art(tomo_engine *e, float beta);
/* ctvlib::randART(beta) with the rows visited in the given permutation (the reference's loop, ctvlib.cpp:158-179,
 * rewrites its own counter and draws from an unseedable device: quirk Q9)  */
int tomo_art_order(tomo_engine *e, float beta, const int32_t *order_host);
/* tomoengine::poisson_ML(lambda): cost accumulates in TOMO_S_COST  tomoengine.cpp:231-246, 293-315 */
int tomo_poisson_ml(tomo_engine *e, float lambda);
/* out = (A x - b)/(A x + 0.1), cost sum(Ax - b log(Ax + 0.1)) -> TOMO_S_COST   multimodal.cpp:284-292, 466-473 */
int tomo_poisson_residual(tomo_engine *e, int vol, int sino_b, int sino_out);
int tomo_scale_volume(tomo_engine *e, int vol, float factor);           /* multimodal.cpp:307-309 rescale_tomograms */
int tomo_positivity(tomo_engine *e, int vol);                           /* ctvlib.cpp:224-231 */
int tomo_soft_threshold(tomo_engine *e, int vol, float lambda);         /* matrix_ops.cu:64-75 */
int tomo_fista_momentum(tomo_engine *e, float beta);                    /* tomoengine.cpp:381-384 */
/* A yk = (1 + beta) A recon - beta A recon_old by linearity, from the projections the driver's cost evaluation made anyway
 * (gpu/reconstructor.py:121-155): call after tomo_data_distance_sq(RECON); a no-op unless provably valid; *done = 1 when the model
 * sinogram now holds A yk and the next tomo_sirt(YK) will start from it */
int tomo_fista_project_yk(tomo_engine *e, int *done);

/* ---- scalar reductions: partial sums of this slab land in the device scalar buffer ------------------ */
int tomo_data_distance_sq(tomo_engine *e, int vol);                     /* tomoengine.cpp:410-413 -> TOMO_S_DD (also fills G) */
/* the same evaluation on a second stream, for a volume the main sequence no longer writes (e.g. the TEMP copy);
 * tomo_async_wait (also implied by tomo_read_scalars) orders the main stream behind it */
int tomo_data_distance_sq_async(tomo_engine *e, int vol);
int tomo_async_wait(tomo_engine *e);
int tomo_diff_norm_sq(tomo_engine *e, int a, int b, int slot);          /* :407 matrix_2norm, :433 rmse */
int tomo_sino_diff_norm_sq(tomo_engine *e, int a, int b, int slot);     /* multimodal.cpp:489 (g - bh).norm() */
int tomo_sino_proj_max(tomo_engine *e, int sino, float *out_host);      /* multimodal.cpp:323-327: max per projection */
int tomo_sino_proj_scale(tomo_engine *e, int sino, const float *div_host, const float *mul_host); /* b <- b/div[p]*mul[p] */
int tomo_l1_norm(tomo_engine *e, int vol);                              /* :436 l1_norm -> TOMO_S_L1 */
int tomo_read_scalars(tomo_engine *e, double *out, int count);          /* synchronises the stream */
/* the same without a pipeline bubble: _snapshot enqueues the copy of all slots behind the kernels that produce them (and behind
 * an evaluation in flight on the second stream) and returns; _read waits for that copy only.  The ASD-POCS drivers enqueue the
 * next SART sweep in between (the scalars of iteration i decide nothing before the TV steps of iteration i+1:
 * examples/sim_ASD.py:90-94). */
int tomo_scalars_snapshot(tomo_engine *e);
int tomo_scalars_snapshot_read(tomo_engine *e, double *out, int count);
int tomo_bind_scalar_buffer(tomo_engine *e, void *device_doubles);      /* >= TOMO_S_COUNT doubles, e.g. a torch tensor */

/* ---- 3-D total variation ---------------------------------------------------------------------------- */
/* Slices of a slab have neighbours on other ranks: two device buffers of Nray*Nray floats hold the slice
 * below the first local slice (lo) and above the last (hi).  tomo_halo_local fills them from this slab
 * (periodic wrap = single-rank behaviour of ctvlib.cpp:348,421-422); a distributed caller packs, exchanges
 * (mpi_ctvlib.cpp:400-422) and leaves the received planes in the bound buffers. */
int tomo_bind_halo(tomo_engine *e, void *device_lo, void *device_hi);
int tomo_halo_pack(tomo_engine *e, int field, int last, void *device_dst); /* field: tomo_volume or TOMO_FIELD_* */
int tomo_halo_pack_both(tomo_engine *e, int field, void *first_plane, void *last_plane);  /* both planes, one launch */
int tomo_halo_local(tomo_engine *e, int field);
enum tomo_field { TOMO_FIELD_FGP_D = 100, TOMO_FIELD_FGP_P1 = 101 };
/* Several slab engines on ONE device (a slab run as K sub-slabs, each with its own allocations and stream, so that their
 * dependent launch chains overlap): the three calls the coupling of the sub-slabs needs -- stream ordering, halo planes taken
 * straight from the neighbours' volumes (ring of sub-slabs = the periodic wrap of ctvlib.cpp:348,421; mpi_ctvlib.cpp:400-422
 * does it with MPI), and the sum of the engines' partial sums on the device (mpi_ctvlib.cpp:547 MPI_Allreduce).
 * "tv_gnorm_slot" (tomo_set_option) names the scalar slot the TV update reads ||g||^2 from. */
int tomo_wait_for(tomo_engine *e, tomo_engine *other);
int tomo_halo_from(tomo_engine *e, int field, tomo_engine *lo_src, tomo_engine *hi_src);
int tomo_scalar_sum_from(tomo_engine *e, int dst_slot, tomo_engine **srcs, int n, int src_slot);
/* global-edge flags for the non-periodic FGP stencil (tv_fgp.cu:57,81) */
int tomo_set_slab_edges(tomo_engine *e, int is_first, int is_last);

/* the volume the tv_gd / tv_grad / tv_update forms descend (default RECON): multimodal::tv_gd_4D runs the same descent
 * on every element's tomogram (multimodal.cpp:494, chemistry/utils/regularizers/tv_gd.cu:208-296) */
int tomo_tv_set_target(tomo_engine *e, int vol);
/* step forms (use the halo buffers as they are) */
int tomo_tv_partial(tomo_engine *e, int vol, float eps);                /* tv_gd.cu:27-47 -> TOMO_S_TV */
int tomo_tv_grad(tomo_engine *e, float eps);                            /* ctvlib.cpp:415-449 -> TOMO_S_GNORM */
/* the same pass, also the TV value of recon -> TOMO_S_TV (tv_gd.cu:177-183: the value "before descent") */
int tomo_tv_grad_tv(tomo_engine *e, float eps);
int tomo_tv_update(tomo_engine *e, float dPOCS, int clamp);             /* ctvlib.cpp:452-458 (+461 when clamp) */
/* the same step, also leaving the new first and last slice in two device planes (N*N floats each): what the ring
 * exchange of the slab-sharded descent sends before the next gradient pass, without a gather launch */
int tomo_tv_update_planes(tomo_engine *e, float dPOCS, int clamp, void *first_plane, void *last_plane);
/* one communication round per inner iteration instead of two (mpi_ctvlib.cpp:400-422 exchanges the slices, :455 reduces the
 * norm): the norm pass also leaves the gradient's first / last slice in device buffers; they travel with the all-reduce of
 * sum g^2, and every rank advances its halo planes itself (same expression as the update pass, same bits) */
int tomo_tv_grad_planes(tomo_engine *e, float eps, int with_tv, void *g_first, void *g_last);
int tomo_tv_halo_apply(tomo_engine *e, float dPOCS, int clamp, const void *g_lo, const void *g_hi);
/* the same step, also ||recon_new - track_vol||^2 -> scalar `slot` and track_vol = recon_new (sim_ASD.py:86-88) */
int tomo_tv_update_tracked(tomo_engine *e, float dPOCS, int clamp, int track_vol, int slot);
int tomo_fgp_begin(tomo_engine *e);                                     /* tv_fgp.cu:216-227 */
int tomo_fgp_begin_vol(tomo_engine *e, int vol);                        /* one element of cuda_tv_fgp_4D (chemistry/.../tv_fgp.cu:192) */
int tomo_fgp_obj(tomo_engine *e, float lambda);                         /* :44-65 + :143-154 */
int tomo_fgp_grad(tomo_engine *e, float lambda);                        /* :67-115 */
int tomo_fgp_end(tomo_engine *e, int iters);                            /* :272 */
/* Fused FGP iteration, step form: Obj + nonneg + Grad + Proj of tv_fgp.cu:244-268 in one pass (28 instead of 48
 * bytes per voxel) with ONE ring exchange per iteration when the volume is slab-sharded.  Caller-owned device planes
 * (Nray*Nray floats each): lo = P1 of the slice below this slab; hi = 4 planes {A, P1, P2, P3} of the slice above;
 * send_first = the same 4 planes of this slab's first slice and send_last = P1 of its last slice, written by
 * begin (A) and by every step (P) -- the caller exchanges send_last -> next.lo and send_first -> prev.hi before each
 * step and before end (tv_fgp.cu:57,81 need exactly these neighbours; mpi_ctvlib.cpp:400-422 is the reference ring). */
int tomo_bind_fgp_halo(tomo_engine *e, void *lo, void *hi, void *send_first, void *send_last);
/* The two-slice-deep set, for TWO fused iterations per pass on slabs (round 6; tv_fgp.cu:57,81 name the neighbours): lo 5 planes
 * [P1(-1), A(-1), P2(-1), P3(-1), P1(-2)], hi 8 planes [A, P1, P2, P3](nx), [A, P1, P2, P3](nx + 1), send_first 8 planes (my slices
 * 0 and 1: [A, P1, P2, P3] each), send_last 5 planes [P1(nx-1), A(nx-1), P2(nx-1), P3(nx-1), P1(nx-2)].  The one-deep planes above
 * are the prefixes (1 / 4 / 4 / 1 planes), so tomo_fgp_fused_step and _end run on the same buffers. */
int tomo_bind_fgp_halo2(tomo_engine *e, void *lo, void *hi, void *send_first, void *send_last);
int tomo_fgp_fused_begin(tomo_engine *e, int vol);
int tomo_fgp_fused_step(tomo_engine *e, float lambda, int first_iteration);
int tomo_fgp_fused_step2(tomo_engine *e, float lambda, int first_iteration);   /* TWO iterations in one pass (P stays on chip between
                                                                                * them; the same bits as two steps).  On a slab of a sharded
                                                                                * volume: tomo_bind_fgp_halo2, one two-deep exchange before
                                                                                * it, at least two slices on every slab of the ring */
int tomo_fgp_fused_end(tomo_engine *e, float lambda);                   /* the last iteration: D over the input volume */
int tomo_fgp_fused_last(tomo_engine *e, float lambda, int first_iteration);   /* one more step AND the end in one pass (the same bits);
                                                                               * whole-volume slabs only */

/* whole-call forms for a single slab (= the reference's single-GPU calls) */
int tomo_tv(tomo_engine *e, int vol, float eps);                        /* tomoengine.cpp:439-442 tv_3D -> TOMO_S_TV */
int tomo_tv_gd(tomo_engine *e, int ng, float dPOCS, float eps);         /* :445 tv_gd_3D; TV before descent -> TOMO_S_TV */
int tomo_tv_gd_tracked(tomo_engine *e, int ng, float dPOCS, float eps, int track_vol, int slot); /* last step tracked */
int tomo_tv_fgp(tomo_engine *e, int iters, float lambda);               /* :448-450 tv_fgp_3D; TV of input -> TOMO_S_TV */
int tomo_tv_fgp_vol(tomo_engine *e, int vol, int iters, float lambda);  /* multimodal.cpp:497 tv_fgp_4D, one element */

/* ---- multimodal (ChemicalTomo) element-wise steps ---------------------------------------------------------
 * Two engines with the same slab shape on one device and stream: `ce` carries the chemical geometry and the
 * per-element tomograms, `he` the HAADF geometry, the model volume Sigma*x^gamma and the SIRT-updated model.
 * Sigma (fusion_helper.py:5-32) is pixel-diagonal with one weight per element, passed as w[nel]. */
int tomo_get_stream(tomo_engine *e, void **hip_stream);
int tomo_mm_model(tomo_engine *ce, const int32_t *xvols, int nel, const float *w, float gamma, tomo_engine *he,
                  int model_vol);                                       /* multimodal.cpp:425-427 */
int tomo_mm_update(tomo_engine *ce, const int32_t *xvols, const int32_t *uvols, int nel, const float *w, float gamma,
                   float lamC_over_L, float lamH, tomo_engine *he, int upd_vol, int model_vol); /* :435-438, :471-476 */

/* Engine options (A/B switches for measurement; defaults are the fast paths):
 *   "sart_fused" (1): run a SART sweep as FP(a0), [BP(a_k)+FP(a_k+1)] fused steps, BP(a_last) instead of separate
 *                     FP/BP launches per angle (same arithmetic per voxel; 8 instead of 12 bytes per voxel-angle)
 *   "tv_lds" (1):     TV gradient kernel: 1 = register march (one wave = 8 z-columns x 64 slices, rows in registers,
 *                     slice neighbours by DPP; bit-identical to the LDS march and 13 % faster), 8 / 16 = LDS march with that
 *                     many z-columns per workgroup, 0 = direct-global stencil
 *   "tv_recompute" (1): a tv_gd inner iteration as "norm pass (sum g^2, nothing stored) + update pass that re-evaluates g and writes
 *                     x_new into a second buffer" instead of "gradient pass (store g) + update pass (read x, g; write x)": one volume
 *                     write instead of two -- HBM writes are the scarce resource (3 TB/s against ~6 for reads); bit-identical
 *   "fgp_fused" (1):  one fused kernel per FGP iteration (single slab)
 *   "fgp_pair" (1):   ... and two iterations per pass where the slab is the whole volume (k_fgp_fused2: 28 bytes per voxel for two
 *                     iterations; the same bits); 0 = one iteration per pass
 *   "art_chain" (1):  tomo_art in natural row order as per-angle FP + ray recurrence + BP (k_art_chain) instead of
 *                     row-by-row steps (k_art, which still serves tomo_art_order with a permutation)
 *   "art_tile" (1):   the chained ART sweep as fused tile steps (BP of the previous angle + FP of the next in k_sart_tile's ART
 *                     form, k_art_chain between them) instead of k_fp_rows + k_art_chain + k_bp_art per angle: 19.2 against
 *                     26.0 ms per sweep at 512^3 x 90
 *   "sart_tile" (1):  fused SART steps on image tiles streamed through LDS (k_sart_tile, in place) instead of the ray-walk
 *                     form (k_sart_seg); equal at 512 slices per GPU, 14-18 % faster on slabs of <= 128 slices
 *   "sart_streams" (0): the SART sweep of a slab as two sub-slabs (64-slice chunks, equal halves) on two streams, the second
 *                     chain enqueued by a second host thread: slices are independent, each stream fills the other's launch
 *                     boundaries and residual-row kernels.  2 = whenever the slab has two chunks, 1 = never, 0 = automatic:
 *                     equal halves and not when a pixel's row of slices is a multiple of 4 KB (measured per ASD-POCS step:
 *                     -2.6 / -3.3 / -5.5 / -5.3 % at 128 / 256 / 512 / 768 slices, +13 % at 1024).  A sub-slab runs its per-row
 *                     kernels at the widest vector that fits its chunks; results are bit-identical to the one-chain sweep
 *   "sart_skip_same" (1): k_sart_tile (in place) leaves out the store of a 256-byte piece (pixel x 64 slices) whose bits did
 *                     not change (clamped zeros, zero residuals): same memory image, fewer HBM writes; gain depends on the data
 *   "sart_nt" (-1):   cache policy of k_sart_tile's voxel accesses: 1 = streamed (non-temporal loads, write-through
 *                     non-temporal stores: -9 % per sweep at 512^3), 0 = plain, -1 = streamed when the slab exceeds 192 MB
 *                     (a slab that fits the Infinity Cache is faster with plain accesses)
 *   "sart_coop" (0):  1 = the residual rows of an angle are formed inside the next tile step by its first workgroups
 *                     instead of one k_resid_finish launch per angle (bit-identical; measured no faster);
 *                     "sart_coop_spin" (4096): polls before a workgroup forms its rows itself (< 0: always, for tests)
 *   "tv_yseg" (0):    rows a wave of the TV register march walks; 0 = 32, shortened on thin slabs until >= 8192 waves
 *   "tv_march4" (1):  norm / update passes of tv_gd by k_tv_march4 (row slots renamed over a 4x unrolled loop, packed edge
 *                     values: no register rotation; -10 % per inner iteration, bit-identical) instead of k_tv_grad_reg
 *   "fp_tile" (1):    all-angle forward projection from LDS-resident image tiles (k_fp_tile + k_fp_tile_reduce);
 *                     0 = ray-driven form selected by "fp_all_lpr"
 *   "fp_tile_scratch_mib" (8192): cap of the tile projector's partial-sum scratch; a larger volume is projected in
 *                     several passes over groups of 64-slice chunks (set before the first projection)
 *   "fp_tile_chunks_per_pass" (0): that number of 64-slice chunks per pass instead (0 = derived from the cap)
 *   "bp_tile" (1):    all-angle back-projection from LDS-staged residual-row windows (k_bp_tile; bit-identical to the
 *                     pixel-driven k_bp_all it replaces; used when every tile's ray window fits, else k_bp_all)
 *   "bp_list" (1):    ... in its entry-list form (k_bp_list: one scalar-loaded entry per nonzero weight, accumulators picked by
 *                     the VGPR index mode; same bits) when the engine built the lists ("bp_list_ready" of tomo_get_option) and
 *                     the slab is a whole number of 128-slice pieces; 0 = k_bp_tile
 *   "fp_all_lpr" (16): ray-driven all-angle forward projection with 16 lanes x float4 per ray and 64-slice chunks
 *                     (0 = wide form)
 *   "fp_reuse" (1):   a tomo_sirt / tomo_sirt_data / tomo_cgls call whose volume is exactly what the model sinogram was last
 *                     projected from (tomo_forward_projection into TOMO_SINO_G, tomo_data_distance_sq; tracked by slot and
 *                     write-version, inherited by tomo_copy_volume) starts from that sinogram instead of projecting again:
 *                     bit-identical.  Also gates tomo_fista_project_yk.  0 = every projection recomputed
 *   "fp_strip" (1):   all-angle forward projection by sheared strips with the ray sums resident in registers (k_fp_strip) when the
 *                     engine built the strip tables (large slabs; "fp_strip_ready" of tomo_get_option); setting "fp_tile"
 *                     explicitly also sets "fp_strip" = 0, so that the tile / ray-driven forms can be selected
 *   "fp_list" (1):    ... the strips as wave-uniform entry lists (k_fp_list: one angle per wave, one accumulator per ray picked by
 *                     the VGPR index mode) when the engine built them ("fp_list_ready": where the strips are built and a tile's
 *                     work spreads evenly enough over the waves; TOMO_FP_LIST = 0 / 1 overrides) and the slab is a whole number
 *                     of 128-slice pieces; 0 = k_fp_strip
 *   "fp_tile_pipe" (0): experimental, P >= 2: the tile projector runs as P groups of 64-slice chunks, the reduce pass of one
 *                     group on a second stream beside the tile pass of the next (no gain measured)
 *   "sart_resident" (-1): the SART sweep as ONE launch of the volume-resident kernel (k_sart_resident: a 64-slice chunk of the whole
 *                     image stays in vector registers over all angles, the workgroups exchange ray sums).  -1 = automatic: wherever
 *                     the tables exist ("sart_resident_ready": N a multiple of 8, at most one 32 x 32 tile per CU, ray windows that
 *                     fit, the tile tables of the streamed chain built too); after a sweep that could not finish (below) the form sits
 *                     out 1, 2, 4 ... 64 sweeps before it is tried again.  0 = never (the streamed chain k_sart_tile), 1 = insist:
 *                     an error where the tables do not exist, no sitting out.  Setting it clears the sitting-out state.
 *                     FAIL-SAFE: every workgroup of the launch must be on the chip at once.  Where that cannot happen (another
 *                     process or a long kernel of the caller holds CUs) the waits run out, the affected 64-slice chunks are NOT
 *                     stored -- a chunk is stored by all of its workgroups or by none -- and tomo_sart / tomo_sart_tracked sweep
 *                     them with the streamed chain before they return: the call returns TOMO_OK with the sweep done (the
 *                     reference's sweep either happens or errors before touching recon, tomoengine.cpp:162-179).  The call waits
 *                     for the launch on the host (that is what knowing costs: ~10 us per sweep)
 *   "sart_resident_spin" (2097152): polls (~1 us each) a wait of the resident sweep makes before it gives up; < 0 = the default,
 *                     0 = every wait gives up at its first unsuccessful look (tests)
 *   "sart_resident_test_fail" (0): tests: chunk + 1 whose first workgroup refuses to commit (that chunk goes to the streamed chain) */
int tomo_set_option(tomo_engine *e, const char *name, int value);
/* read back a switch, or a fact about the engine: "fp_strip", "fp_list", "fp_tile", "bp_tile", "bp_list", "fgp_pair", "fp_reuse", "sart_tile", and
 * "fp_list_ready" (1: the list form of the strips was built),
 * "bp_list_ready" (1: the entry lists of k_bp_list were built: every tile's ray windows fit and there are at most 192 angles),
 * "fp_strip_ready" (1: the sheared-strip tables were built at creation -- by the slab-size rule or TOMO_FP_STRIP=1 -- so
 * "fp_strip" = 1 takes effect), "fp_strip_slots" (accumulator slots per lane group the strip kernel runs with),
 * "comm_rounds" (RCCL rounds -- one ncclGroup or one lone collective each -- this engine has enqueued since creation),
 * "rccl_version" (ncclGetVersion's code of the librccl the library opened, e.g. 22707; 0 before the first communicator),
 * "sart_resident", "sart_resident_ready" (the tables of the volume-resident SART sweep were built), "sart_resident_active"
 * (ready and not switched off), "sart_resident_spin", "sart_resident_fallbacks" (sweeps of this engine that needed the streamed chain
 * for chunks the resident launch did not store), "sart_resident_fallback_chunks" (how many 64-slice chunks that were),
 * "sart_resident_skip" (sweeps the resident form still sits out), "sart_resident_last_code" (what the last give-up waited for:
 * 1 residual rows, 2 tile sums, 0 the commit),
 * "table_kib" (device memory of the tables built at creation: every kernel family's, eagerly), "create_ms" (what creation took),
 * "form_fp" / "form_bp" / "form_sart": which kernel family the all-angle forward projection, the all-angle back projection and the
 * SART sweep of THIS engine run as under the options in force (tomo_form; one function of the engine decides it for the launchers
 * and for this query: tomo_engine.hip select_forms) */
typedef enum { TOMO_FORM_FP_ROWS = 0, TOMO_FORM_FP_TILE = 1, TOMO_FORM_FP_STRIP = 2, TOMO_FORM_FP_LIST = 3,
               TOMO_FORM_BP_ALL = 0, TOMO_FORM_BP_TILE = 1, TOMO_FORM_BP_LIST = 2,
               TOMO_FORM_SART_ANGLE = 0, TOMO_FORM_SART_TILE = 1, TOMO_FORM_SART_RESIDENT = 2 } tomo_form;
int tomo_get_option(tomo_engine *e, const char *name, int *value);
/* ---- native communicator: the slab-sharded path over RCCL on the engine's own stream ---------------------------------------
 * Replaces, for a C / C++ host as for the Python one, the MPI calls of the reference's sharded CPU engine (mpi_ctvlib.cpp:400-422
 * ring exchange of boundary slices, :455 / :547 MPI_Allreduce of the norms) and the OpenMP-over-GPUs loop of multigpuengine.cpp:
 * one engine per rank holding rank's slab (tomo_create with the slab's slice count), one communicator shared by the rank's
 * engines.  librccl is opened with dlopen on first use.  Rank 0 makes the 128-byte id and hands it to the others by any means
 * (MPI_Bcast, torch.distributed.broadcast, a file); tomo_comm_init is collective over the group. */
int tomo_comm_unique_id(void *id128);
int tomo_comm_init(tomo_engine *e, const void *id128, int world, int rank);   /* also sets the slab's global-edge flags */
int tomo_comm_share(tomo_engine *e, tomo_engine *other);                       /* another engine of the same rank and device */
int tomo_comm_destroy(tomo_engine *e);
int tomo_comm_info(tomo_engine *e, int *world, int *rank);                     /* world = 0: no communicator */
/* boundary slices of a field into the ring neighbours' halo planes: pack + ONE group of 2 sends and 2 receives */
int tomo_comm_exchange_halo(tomo_engine *e, int field);
/* every scalar slot summed over the ranks (into a copy: the buffer keeps the slab's partial sums), read back; and the
 * non-blocking form collected by tomo_scalars_snapshot_read */
int tomo_comm_read_scalars(tomo_engine *e, double *out, int count);
int tomo_comm_scalars_snapshot(tomo_engine *e);
/* slab-sharded tv_gd / tv_gd_tracked (track_vol < 0: plain), whole call: per inner iteration ONE group {all-reduce of sum g^2 +
 * the gradient's boundary planes to both neighbours}; every rank advances its halo planes itself.  TOMO_S_TV keeps the slab's
 * share of the TV value before descent. */
int tomo_comm_tv_gd(tomo_engine *e, int ng, float dPOCS, float eps, int track_vol, int slot);
/* the plane exchange between two fused FGP iterations (buffers of tomo_bind_fgp_halo) */
int tomo_comm_fgp_exchange(tomo_engine *e);
/* ... and before a PAIR of them (tomo_fgp_fused_step2 on slabs): the two-slice-deep planes of tomo_bind_fgp_halo2, one round per two iterations */
int tomo_comm_fgp_exchange2(tomo_engine *e);

/* launch chains a SART / ART sweep of this engine's slab runs as under the current "sart_streams" (1 = one chain on the
 * engine's stream; 2..4 = that many sub-slabs of 64-slice chunks on their own streams).  What the reference hides inside
 * ASTRA's run(Nproj*nIter) (tomoengine.cpp:162-179); bench.py derives the bytes one launch moves from it. */
int tomo_sart_chain_count(tomo_engine *e, int *count);

/* ---- measurement hooks (bench.py) -------------------------------------------------------------------
 * While enabled, every launch of the named kernel is bracketed by HIP events on the engine's stream (on = N > 1: every N-th
 * launch only -- an event pair costs about 3 us of command-processor time, 0.7 ms per ASD-POCS step when all ~210 launches of
 * a step are bracketed); tomo_profile_read synchronises, returns the count and summed device time of the bracketed launches,
 * and resets the log. */
enum tomo_kernel_id { TOMO_K_BP_ANGLE = 0, TOMO_K_FP_ANGLE = 1, TOMO_K_TV_GRAD = 2, TOMO_K_TV_UPDATE = 3,
                      TOMO_K_FGP_OBJ = 4, TOMO_K_FGP_GRAD = 5 /* also the fused FGP iteration */, TOMO_K_SART_FUSED = 6,
                      TOMO_K_FP_TILE = 7, TOMO_K_BP_TILE = 8, TOMO_K_FP_REDUCE = 9,
                      TOMO_K_SART_RESIDENT = 10 /* one launch = one whole SART sweep of the slab (volume-resident form) */ };
int tomo_profile_enable(tomo_engine *e, int kernel, int on);
int tomo_profile_read(tomo_engine *e, int kernel, int64_t *launches, double *total_ms);
/* the same, also busy_ms = time during which at least one launch of the kernel was executing (union of the launch
 * intervals): the SART sweep runs as two sub-slabs on two streams ("sart_streams"), so two launches of a kernel overlap */
int tomo_profile_read2(tomo_engine *e, int kernel, int64_t *launches, double *total_ms, double *busy_ms);
/* launch intervals [t0, t1) in ms on the time base of ref_engine's log (engines of a slab group); cap = 0 only counts */
int tomo_profile_intervals(tomo_engine *e, int kernel, tomo_engine *ref_engine, double *t0, double *t1, int cap, int *count);

#ifdef __cplusplus
}
#endif
#endif /* TOMO_HIP_H */
