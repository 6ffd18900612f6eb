// tomo_engine.hip -- engine state + C ABI (include/tomo_hip.h) over the kernels in kernels.hip.h.
//
// Mirrors the method surface of the reference's `tomoengine` (tomofusion/gpu/utils/tomoengine.{hpp,cpp}) and
// `ctvlib` (tomofusion/cpu/utils/ctvlib.{hpp,cpp}); each entry point cites the lines it replaces in the header.
// Unlike the reference (host-resident volume, cudaMalloc + H2D + D2H inside every call) all fields stay
// resident in HBM for the life of the engine.
#include "../../include/tomo_hip.h"
#include "kernels.hip.h"
#include "sart_resident.hip.h"
#include "sysmat.h"
#include "resident.h"

#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A ROCm install without the RCCL development headers still builds the engine: librccl is only ever opened with dlopen
// (rccl_load), so the handful of opaque types and constants of its C API that the tomo_comm_* entries use are restated here
// (nccl.h: ncclUniqueId is 128 opaque bytes; ncclFloat32 = 7, ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0).
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

using namespace tomo;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(call)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (call);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return fail(TOMO_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e));                 \
    } while (0)
#define LAUNCHCHK() HIPCHK(hipGetLastError())
#define NEED(e) do { if (!(e)) return fail(TOMO_ERR_ARG, "null engine"); if ((e)->geometry_released) return fail(TOMO_ERR_STATE, "engine geometry was released"); HIPCHK(hipSetDevice((e)->device)); } while (0)
enum { PROF_MAX_KERNELS = 12, PROF_MAX_EVENTS = 1 << 19 };   // 262144 launches per kernel id between two reads

struct ProfSlot {
    bool on = false;
    unsigned stride = 1, seen = 0;   // every stride-th launch is bracketed (an event pair costs ~3 us of command-processor time)
    std::vector<hipEvent_t> ev;  // pairs
    hipEvent_t ref = nullptr;    // recorded when the log is switched on: launches on different streams share this time base
    size_t used = 0, dropped = 0;   // dropped: launches that could not be recorded (tomo_profile_read reports them)
};

// A persistent host thread that enqueues the launch chain of one sub-slab (run_chains).  Round 2 started and joined a std::thread
// per sweep (50-100 us of host time on the path of a 4 ms step of a 64-slice slab); this one is started once per
// engine and sleeps on a condition variable between sweeps.
struct ChainHelper {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, stop = false;
    void start(int device)
    {
        th = std::thread([this, device]() {
            (void)hipSetDevice(device);
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return has_job || stop; });
                if (stop) return;
                has_job = false;
                lk.unlock();
                job();
                lk.lock();
                done = true;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<void()> f)
    {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(f);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
    }
    ~ChainHelper()
    {
        if (th.joinable()) {
            { std::lock_guard<std::mutex> lk(mu); stop = true; }
            cv.notify_all();
            th.join();
        }
    }
};

struct CommRef;
struct tomo_engine;
static void comm_release(tomo_engine *e);
struct tomo_engine {
    int nx = 0, n = 0, np = 0, sx = 0, sxc = 0, vec = 1, device = 0;   // sx = row pitch, sxc = computed width
    int64_t npix = 0, nrows = 0, nnz = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // tables
    uint32_t *d_rptr = nullptr;
    uint2 *d_rent = nullptr;
    float *d_rowsum = nullptr, *d_rowinner = nullptr, *d_colsum_all = nullptr, *d_rowcross = nullptr;
    bool art_chain_ok = false;
    int art_chain = 1;                            // natural-order ART: per-angle FP + ray recurrence + BP instead of row steps
    CellD *d_cell = nullptr;
    uint32_t *d_wptr = nullptr;                   // walk lists of the fused SART step
    uint2 *d_went = nullptr;
    float lipschitz = 0.f, lipschitz_cimmino = 0.f;
    int sart_fused = 1;                      // 1: SART sweep as a chain of fused BP+FP steps; 0: separate FP and BP per angle
    int tv_lds = 1, fp_all_lpr = 16;
    // tv_recompute: a tv_gd inner iteration as "norm pass (no store) + recompute-and-update pass into a second buffer" instead
    // of "gradient pass (store g) + update pass": one volume write instead of two (HBM writes are the scarce resource)
    int tv_recompute = 1, tv_tz = 8;              // tv_tz: z-columns per wave of the register march (8 or 4)
    int tv_yseg = 0;                              // rows per wave of the register march; 0 = by slab size (tv_rows_per_wave)
    int tv_march4 = 1;                            // norm / update passes by k_tv_march4 (no row rotation) instead of k_tv_grad_reg
    int tv_halo_fold = 1;                         // slab-sharded descent: the update pass advances the halo planes itself (no k_halo_apply launch)
    int gnorm_slot = TOMO_S_GNORM;                // scalar slot the TV update takes ||g||^2 from (slab groups: TOMO_S_GNORM_ALL)
    double *gnorm_override = nullptr;             // tomo_comm_tv_gd: the all-reduced sum g^2 (the slot itself keeps the slab's partial sum)
    hipEvent_t ev_peer = nullptr;                 // tomo_wait_for: "everything enqueued on this engine so far"
    float tv_last_eps = 1e-6f;
    float *tv_alt = nullptr, *halo_lo_alt = nullptr, *halo_hi_alt = nullptr;
    // fp_all_lpr: all-angle FP: lanes per ray of the narrow-chunk form (0 = wide form)
    SegItemD *d_seg_exec = nullptr;
    std::vector<uint32_t> h_seg_exec_ptr;
    uint32_t *d_row_first = nullptr, *d_row_nseg = nullptr;
    float *seg_partial = nullptr;
    uint32_t max_items = 0;
    float *sart_alt = nullptr;                    // ping-pong partner of the volume being swept
    // tile-stationary all-angle FP (k_fp_tile): tables, partial-sum scratch (one per stream that can run it)
    int fp_tile = 1, ft_tiles_z = 0, ft_ntiles = 0;
    uint32_t ft_nseg = 0;
    uint32_t *d_ft_slot_ptr = nullptr, *d_ft_slot_seg0 = nullptr, *d_ft_rsptr = nullptr, *d_ft_rsidx = nullptr;
    uint2 *d_ft_tent = nullptr;
    float *ft_part = nullptr, *ft_part_aux = nullptr;
    // sheared-strip all-angle FP (k_fp_strip, round 4): ray sums resident in registers, ~5 partial sums per ray instead of ~27
    int fp_strip = 1, fs_nitems = 0, fs_kused = 0, fs_ncp = 0;
    bool fs_ok = false, attr_fs = false;
    uint32_t fs_nseg = 0;
    FsItemD *d_fs_items = nullptr;
    int *d_fs_orient = nullptr, *d_fs_shift = nullptr;
    uint4 *d_fs_cnt = nullptr;
    uint32_t *d_fs_gstart = nullptr, *d_fs_gseg0 = nullptr, *d_fs_rsptr = nullptr, *d_fs_rsidx = nullptr;
    uint2 *d_fs_ent = nullptr;
    float *d_fs_zero = nullptr;                   // 256 bytes of zeros: what a strip tile's pixels outside the image are staged from
    float *fs_part = nullptr, *fs_part_aux = nullptr;
    // ... and as wave-uniform entry lists (k_fp_list) when the slab is a whole number of 128-slice pieces
    int fp_list = 1, fl_nitems = 0, fl_ncp = 0;
    bool fl_ok = false, attr_fl = false;
    uint32_t fl_nseg = 0;
    FlItemD *d_fl_items = nullptr;
    int *d_fl_orient = nullptr, *d_fl_shift = nullptr;
    uint2 *d_fl_ent = nullptr, *d_fl_fent = nullptr;
    uint32_t *d_fl_ptr = nullptr, *d_fl_fptr = nullptr, *d_fl_rsptr = nullptr, *d_fl_rsidx = nullptr;
    float *d_fl_zero = nullptr, *fl_part = nullptr, *fl_part_aux = nullptr;
    // all-angle FP as a two-stage pipeline over groups of 64-slice chunks ("fp_tile_pipe"): [0] main stream, [1] second stream
    int fp_tile_pipe = 0;   // off: measured (round 3) 1.50 vs 1.52 ms at 512^3 x 90, 1.91 vs 1.83 ms at 128 x 1024^2 x 120, 0.127 vs 0.154 ms at 256^3 x 60
    hipStream_t fp_red_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_fp_tile[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, ev_fp_red[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    bool attr_fp = false, attr_bp = false, attr_st = false;   // dynamic-LDS limits raised on this engine's device
    // The SART chain of one slab is a string of dependent launches (tile step -> residual finish -> tile step ...): ~5.7 us of
    // idle chip after each and a tail of partly filled CUs at the end of each.  Slices are independent, so the sweep CAN run as
    // two sub-slabs on two streams, each filling the other's gaps ("sart_streams" = 2).  Measured (round 2, per sweep): two
    // separate 256-slice engines side by side 18.0 ms against 20.7 for one 512-slice engine -- but two sub-slabs of ONE slab
    // interleave inside every pixel row (1 KB of every 2 KB) and reached only 19.4 ms at 512 slices and 51.3 against 40.6 ms
    // at 1024: two kernels striding over alternate halves of the same rows collide in the memory system.  With the streamed tile
    // accesses and the skipped stores of round 2 the gain at 512 slices grew to 5.5 % of the ASD-POCS step (23.4 -> 22.1 ms),
    // so the default is now 0 = automatic (the rule is in sart_impl); the sub-slabs split at 64-slice chunks and run their
    // per-row kernels at the widest vector that fits (k_bp_angle writes its roundings out, so every width gives the same bits).
    // sub_c0 / sub_nc: the 64-slice chunk range the launch helpers address (0 / 0 = whole slab).
    int sart_streams = 0, sub_c0 = 0, sub_nc = 0;
    static constexpr int MAX_CHAINS = 4;
    hipStream_t sub_stream[MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_sfork = nullptr, ev_sjoin[MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};
    std::unique_ptr<ChainHelper> chain_helper[MAX_CHAINS];   // [u]: the thread that enqueues chain u (u >= 1), started on first use
    int sart_tile = 1;                            // fused SART step on streamed image tiles (k_sart_tile) when the geometry allows it
    bool st_ok = false;
    int st_ntiles = 0, st_tiles_z = 0;
    uint32_t st_max_ids = 0;
    uint4 *d_st_cell = nullptr;
    uint32_t *d_st_win = nullptr, *d_st_segid = nullptr, *d_st_row_first = nullptr, *d_st_row_nseg = nullptr;
    uint2 *d_st_seg = nullptr, *d_st_ent = nullptr;
    float *st_partial = nullptr, *st_partial2 = nullptr;   // tile partial sums; the second buffer of the cooperative chain (links alternate)
    uint32_t *st_flags = nullptr;                          // [ray of an angle][64-slice chunk]: epoch of the launch that published the residual row
    uint32_t st_epoch = 0;
    // "sart_coop" = 1: residual rows inside the tile step (k_sart_tile COOP) instead of one k_resid_finish launch per angle.
    // Off by default -- measured (round 2, 512^2 x 90): the step kernel grows by what the dropped launch and its two
    // boundaries cost (221 vs 210 us at 512 slices: 27.3 ms per ASD-POCS step either way) and by more on thin slabs
    // (64 slices: 37.0 vs 30.8 us, 5.25 vs 4.87 ms per step): the first workgroups cannot start their voxel update before
    // the rows exist, so the reduction is serial either way and only moves inside the launch.
    int sart_coop = 0, sart_coop_spin = 4096, st_resident = 0;
    // "sart_resident" (round 5): the sweep as ONE launch of k_sart_resident -- a 64-slice chunk of the whole image stays in the chip's
    // vector registers over all angles of the sweep and the workgroups exchange only ray sums (sart_resident.hip.h).  -1 = automatic
    // (whenever the tables exist: N a multiple of 8 with at most one 32 x 32 tile per CU, a matrix whose ray windows fit), 0 = never
    // (the streamed tile steps), 1 = insist (an error where the tables do not exist).
    int sart_resident = -1;
    bool rs_ok = false;
    int rs_ntiles = 0, rs_tiles = 0, rs_rpt = 0, rs_groups = 0, rs_cus = 0;
    RsHdrD *d_rs_hdr = nullptr;
    uint4 *d_rs_cell = nullptr;
    uint2 *d_rs_ts = nullptr;
    uint16_t *d_rs_rl = nullptr;
    rs_u64 *rs_pb = nullptr, *rs_rb = nullptr;     // granules {value, tag}: tile sums, residual rows
    size_t rs_pb_bytes = 0, rs_rb_bytes = 0;
    int *d_rs_angs = nullptr;                      // angle of every step of the sweep in flight
    size_t rs_angs_cap = 0;
    std::vector<int> rs_angs_host;                 // (what d_rs_angs holds: an unchanged sequence is not uploaded again)
    int *rs_abort = nullptr, *d_rs_abort = nullptr; // pinned host word / device word a workgroup sets when a bounded spin gave up
    // fail-safe (round 6): a chunk is stored by all of its workgroups or by none (rs_commit); d_rs_commit = the chunks' commit words,
    // rs_done = pinned [chunk]: the sequence number of the launch that stored it.  Chunks that did not commit are swept by the streamed
    // chain, and the resident form then sits out rs_skip sweeps (doubling up to 64 while the failures go on; "sart_resident" = 1 insists)
    unsigned *d_rs_commit = nullptr;
    int *rs_done = nullptr;
    uint32_t rs_seq = 0, rs_commit_base = 0;
    bool rs_commit_dirty = false;
    size_t table_bytes = 0;                        // device bytes of the tables built at creation (everything but volumes, sinograms, halos)
    double create_ms = 0.0;                        // wall clock of the creation (matrix, tables, uploads)
    int rs_fallbacks = 0, rs_fallback_chunks = 0, rs_skip = 0, rs_backoff = 0, rs_last_code = 0, rs_test_fail = 0;
    uint32_t rs_epoch = 0;                         // tags handed out so far (a granule's tag is unique per sweep, chunk round and step)
    uint32_t rs_spin_limit = 1u << 21;
    int art_tile = 1;                              // chained ART sweep as fused tile steps (k_sart_tile ART) instead of k_fp_rows + k_bp_art per angle
    int sart_skip_same = 1;                        // k_sart_tile stores only the 256-byte pieces whose bits changed (in place)
    int sart_nt = -1;                              // tile accesses: -1 streaming form by slab size (slab_streams), 0 plain, 1 streaming
    int bp_tile = 1;                              // tile-stationary all-angle BP (k_bp_tile) when the geometry allows it
    int bp_list = 1;                              // ... in its entry-list form (k_bp_list) when the slab is whole pairs of 64-slice chunks
    bool attr_bp2 = false, bl_ok = false;
    int bp_list_band = 0;                         // k_bp_list: 1 = an XCD owns a contiguous band of tiles (see the kernel)
    uint4 *d_bl_ent = nullptr;                    // k_bp_list: entry batches and the first batch of every (tile, stage, wave) list
    uint32_t *d_bl_ptr = nullptr, *d_bl_win = nullptr;
    int bl_tiles_z = 0, bl_ntiles = 0;
    bool fb_ok = false;
    uint4 *d_fb_cell = nullptr;
    uint32_t *d_fb_win = nullptr;
    int ft_ncp = 0, ft_ncp_forced = 0;            // slice chunks per pass (bounds the scratch); forced value for tests
    size_t ft_scratch_cap = (size_t)8 << 30;      // >= 4 chunks per pass up to 1024^2 x 120 (one pass measured 12 % faster than one chunk per pass)
    // fields
    float *vol[TOMO_VOL_SLOTS] = {};
    float *sino[TOMO_SINO_SLOTS] = {};
    int fgp_target = TOMO_VOL_RECON;
    int tv_target = TOMO_VOL_RECON;               // volume the tv_gd / tv_grad / tv_update forms act on (tomo_tv_set_target)
    float *cur_b = nullptr;
    float *cg_p = nullptr, *cg_z = nullptr, *cg_w = nullptr, *fbp_h = nullptr;   // CGLS direction / A^T r / A p; WBP kernel
    double *cg_sums = nullptr;                    // 2*sx per-slice sums
    double *cg_part = nullptr;                    // 1024 x sx: the workgroups' partial sums of a per-slice reduction
    float *cg_coef = nullptr;                     // sx per-slice coefficients                      // data sinogram of the SART call in progress
    float *tvg = nullptr;                         // TV gradient tensor; doubles as FGP "D"
    float *fgp_p[3] = {nullptr, nullptr, nullptr};
    float *fgp_q[3] = {nullptr, nullptr, nullptr};   // ping-pong partners for the fused FGP iteration
    int fgp_fused = 1;
    int fgp_pair = 1;                             // ... two iterations per pass (k_fgp_fused2) where the slab is not sharded
    float *stage = nullptr;
    size_t stage_bytes = 0;
    // scalars
    double *d_scal = nullptr, *d_scal_own = nullptr, *d_part = nullptr, *d_part_aux = nullptr, *d_part_tv = nullptr;
    double *h_snap = nullptr;          // pinned: tomo_scalars_snapshot
    hipEvent_t ev_snap = nullptr;
    bool snap_pending = false;
    bool part_open[3] = {false, false, false};   // main / aux / tv partial sums: a reduction is in flight (see part_begin)
    hipStream_t aux = nullptr;                    // second stream for work that is independent of the main sequence
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool async_pending = false;
    struct CommRef *comm = nullptr;               // native RCCL communicator (tomo_comm_init / tomo_comm_share); see the comm section
    int64_t comm_rounds = 0;                      // RCCL rounds (one ncclGroup or one lone collective) this engine has enqueued: bench.py's rccl_rounds_per_step
    float *comm_send_first = nullptr, *comm_send_last = nullptr, *comm_g_lo = nullptr, *comm_g_hi = nullptr;   // N*N planes
    float *comm_fgp = nullptr;                    // the engine's own planes of the fused FGP exchange when the host binds none
    double *comm_scal = nullptr;                  // TOMO_S_COUNT doubles: the all-reduced copy of the scalar buffer
    // "the model sinogram G is A * (volume v in its present state)": set by a plain projection into G (tomo_forward_projection,
    // tomo_data_distance_sq), inherited by a copy of v, dropped by anything else that touches G or writes v (fp_reuse, below)
    uint64_t vol_version[TOMO_VOL_SLOTS] = {};
    struct { int vol = -1; uint64_t ver = 0; } g_valid[2];
    int fp_reuse = 1;
    // FISTA: the projection of the extrapolated point by linearity (tomo_fista_project_yk)
    float *g_prev = nullptr;                      // A * (the iterate before the last Nesterov step)
    float *g_yk = nullptr;                        // A * yk formed by linearity; G itself stays A * recon (tomoengine.cpp:410-427,459)
    struct { bool valid = false; uint64_t ver = 0; } yk_claim;   // g_yk is A * (volume YK at this write-version)
    bool g_prev_valid = false, mom_p_ok = false;
    uint64_t g_prev_recon_ver = 0;
    struct { float beta = 0.f; uint64_t ver_recon = 0, ver_yk = 0, ver_old = 0; bool set = false; } mom;
    bool old_is_recon = false;                    // RECON_OLD's content is RECON's (tomo_fista_momentum; see get_vol)
    bool geometry_released = false;               // tomo_release_geometry: only tomo_adopt_volumes / tomo_destroy remain valid
    // halos
    float *halo_lo = nullptr, *halo_hi = nullptr, *halo_lo_own = nullptr, *halo_hi_own = nullptr;
    // planes of the fused slab-sharded FGP iteration (caller-owned device buffers, tomo_bind_fgp_halo)
    float *fgp_lo = nullptr, *fgp_hi = nullptr, *fgp_send_first = nullptr, *fgp_send_last = nullptr;
    bool fgp_planes2 = false;                     // the bound planes are the two-slice-deep set (5 / 8 / 8 / 5 planes: tomo_bind_fgp_halo2)
    int is_first = 1, is_last = 1;
    ProfSlot prof[PROF_MAX_KERNELS];
    std::mutex prof_mu;
    size_t vol_elems() const { return (size_t)npix * sx; }
    size_t sino_elems() const { return (size_t)nrows * sx; }
};

static inline double *gnorm_ptr(const tomo_engine *e) { return e->gnorm_override ? e->gnorm_override : e->d_scal + e->gnorm_slot; }

// (bytes handed out while an engine is being created are counted for it: tomo_get_option "table_kib")
static thread_local size_t *g_alloc_meter = nullptr;
static int dev_alloc(void **p, size_t bytes, bool zero, hipStream_t st)
{
    HIPCHK(hipMalloc(p, bytes ? bytes : 4));
    if (g_alloc_meter) *g_alloc_meter += bytes;
    if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes ? bytes : 4, st));
    return TOMO_OK;
}

// ---- a projection already in hand is not computed again -------------------------------------------------------------------
// The reference's drivers evaluate `data_distance()` after every SIRT / CGLS step (gpu/reconstructor.py:61-71, show_convergence
// defaults to True) and the next step starts by projecting the very same volume; multimodal::data_fusion projects the model
// volume for its cost and then starts a SIRT run from a copy of it (multimodal.cpp:452-470).  The engine remembers which volume
// (slot and write-version) the model sinogram G was projected from; a SIRT / CGLS call whose volume is still that one forms its
// first residual from G instead of projecting again -- the same kernels produced G, so the bits are the same ("fp_reuse" = 0
// switches it off).  Conservative by construction: every write-intent access of a volume (get_vol) bumps its version, every
// access of G through the slot accessors drops the claim, and only the two plain projections make it.
static void g_clear(tomo_engine *e) { e->g_valid[0].vol = e->g_valid[1].vol = -1; }
static void g_set(tomo_engine *e, int vol) { e->g_valid[0].vol = vol; e->g_valid[0].ver = e->vol_version[vol]; e->g_valid[1].vol = -1; }
static bool g_is_projection_of(const tomo_engine *e, int vol)
{
    if (!e->fp_reuse || vol < 0 || vol >= TOMO_VOL_SLOTS || !e->sino[TOMO_SINO_G]) return false;
    for (int k = 0; k < 2; ++k) if (e->g_valid[k].vol == vol && e->g_valid[k].ver == e->vol_version[vol]) return true;
    return false;
}

// the sinogram that holds A * (volume vol as it stands), or nullptr: G through its claim, or the extrapolated point's own buffer
static const float *projection_in_hand(const tomo_engine *e, int vol)
{
    if (g_is_projection_of(e, vol)) return e->sino[TOMO_SINO_G];
    if (e->fp_reuse && vol == TOMO_VOL_YK && e->g_yk && e->yk_claim.valid && e->yk_claim.ver == e->vol_version[TOMO_VOL_YK]) return e->g_yk;
    return nullptr;
}

// After a Nesterov step recon_old == recon (tomoengine.cpp:381-384 copies the prox result into both).  The step keeps that as a
// FLAG instead of a second store (old_is_recon: the logical content of RECON_OLD is RECON's; its own buffer is stale), so the
// step reads two volumes and writes one.  Whoever may WRITE recon, or touches recon_old, goes through get_vol, which first makes
// the copy real; readers of recon use get_vol_ro and leave the flag alone.
static int get_vol_ro(tomo_engine *e, int id, float **out)
{
    if (id < 0 || id >= TOMO_VOL_SLOTS) return fail(TOMO_ERR_ARG, "bad volume id");
    if (id == TOMO_VOL_RECON_OLD && e->old_is_recon) id = TOMO_VOL_RECON;     // read-only view of the same content
    if (!e->vol[id]) {
        int rc = dev_alloc((void **)&e->vol[id], e->vol_elems() * sizeof(float), true, e->stream);
        if (rc) return rc;
    }
    *out = e->vol[id];
    return TOMO_OK;
}

static int get_vol(tomo_engine *e, int id, float **out)
{
    if (id < 0 || id >= TOMO_VOL_SLOTS) return fail(TOMO_ERR_ARG, "bad volume id");
    ++e->vol_version[id];                               // the caller may write it
    if (e->old_is_recon && (id == TOMO_VOL_RECON || id == TOMO_VOL_RECON_OLD)) {
        e->old_is_recon = false;
        float *src, *dst; int rc;
        if ((rc = get_vol_ro(e, TOMO_VOL_RECON, &src)) || (rc = get_vol_ro(e, TOMO_VOL_RECON_OLD, &dst))) return rc;
        HIPCHK(hipMemcpyAsync(dst, src, e->vol_elems() * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    }
    return get_vol_ro(e, id, out);
}

static int get_sino(tomo_engine *e, float **slot, float **out)
{
    if (slot == &e->sino[TOMO_SINO_G]) g_clear(e);      // whoever asks for G may overwrite it
    if (!*slot) {
        int rc = dev_alloc((void **)slot, e->sino_elems() * sizeof(float), true, e->stream);
        if (rc) return rc;
    }
    *out = *slot;
    return TOMO_OK;
}

static int sino_slot(tomo_engine *e, int id, float **out)
{
    if (id < 0 || id >= TOMO_SINO_SLOTS) return fail(TOMO_ERR_ARG, "bad sinogram id");
    return get_sino(e, &e->sino[id], out);
}

static int get_scratch(tomo_engine *e, float **slot, float **out)
{
    if (!*slot) {
        int rc = dev_alloc((void **)slot, e->vol_elems() * sizeof(float), true, e->stream);
        if (rc) return rc;
    }
    *out = *slot;
    return TOMO_OK;
}

// A write to a volume or to the re-projection G on the main stream must not overtake an evaluation still reading them
// on the second stream (tomo_data_distance_sq_async): order the main stream behind it.  No-op when nothing is pending.
static int order_after_async(tomo_engine *e)
{
    if (e->async_pending) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join, 0));
    return TOMO_OK;
}

static int ensure_stage(tomo_engine *e, size_t bytes)
{
    if (e->stage_bytes >= bytes) return TOMO_OK;
    if (e->stage) { HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipFree(e->stage)); e->stage = nullptr; }
    HIPCHK(hipMalloc((void **)&e->stage, bytes));
    e->stage_bytes = bytes;
    return TOMO_OK;
}

// ---- profiling brackets (bench.py roofline: HIP events on the launch stream) --------------------------
// Where a launch goes: the engine's stream and the whole slab, or one sub-slab (64-slice chunks [c0, c0 + nc)) on its own
// stream.  Passed explicitly so that two host threads can enqueue the two sub-slab chains of a SART sweep side by side.
struct Sub {
    hipStream_t stream; int c0 = 0, nc = 0;
    int vec = 0;       // vector width of the per-row kernels on this sub-slab (0: the engine's); divides c0 and nc
};
static int sub_vec(const tomo_engine *e, const Sub &sb) { return sb.nc && sb.vec ? sb.vec : e->vec; }
static Sub whole(const tomo_engine *e) { return Sub{e->stream, e->sub_c0, e->sub_nc}; }

struct ProfScope {
    tomo_engine *e; int k; hipEvent_t stop = nullptr; hipStream_t st;
    // key >= 0: the launch's position in its chain (the two sub-slab chains of a SART sweep then bracket the SAME links, so
    // that the overlap of sibling launches can be seen); key < 0: every stride-th launch in arrival order
    ProfScope(tomo_engine *e_, int k_, hipStream_t st_ = nullptr, int64_t key = -1) : e(e_), k(k_), st(st_ ? st_ : e_->stream)
    {
        ProfSlot &p = e->prof[k];
        if (!p.on) return;
        hipEvent_t start;
        {
            std::lock_guard<std::mutex> lk(e->prof_mu);       // two threads may log launches of one kernel
            if (p.stride > 1 && ((key >= 0 ? (uint64_t)key : (uint64_t)p.seen++) % p.stride) != 0) return;
            if (p.used + 2 > p.ev.size()) {
                if (p.ev.size() >= PROF_MAX_EVENTS) { ++p.dropped; return; }
                hipEvent_t a, b;
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { ++p.dropped; return; }
                p.ev.push_back(a); p.ev.push_back(b);
            }
            start = p.ev[p.used];
            stop = p.ev[p.used + 1];
            p.used += 2;
        }
        (void)hipEventRecord(start, st);
    }
    ~ProfScope() { if (stop) (void)hipEventRecord(stop, st); }
};

// ---- reductions ------------------------------------------------------------------------------------------
// The partial-sum buffers are zero between reductions: they are allocated zeroed and k_finalize clears what it has read,
// so a reduction costs no memset launch.  Only a reduction that was abandoned half-way (an error return between begin
// and end) leaves its buffer marked open, and the next begin clears it.
static bool &part_open(tomo_engine *e, const double *part)
{
    return part == e->d_part_tv ? e->part_open[2] : part == e->d_part_aux ? e->part_open[1] : e->part_open[0];
}
static int part_begin(tomo_engine *e, double *part)
{
    bool &open = part_open(e, part);
    if (open) HIPCHK(hipMemsetAsync(part, 0, NPART * sizeof(double), e->stream));
    open = true;
    return TOMO_OK;
}
static int part_end(tomo_engine *e, double *part, int slot)
{
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(NPART), 0, e->stream, part, e->d_scal + slot);
    LAUNCHCHK();
    part_open(e, part) = false;
    return TOMO_OK;
}
static int reduce_begin(tomo_engine *e) { return part_begin(e, e->d_part); }
static int reduce_end(tomo_engine *e, int slot) { return part_end(e, e->d_part, slot); }

constexpr int TV_YSEG_MIN = 8, TV_WAVES_WANTED = 4096;   // round 3: the march holds 4 waves per SIMD = 4096 resident waves: one full round
                                                         // (8192 before; 128 slices: 16 rows per wave 94.5 us per inner iteration against 8 rows ~100)
static int grid_1d(int64_t n4) { int64_t b = (n4 + 255) / 256; return (int)std::min<int64_t>(std::max<int64_t>(b, 1), 4096); }

// ---- projector launches -------------------------------------------------------------------------------------
// vec_override: 0 = wide form (64*vec slices per workgroup, scalar table walk); 16 / 32 = narrow-chunk form with that
// many lanes per ray (k_fp_rows_g)
template <int MODE>
static int launch_fp(tomo_engine *e, const float *x, int row0, int nrows, const float *b, float *out, int lpr = 0)
{
    if (lpr == 16 || lpr == 32) {
        int R = 64 / lpr;
        int nchunk = e->sxc / (lpr * 4);
        int64_t waves = (int64_t)((nrows + R - 1) / R) * nchunk;
        dim3 grid((unsigned)((waves + 3) / 4)), block(256);
        if (lpr == 16) hipLaunchKernelGGL((k_fp_rows_g<16, MODE>), grid, block, 0, e->stream, x, e->d_rptr, e->d_rent, b, e->d_rowsum, out, e->d_part, row0, nrows, e->sx, nchunk);
        else hipLaunchKernelGGL((k_fp_rows_g<32, MODE>), grid, block, 0, e->stream, x, e->d_rptr, e->d_rent, b, e->d_rowsum, out, e->d_part, row0, nrows, e->sx, nchunk);
        LAUNCHCHK();
        return TOMO_OK;
    }
    int vec = e->vec;
    int nchunk = e->sxc / (64 * vec);
    dim3 grid((unsigned)((int64_t)nrows * nchunk)), block(256);
    switch (vec) {
    case 4: hipLaunchKernelGGL((k_fp_rows<4, MODE>), grid, block, 0, e->stream, x, e->d_rptr, e->d_rent, b, e->d_rowsum, out, e->d_part, row0, nrows, e->sx); break;
    case 2: hipLaunchKernelGGL((k_fp_rows<2, MODE>), grid, block, 0, e->stream, x, e->d_rptr, e->d_rent, b, e->d_rowsum, out, e->d_part, row0, nrows, e->sx); break;
    default: hipLaunchKernelGGL((k_fp_rows<1, MODE>), grid, block, 0, e->stream, x, e->d_rptr, e->d_rent, b, e->d_rowsum, out, e->d_part, row0, nrows, e->sx); break;
    }
    LAUNCHCHK();
    return TOMO_OK;
}

template <int MODE>
static void launch_fp_reduce(tomo_engine *e, hipStream_t rs, const float *part, const uint32_t *rsptr, const uint32_t *rsidx, const float *b,
                             float *out, int c0, int ncp)
{
    int lpr = (ncp % 4 == 0) ? 64 : (ncp % 2 == 0) ? 32 : 16;
    int64_t items = (int64_t)e->nrows * (ncp * 16 / lpr);
    int64_t waves = (items + 64 / lpr - 1) / (64 / lpr);
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    switch (lpr) {
    case 64: hipLaunchKernelGGL((k_fp_tile_reduce<64, MODE>), grid, block, 0, rs, part, rsptr, rsidx, b, e->d_rowsum, out, e->d_part, (int)e->nrows, e->sx, c0, ncp); break;
    case 32: hipLaunchKernelGGL((k_fp_tile_reduce<32, MODE>), grid, block, 0, rs, part, rsptr, rsidx, b, e->d_rowsum, out, e->d_part, (int)e->nrows, e->sx, c0, ncp); break;
    default: hipLaunchKernelGGL((k_fp_tile_reduce<16, MODE>), grid, block, 0, rs, part, rsptr, rsidx, b, e->d_rowsum, out, e->d_part, (int)e->nrows, e->sx, c0, ncp); break;
    }
}

// all-angle FP, sheared-strip form (k_fp_strip + k_fp_tile_reduce on the strips' row lists)
template <int MODE>
static int launch_fp_strip(tomo_engine *e, const float *x, const float *b, float *out)
{
    const int nchunk = e->sxc / 64;
    if (!e->fs_ncp) {
        size_t per_chunk = (size_t)std::max<uint32_t>(1, e->fs_nseg) * 64 * sizeof(float);
        int ncp = (int)std::min<size_t>(nchunk, std::max<size_t>(1, e->ft_scratch_cap / per_chunk));
        if (e->ft_ncp_forced > 0) ncp = std::min(nchunk, e->ft_ncp_forced);
        else if (ncp >= 4) ncp &= ~3; else if (ncp >= 2) ncp &= ~1;
        e->fs_ncp = ncp;
    }
    const int which = (e->aux && e->stream == e->aux) ? 1 : 0;
    float **slot = which ? &e->fs_part_aux : &e->fs_part;
    if (!*slot) {
        int rc = dev_alloc((void **)slot, (size_t)std::max<uint32_t>(1, e->fs_nseg) * e->fs_ncp * 64 * sizeof(float), false, e->stream);
        if (rc) return rc;
    }
    for (int c0 = 0; c0 < nchunk; c0 += e->fs_ncp) {
        const int ncp = std::min(e->fs_ncp, nchunk - c0);
        {
            ProfScope ps(e, TOMO_K_FP_TILE);
            dim3 grid((unsigned)(8 * ((e->fs_nitems + 7) / 8) * ncp)), block(FS_THREADS);
#define FS_LAUNCH(KK) hipLaunchKernelGGL((k_fp_strip<KK>), grid, block, 0, e->stream, x, e->d_fs_items, e->d_fs_orient, e->d_fs_shift, e->d_fs_cnt, \
                                         e->d_fs_gstart, e->d_fs_gseg0, e->d_fs_ent, *slot, e->n, e->sx, e->fs_nitems, c0, ncp, e->d_fs_zero)
            if (e->fs_kused <= 8) FS_LAUNCH(8); else if (e->fs_kused <= 12) FS_LAUNCH(12); else FS_LAUNCH(16);
#undef FS_LAUNCH
            LAUNCHCHK();
        }
        {
            ProfScope ps(e, TOMO_K_FP_REDUCE);
            launch_fp_reduce<MODE>(e, e->stream, *slot, e->d_fs_rsptr, e->d_fs_rsidx, b, out, c0, ncp);
            LAUNCHCHK();
        }
    }
    return TOMO_OK;
}

// all-angle FP, sheared strips as wave-uniform entry lists (k_fp_list + k_fp_tile_reduce on the lists' row lists)
template <int MODE>
static int launch_fp_list(tomo_engine *e, const float *x, const float *b, float *out)
{
    const int nchunk = e->sxc / 64;
    if (!e->fl_ncp) {
        size_t per_chunk = (size_t)std::max<uint32_t>(1, e->fl_nseg) * 64 * sizeof(float);
        int ncp = (int)std::min<size_t>(nchunk, std::max<size_t>(2, e->ft_scratch_cap / per_chunk));
        if (e->ft_ncp_forced > 0) ncp = std::min(nchunk, std::max(2, e->ft_ncp_forced));
        if (ncp >= 4) ncp &= ~3; else ncp = 2;                       // whole 128-slice pieces
        e->fl_ncp = ncp;
    }
    if (!e->attr_fl) {
        HIPCHK(hipFuncSetAttribute((const void *)k_fp_list, hipFuncAttributeMaxDynamicSharedMemorySize, FL_LDS_BYTES));
        e->attr_fl = true;
    }
    const int which = (e->aux && e->stream == e->aux) ? 1 : 0;
    float **slot = which ? &e->fl_part_aux : &e->fl_part;
    if (!*slot) {
        int rc = dev_alloc((void **)slot, (size_t)std::max<uint32_t>(1, e->fl_nseg) * e->fl_ncp * 64 * sizeof(float), false, e->stream);
        if (rc) return rc;
    }
    for (int c0 = 0; c0 < nchunk; c0 += e->fl_ncp) {
        const int ncp = std::min(e->fl_ncp, nchunk - c0);            // even: the slab is whole 128-slice pieces
        {
            ProfScope ps(e, TOMO_K_FP_TILE);
            dim3 grid((unsigned)(8 * ((e->fl_nitems + 7) / 8) * (ncp / 2))), block(FL_THREADS);
            hipLaunchKernelGGL(k_fp_list, grid, block, FL_LDS_BYTES, e->stream, x, e->d_fl_items, e->d_fl_orient, e->d_fl_shift, e->d_fl_ent, e->d_fl_ptr,
                               e->d_fl_fent, e->d_fl_fptr, *slot, e->n, e->sx, e->fl_nitems, c0 / 2, ncp / 2, ncp, e->d_fl_zero);
            LAUNCHCHK();
        }
        {
            ProfScope ps(e, TOMO_K_FP_REDUCE);
            launch_fp_reduce<MODE>(e, e->stream, *slot, e->d_fl_rsptr, e->d_fl_rsidx, b, out, c0, ncp);
            LAUNCHCHK();
        }
    }
    return TOMO_OK;
}

// ---- which form runs ---------------------------------------------------------------------------------------------------------
// ONE place decides which kernel family an operation of this engine uses, from what finish_create_impl could build for the geometry
// (the *_ok flags), the slab's shape and the options in force; the launchers below ask it, and so can a host
// (tomo_get_option "form_fp" / "form_bp" / "form_sart": the TOMO_FORM_* codes of include/tomo_hip.h).
//   all-angle forward projection   LIST   the sheared strips as wave-uniform entry lists (k_fp_list): slab = whole 128-slice pieces,
//                                         tables built (N >= 384 and >= 3000 strip workgroups, or TOMO_FP_LIST = 1), balance >= 0.8
//                                  STRIP  k_fp_strip: the same geometry rule where the slab is no multiple of 128 slices or the lists
//                                         could not be balanced
//                                  TILE   k_fp_tile + k_fp_tile_reduce: small images and thin slabs (the default there), user matrices
//                                         whose rays are no lines
//                                  ROWS   ray-driven k_fp_rows / k_fp_rows_g: "fp_tile" = 0 only
//   all-angle back projection      LIST   k_bp_list: whole 128-slice pieces, P <= 192;  TILE  k_bp_tile: other slabs, P <= FB_MAX_PROJ;
//                                  ALL    voxel-driven k_bp_all: the fallback
//   SART sweep                     RESIDENT  k_sart_resident: N a multiple of 8, one 32 x 32 tile per CU, ray windows fit (resident.cpp)
//                                  TILE      k_sart_tile chain + k_resid_finish: N = 1024, fallbacks, "sart_resident" = 0
//                                  ANGLE     one FP + one BP launch per angle: "sart_fused" = 0 or no tile tables
struct Forms { int fp, bp, sart; };
static Forms select_forms(const tomo_engine *e)
{
    Forms f;
    f.fp = (e->fp_strip && e->fp_list && e->fl_ok && e->sxc % 128 == 0) ? TOMO_FORM_FP_LIST
         : (e->fp_strip && e->fs_ok) ? TOMO_FORM_FP_STRIP
         : e->fp_tile ? TOMO_FORM_FP_TILE : TOMO_FORM_FP_ROWS;
    f.bp = (e->bp_tile && e->bl_ok && e->bp_list && e->sxc % 128 == 0) ? TOMO_FORM_BP_LIST
         : (e->bp_tile && e->fb_ok) ? TOMO_FORM_BP_TILE : TOMO_FORM_BP_ALL;
    f.sart = !e->sart_fused ? TOMO_FORM_SART_ANGLE
           : (e->sart_resident != 0 && e->rs_ok) ? TOMO_FORM_SART_RESIDENT
           : (e->sart_tile && e->st_ok) ? TOMO_FORM_SART_TILE : TOMO_FORM_SART_ANGLE;
    return f;
}

// all-angle FP: sheared-strip form, else the tile-stationary form (k_fp_tile + k_fp_tile_reduce) unless switched off, else the ray-driven form
template <int MODE>
static int launch_fp_all(tomo_engine *e, const float *x, const float *b, float *out)
{
    const int form = select_forms(e).fp;
    if (form == TOMO_FORM_FP_LIST) return launch_fp_list<MODE>(e, x, b, out);
    if (form == TOMO_FORM_FP_STRIP) return launch_fp_strip<MODE>(e, x, b, out);
    if (form == TOMO_FORM_FP_ROWS) return launch_fp<MODE>(e, x, 0, (int)e->nrows, b, out, e->fp_all_lpr);
    const int nchunk = e->sxc / 64;
    if (!e->ft_ncp) {
        size_t per_chunk = (size_t)std::max<uint32_t>(1, e->ft_nseg) * 64 * sizeof(float);
        int ncp = (int)std::min<size_t>(nchunk, std::max<size_t>(1, e->ft_scratch_cap / per_chunk));
        if (e->ft_ncp_forced > 0) ncp = std::min(nchunk, e->ft_ncp_forced);
        else if (ncp >= 4) ncp &= ~3; else if (ncp >= 2) ncp &= ~1;
        e->ft_ncp = ncp;
        if (!e->attr_fp) {   // per engine: the attribute belongs to the (function, device) pair
            HIPCHK(hipFuncSetAttribute((const void *)k_fp_tile, hipFuncAttributeMaxDynamicSharedMemorySize, FT_LDS_BYTES));
            e->attr_fp = true;
        }
    }
    const int which = (e->aux && e->stream == e->aux) ? 1 : 0;
    float **slot = which ? &e->ft_part_aux : &e->ft_part;
    // The tile kernel is LDS / vector-ALU bound and WRITES the partial sums; the reduce kernel is HBM-read bound.  With
    // "fp_tile_pipe" = P >= 2 the projection runs as P groups of chunks, the reduce of group k on a helper stream beside the tile
    // kernel of group k+1 (two halves of the scratch, events both ways): the two kernels want different parts of the chip.
    // Built and measured in round 3 -- and it buys nothing (numbers at the option's declaration): the tile kernel's one 1024-thread
    // workgroup per CU leaves room for one reduce wave per SIMD, which reads no faster than the tile kernel's own stores leave
    // the memory system idle.  Kept as an option, default off.
    int ncp_call = e->ft_ncp, pipe = 0;
    if (e->fp_tile_pipe >= 2 && nchunk >= 2 * 2) {          // groups of at least two chunks (64-lane reduce spans)
        int P = std::min(e->fp_tile_pipe, nchunk / 2);
        int per = (nchunk + P - 1) / P;
        per = (per + 1) & ~1;
        if (per <= e->ft_ncp && per < nchunk) { ncp_call = per; pipe = 1; }
    }
    if (!*slot) {
        // sized for both schemes: one pass of ft_ncp chunks, or two halves of a pipelined group each
        size_t chunks = std::max<size_t>(e->ft_ncp, 2 * (size_t)((((e->sxc / 64 + 1) / 2) + 1) & ~1));
        int rc = dev_alloc((void **)slot, (size_t)std::max<uint32_t>(1, e->ft_nseg) * chunks * 64 * sizeof(float), false, e->stream);
        if (rc) return rc;
    }
    if (pipe && !e->fp_red_stream[which]) {
        HIPCHK(hipStreamCreateWithFlags(&e->fp_red_stream[which], hipStreamNonBlocking));
        for (int h = 0; h < 2; ++h) {
            HIPCHK(hipEventCreateWithFlags(&e->ev_fp_tile[which][h], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&e->ev_fp_red[which][h], hipEventDisableTiming));
        }
    }
    const size_t half_elems = (size_t)std::max<uint32_t>(1, e->ft_nseg) * ncp_call * 64;
    int k = 0;
    for (int c0 = 0; c0 < nchunk; c0 += ncp_call, ++k) {
        int ncp = std::min(ncp_call, nchunk - c0);
        const int half = k & 1;
        float *part = *slot + (pipe ? half * half_elems : 0);
        hipStream_t rs = pipe ? e->fp_red_stream[which] : e->stream;
        if (pipe && k >= 2) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_fp_red[which][half], 0));   // this half of the scratch is free again
        {
            ProfScope ps(e, TOMO_K_FP_TILE);
            dim3 grid((unsigned)(8 * ((e->ft_ntiles + 7) / 8) * ncp)), block(FT_THREADS);
            hipLaunchKernelGGL(k_fp_tile, grid, block, FT_LDS_BYTES, e->stream, x, e->d_ft_slot_ptr, e->d_ft_slot_seg0, e->d_ft_tent, part,
                               e->n, e->sx, e->ft_tiles_z, e->ft_ntiles, c0, ncp);
            LAUNCHCHK();
        }
        if (pipe) {
            HIPCHK(hipEventRecord(e->ev_fp_tile[which][half], e->stream));
            HIPCHK(hipStreamWaitEvent(rs, e->ev_fp_tile[which][half], 0));
        }
        {
            ProfScope ps(e, TOMO_K_FP_REDUCE, rs);
            launch_fp_reduce<MODE>(e, rs, part, e->d_ft_rsptr, e->d_ft_rsidx, b, out, c0, ncp);
            LAUNCHCHK();
        }
        if (pipe) HIPCHK(hipEventRecord(e->ev_fp_red[which][half], rs));
    }
    if (pipe) {                                             // the projection is complete on the caller's stream
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_fp_red[which][0], 0));
        if (k >= 2) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_fp_red[which][1], 0));
    }
    return TOMO_OK;
}

// residual rows from a projection already in G (fp_reuse)
template <int MODE>
static int launch_sino_resid(tomo_engine *e, const float *b, const float *g, float *out)
{
    const int64_t n4 = (int64_t)e->sino_elems() / 4;
    hipLaunchKernelGGL((k_sino_resid<MODE>), dim3(grid_1d(n4)), dim3(256), 0, e->stream, b, g, e->d_rowsum, out, n4, e->sx / 4);
    LAUNCHCHK();
    return TOMO_OK;
}

constexpr int BP_PPW = 4;

static int launch_bp_angle(tomo_engine *e, const Sub &sb, float *x, int angle, const float *r_angle, float beta, float *track = nullptr)
{
    ProfScope ps(e, TOMO_K_BP_ANGLE, sb.stream);
    const int vec = sub_vec(e, sb);
    int nchunk = e->sxc / (64 * vec), chunk0 = 0;
    if (sb.nc) { nchunk = sb.nc / vec; chunk0 = sb.c0 / vec; }   // sub-slab (whole multiples of 64*vec slices)
    int ngroups = (int)((e->npix + BP_PPW - 1) / BP_PPW);
    int64_t waves = (int64_t)ngroups * nchunk;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    const CellD *cell = e->d_cell + (size_t)angle * e->npix;
    if (track) {   // caller brackets with reduce_begin / reduce_end
        switch (vec) {
        case 4: hipLaunchKernelGGL((k_bp_angle<4, BP_PPW, true>), grid, block, 0, sb.stream, x, cell, r_angle, beta, (int)e->npix, e->sx, ngroups, nchunk, track, e->d_part, chunk0); break;
        case 2: hipLaunchKernelGGL((k_bp_angle<2, BP_PPW, true>), grid, block, 0, sb.stream, x, cell, r_angle, beta, (int)e->npix, e->sx, ngroups, nchunk, track, e->d_part, chunk0); break;
        default: hipLaunchKernelGGL((k_bp_angle<1, BP_PPW, true>), grid, block, 0, sb.stream, x, cell, r_angle, beta, (int)e->npix, e->sx, ngroups, nchunk, track, e->d_part, chunk0); break;
        }
        LAUNCHCHK();
        return TOMO_OK;
    }
    switch (vec) {
    case 4: hipLaunchKernelGGL((k_bp_angle<4, BP_PPW, false>), grid, block, 0, sb.stream, x, cell, r_angle, beta, (int)e->npix, e->sx, ngroups, nchunk, (float *)nullptr, (double *)nullptr, chunk0); break;
    case 2: hipLaunchKernelGGL((k_bp_angle<2, BP_PPW, false>), grid, block, 0, sb.stream, x, cell, r_angle, beta, (int)e->npix, e->sx, ngroups, nchunk, (float *)nullptr, (double *)nullptr, chunk0); break;
    default: hipLaunchKernelGGL((k_bp_angle<1, BP_PPW, false>), grid, block, 0, sb.stream, x, cell, r_angle, beta, (int)e->npix, e->sx, ngroups, nchunk, (float *)nullptr, (double *)nullptr, chunk0); break;
    }
    LAUNCHCHK();
    return TOMO_OK;
}

static int launch_bp_angle(tomo_engine *e, float *x, int angle, const float *r_angle, float beta, float *track = nullptr)
{
    return launch_bp_angle(e, whole(e), x, angle, r_angle, beta, track);
}

// segmented per-angle step: FUSED -> BP(prev) + FP(next); else plain FP(next).  Leaves the residual rows of `next` in r.
template <bool FUSED>
static int launch_sart_seg(tomo_engine *e, const float *x_old, float *x_new, int prev, int next, float *r, float beta)
{
    int rc;
    if (!e->seg_partial) {
        if ((rc = dev_alloc((void **)&e->seg_partial, (size_t)std::max<uint32_t>(1, e->max_items) * e->sx * sizeof(float), true, e->stream))) return rc;
    }
    int nchunk = e->sxc / (64 * e->vec);
    uint32_t b0 = e->h_seg_exec_ptr[next], b1 = e->h_seg_exec_ptr[next + 1];
    int L = (int)((b1 - b0) / 8);
    const SegItemD *exec = e->d_seg_exec + b0;
    const CellD *cell = FUSED ? e->d_cell + (size_t)prev * e->npix : nullptr;
    const float *rp = FUSED ? r + (size_t)prev * e->n * e->sx : nullptr;
    if (L > 0) {
        ProfScope ps(e, FUSED ? TOMO_K_SART_FUSED : TOMO_K_FP_ANGLE);
        dim3 grid((unsigned)(8 * (int64_t)L * nchunk)), block(64);
        switch (e->vec) {
        case 4: hipLaunchKernelGGL((k_sart_seg<4, 8, FUSED>), grid, block, 0, e->stream, x_old, x_new, exec, L, e->d_went, cell, rp, beta, e->seg_partial, e->sx); break;
        case 2: hipLaunchKernelGGL((k_sart_seg<2, 8, FUSED>), grid, block, 0, e->stream, x_old, x_new, exec, L, e->d_went, cell, rp, beta, e->seg_partial, e->sx); break;
        default: hipLaunchKernelGGL((k_sart_seg<1, 8, FUSED>), grid, block, 0, e->stream, x_old, x_new, exec, L, e->d_went, cell, rp, beta, e->seg_partial, e->sx); break;
        }
        LAUNCHCHK();
    }
    {
        dim3 grid((unsigned)((int64_t)e->n * nchunk)), block(256);
        switch (e->vec) {
        case 4: hipLaunchKernelGGL((k_resid_finish<4>), grid, block, 0, e->stream, e->seg_partial, e->d_row_first, e->d_row_nseg, e->cur_b, e->d_rowsum, r, next * e->n, e->n, nchunk, e->sx, 0); break;
        case 2: hipLaunchKernelGGL((k_resid_finish<2>), grid, block, 0, e->stream, e->seg_partial, e->d_row_first, e->d_row_nseg, e->cur_b, e->d_rowsum, r, next * e->n, e->n, nchunk, e->sx, 0); break;
        default: hipLaunchKernelGGL((k_resid_finish<1>), grid, block, 0, e->stream, e->seg_partial, e->d_row_first, e->d_row_nseg, e->cur_b, e->d_rowsum, r, next * e->n, e->n, nchunk, e->sx, 0); break;
        }
        LAUNCHCHK();
    }
    return TOMO_OK;
}

// tile form of the per-angle step (k_sart_tile): FUSED -> BP(prev) + FP(next), in place; else plain FP(next).
// Leaves the residual rows of `next` in r.
// (function attributes and the partial-sum buffer are set up by sart_tile_prepare, on the caller's thread and stream)
static int sart_tile_prepare(tomo_engine *e, bool coop)
{
    if (!e->attr_st) {
        const void *forms[] = {(const void *)k_sart_tile<true, false, true>, (const void *)k_sart_tile<true, false, false>,
                               (const void *)k_sart_tile<false, false, true>, (const void *)k_sart_tile<false, false, false>,
                               (const void *)k_sart_tile<true, true, true>, (const void *)k_sart_tile<true, true, false>,
                               (const void *)k_sart_tile<true, false, true, true>, (const void *)k_sart_tile<true, false, false, true>};
        for (const void *f : forms) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS_V * 16));
        e->attr_st = true;
    }
    const size_t pbytes = (size_t)std::max<uint32_t>(1, e->st_max_ids) * e->sx * sizeof(float);
    if (!e->st_partial) {
        int rc = dev_alloc((void **)&e->st_partial, pbytes, true, e->stream);
        if (rc) return rc;
    }
    if (coop && !e->st_partial2) {
        int rc = dev_alloc((void **)&e->st_partial2, pbytes, true, e->stream);
        if (rc) return rc;
        if ((rc = dev_alloc((void **)&e->st_flags, (size_t)e->n * (e->sxc / 64) * sizeof(uint32_t), true, e->stream))) return rc;
        // workgroups that start together: the reducer duty is dealt to that many (2 per CU by LDS; any value is correct)
        int per_cu = 0;
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, e->device));
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_sart_tile<true, true, true>, ST_THREADS, ST_LDS_V * 16));
        e->st_resident = std::max(1, per_cu) * std::max(1, prop.multiProcessorCount);
    }
    return TOMO_OK;
}

// A slab larger than the Infinity Cache is streamed (non-temporal tile accesses); a smaller one stays cached between the
// launches of consecutive angles and keeps plain accesses (see st_xload / st_xstore).  "sart_nt": 0 never, 1 always, -1 by size.
static bool slab_streams(const tomo_engine *e)
{
    if (e->sart_nt >= 0) return e->sart_nt != 0;
    return (size_t)e->npix * e->sx * sizeof(float) >= ((size_t)192 << 20);
}

// residual rows of angle `next` from the tile partial sums in `partial` (k_resid_finish)
static int launch_resid_finish_tile(tomo_engine *e, const Sub &sb, const float *partial, int next, float *r, bool sum = false)
{
    const int nchunk64 = sb.nc ? sb.nc : e->sxc / 64, c64 = sb.nc ? sb.c0 : 0;
    const int vec = sub_vec(e, sb);
    int nchunk = nchunk64 / vec, chunk0 = c64 / vec;
    dim3 grid((unsigned)((int64_t)e->n * nchunk)), block(256);
    if (sum) {      // plain row sums (chained ART: k_art_chain forms the residuals)
        switch (vec) {
        case 4: hipLaunchKernelGGL((k_resid_finish<4, true>), grid, block, 0, sb.stream, partial, e->d_st_row_first, e->d_st_row_nseg, e->cur_b, e->d_rowsum, r, next * e->n, e->n, nchunk, e->sx, chunk0); break;
        case 2: hipLaunchKernelGGL((k_resid_finish<2, true>), grid, block, 0, sb.stream, partial, e->d_st_row_first, e->d_st_row_nseg, e->cur_b, e->d_rowsum, r, next * e->n, e->n, nchunk, e->sx, chunk0); break;
        default: hipLaunchKernelGGL((k_resid_finish<1, true>), grid, block, 0, sb.stream, partial, e->d_st_row_first, e->d_st_row_nseg, e->cur_b, e->d_rowsum, r, next * e->n, e->n, nchunk, e->sx, chunk0); break;
        }
        LAUNCHCHK();
        return TOMO_OK;
    }
    switch (vec) {
    case 4: hipLaunchKernelGGL((k_resid_finish<4>), grid, block, 0, sb.stream, partial, e->d_st_row_first, e->d_st_row_nseg, e->cur_b, e->d_rowsum, r, next * e->n, e->n, nchunk, e->sx, chunk0); break;
    case 2: hipLaunchKernelGGL((k_resid_finish<2>), grid, block, 0, sb.stream, partial, e->d_st_row_first, e->d_st_row_nseg, e->cur_b, e->d_rowsum, r, next * e->n, e->n, nchunk, e->sx, chunk0); break;
    default: hipLaunchKernelGGL((k_resid_finish<1>), grid, block, 0, sb.stream, partial, e->d_st_row_first, e->d_st_row_nseg, e->cur_b, e->d_rowsum, r, next * e->n, e->n, nchunk, e->sx, chunk0); break;
    }
    LAUNCHCHK();
    return TOMO_OK;
}

// finish = false leaves the partial sums of `next` in `partial` for the next link's reducer duty (cooperative chain)
// ART: the fused step of the chained ART sweep (r = the rows k_art_chain left for `prev`; the finish stores plain row sums into
// fp_out, from which k_art_chain forms the rows of `next`)
template <bool FUSED, bool ART = false>
static int launch_sart_tile(tomo_engine *e, const Sub &sb, float *x, int prev, int next, float *r, float beta,
                            float *partial = nullptr, bool finish = true, int64_t key = -1, float *fp_out = nullptr)
{
    const int nchunk64 = sb.nc ? sb.nc : e->sxc / 64, c64 = sb.nc ? sb.c0 : 0;
    const size_t nt = (size_t)e->st_ntiles;
    if (!partial) partial = e->st_partial;
    {
        ProfScope ps(e, FUSED ? TOMO_K_SART_FUSED : TOMO_K_FP_ANGLE, sb.stream, key);
        dim3 grid((unsigned)(8 * ((e->st_ntiles + 7) / 8) * nchunk64)), block(ST_THREADS);
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, grid, block, ST_LDS_V * 16, sb.stream, x, x,
                               FUSED ? e->d_st_cell + (size_t)prev * nt * ST_PIX : nullptr, FUSED ? e->d_st_win + (size_t)prev * nt : nullptr,
                               FUSED ? r + (size_t)prev * e->n * e->sx : nullptr, beta,
                               e->d_st_seg + (size_t)next * nt * ST_MAXSEG, e->d_st_segid + (size_t)next * nt * ST_MAXSEG, e->d_st_ent, partial,
                               e->n, e->sx, e->st_tiles_z, e->st_ntiles, nchunk64, c64, e->sart_skip_same, StCoop{});
        };
        if (slab_streams(e)) go(k_sart_tile<FUSED, false, true, ART && FUSED>); else go(k_sart_tile<FUSED, false, false, ART && FUSED>);
        LAUNCHCHK();
    }
    if (ART) return launch_resid_finish_tile(e, sb, partial, next, fp_out, true);
    return finish ? launch_resid_finish_tile(e, sb, partial, next, r) : TOMO_OK;
}

// cooperative link: residual rows of `prev` from p_read (reducer duty of the first workgroups), BP(prev) + FP(next) -> p_write
static int launch_sart_coop(tomo_engine *e, const Sub &sb, float *x, int prev, int next, float *r, float beta,
                            const float *p_read, float *p_write, uint32_t epoch, int64_t key = -1)
{
    const int nchunk64 = sb.nc ? sb.nc : e->sxc / 64, c64 = sb.nc ? sb.c0 : 0;
    const size_t nt = (size_t)e->st_ntiles;
    ProfScope ps(e, TOMO_K_SART_FUSED, sb.stream, key);
    const unsigned nblocks = (unsigned)(8 * ((e->st_ntiles + 7) / 8) * nchunk64);
    StCoop co;
    co.p_read = p_read;
    co.row_first = e->d_st_row_first + (size_t)prev * e->n;
    co.row_nseg = e->d_st_row_nseg + (size_t)prev * e->n;
    co.b = e->cur_b + (size_t)prev * e->n * e->sx;
    co.rowsum = e->d_rowsum + (size_t)prev * e->n;
    co.r_out = r + (size_t)prev * e->n * e->sx;
    co.flags = e->st_flags;
    co.epoch = epoch;
    co.nitems = e->n * nchunk64;
    co.nred = (int)std::min<unsigned>(nblocks, (unsigned)std::max(1, e->st_resident));
    co.nchunk_all = e->sxc / 64;
    co.spin = e->sart_coop_spin;
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(ST_THREADS), ST_LDS_V * 16, sb.stream, x, x,
                           e->d_st_cell + (size_t)prev * nt * ST_PIX, e->d_st_win + (size_t)prev * nt, r + (size_t)prev * e->n * e->sx, beta,
                           e->d_st_seg + (size_t)next * nt * ST_MAXSEG, e->d_st_segid + (size_t)next * nt * ST_MAXSEG, e->d_st_ent, p_write,
                           e->n, e->sx, e->st_tiles_z, e->st_ntiles, nchunk64, c64, e->sart_skip_same, co);
    };
    if (slab_streams(e)) go(k_sart_tile<true, true, true>); else go(k_sart_tile<true, true, false>);
    LAUNCHCHK();
    return TOMO_OK;
}

static int launch_bp_all(tomo_engine *e, float *x, const float *r, const float *colsum, float alpha, float beta, int clamp)
{
    const int form = select_forms(e).bp;
    if (form == TOMO_FORM_BP_LIST) {     // entry lists: whole pairs of 64-slice chunks
        if (!e->attr_bp2) {
            HIPCHK(hipFuncSetAttribute((const void *)k_bp_list, hipFuncAttributeMaxDynamicSharedMemorySize, BL_LDS_BYTES));
            e->attr_bp2 = true;
        }
        const int nchunk2 = e->sxc / 128;
        ProfScope ps(e, TOMO_K_BP_TILE);
        dim3 grid((unsigned)(8 * ((e->bl_ntiles + 7) / 8) * nchunk2)), block(BL_THREADS);
        hipLaunchKernelGGL(k_bp_list, grid, block, BL_LDS_BYTES, e->stream, x, e->d_bl_ent, e->d_bl_ptr, e->d_bl_win, r, colsum, alpha, beta, clamp,
                           e->np, e->n, e->sx, e->bl_tiles_z, e->bl_ntiles, nchunk2, e->bp_list_band);
        LAUNCHCHK();
        return TOMO_OK;
    }
    if (form == TOMO_FORM_BP_TILE) {
        if (!e->attr_bp) {
            HIPCHK(hipFuncSetAttribute((const void *)k_bp_tile, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS_BYTES + FB_MAX_PROJ * 4));
            e->attr_bp = true;
        }
        const int nchunk64 = e->sxc / 64;
        ProfScope ps(e, TOMO_K_BP_TILE);
        dim3 grid((unsigned)(8 * ((e->ft_ntiles + 7) / 8) * nchunk64)), block(FT_THREADS);
        hipLaunchKernelGGL(k_bp_tile, grid, block, FB_LDS_BYTES + e->np * 4, e->stream, x, e->d_fb_cell, e->d_fb_win, r, colsum, alpha, beta, clamp,
                           e->np, e->n, e->sx, e->ft_tiles_z, e->ft_ntiles, nchunk64);
        LAUNCHCHK();
        return TOMO_OK;
    }
    int nchunk = e->sxc / (64 * e->vec);
    int ngroups = (int)((e->npix + BP_PPW - 1) / BP_PPW);
    int64_t waves = (int64_t)ngroups * nchunk;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    switch (e->vec) {
    case 4: hipLaunchKernelGGL((k_bp_all<4, BP_PPW>), grid, block, 0, e->stream, x, e->d_cell, r, colsum, alpha, beta, clamp, e->np, e->n, (int)e->npix, e->sx, ngroups, nchunk); break;
    case 2: hipLaunchKernelGGL((k_bp_all<2, BP_PPW>), grid, block, 0, e->stream, x, e->d_cell, r, colsum, alpha, beta, clamp, e->np, e->n, (int)e->npix, e->sx, ngroups, nchunk); break;
    default: hipLaunchKernelGGL((k_bp_all<1, BP_PPW>), grid, block, 0, e->stream, x, e->d_cell, r, colsum, alpha, beta, clamp, e->np, e->n, (int)e->npix, e->sx, ngroups, nchunk); break;
    }
    LAUNCHCHK();
    return TOMO_OK;
}

// ---- creation ----------------------------------------------------------------------------------------------
// host tables are released as soon as they are on the device (peak host memory: a few times nnz * 8 bytes per engine,
// and one process per GPU builds its own)
template <class V> static void release(V &v) { V().swap(v); }

static int finish_create_impl(tomo_engine *e, Coo &m, tomo_engine **out)
{
    Tables t;
    std::string err;
    const bool timing = std::getenv("TOMO_BUILD_TIMING") != nullptr;      // per-phase wall clock of the table builders to stderr
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = now();
    auto lap = [&](const char *what) { if (timing) { double x = now(); std::fprintf(stderr, "tomo_create: %-32s %.3f s\n", what, x - t_last); t_last = x; } };
    struct Meter { Meter(size_t *p) { g_alloc_meter = p; } ~Meter() { g_alloc_meter = nullptr; } } meter(&e->table_bytes);
    sort_rows(m);
    lap("sort_rows");
    if (!build_tables(m, e->n, e->np, t, err)) return fail(TOMO_ERR_GEOMETRY, err);
    lap("build_tables");
    e->nnz = m.ptr[m.nrow];
    if (e->nnz >= (int64_t)0xFFFFFFFFu) return fail(TOMO_ERR_ARG, "matrix too large for 32-bit entry offsets");
    e->lipschitz = t.lipschitz;
    e->lipschitz_cimmino = t.lipschitz_cimmino;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->own_stream = true;

    std::vector<uint32_t> ptr32(m.nrow + 1);
    for (int64_t r = 0; r <= m.nrow; ++r) ptr32[r] = (uint32_t)m.ptr[r];
    std::vector<uint2> ent(e->nnz ? e->nnz : 1);
    for (int64_t k = 0; k < e->nnz; ++k) { uint32_t bits; std::memcpy(&bits, &m.val[k], 4); ent[k] = make_uint2(m.col[k], bits); }
    int rc;
    if ((rc = dev_alloc((void **)&e->d_rptr, ptr32.size() * 4, false, e->stream))) return rc;
    if ((rc = dev_alloc((void **)&e->d_rent, ent.size() * sizeof(uint2), false, e->stream))) return rc;
    if ((rc = dev_alloc((void **)&e->d_rowsum, t.rowsum.size() * 4, false, e->stream))) return rc;
    if ((rc = dev_alloc((void **)&e->d_rowinner, t.rowinner.size() * 4, false, e->stream))) return rc;
    if ((rc = dev_alloc((void **)&e->d_colsum_all, t.colsum_all.size() * 4, false, e->stream))) return rc;
    if ((rc = dev_alloc((void **)&e->d_rowcross, t.rowcross.size() * 4, false, e->stream))) return rc;
    e->art_chain_ok = t.art_chain_ok;
    if ((rc = dev_alloc((void **)&e->d_cell, t.cell.size() * sizeof(CellD), false, e->stream))) return rc;
    HIPCHK(hipMemcpy(e->d_rptr, ptr32.data(), ptr32.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_rent, ent.data(), (size_t)e->nnz * sizeof(uint2), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_rowsum, t.rowsum.data(), t.rowsum.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_rowinner, t.rowinner.data(), t.rowinner.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_colsum_all, t.colsum_all.data(), t.colsum_all.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_rowcross, t.rowcross.data(), t.rowcross.size() * 4, hipMemcpyHostToDevice));
    lap("csr / cells upload");
    build_walk(m, e->n, e->np, t);
    {
        std::vector<uint2> went(t.walk_pix.size() ? t.walk_pix.size() : 1);
        for (size_t k = 0; k < t.walk_pix.size(); ++k) { uint32_t bits; std::memcpy(&bits, &t.walk_w[k], 4); went[k] = make_uint2(t.walk_pix[k], bits); }
        if ((rc = dev_alloc((void **)&e->d_wptr, t.walk_ptr.size() * 4, false, e->stream))) return rc;
        if ((rc = dev_alloc((void **)&e->d_went, went.size() * sizeof(uint2), false, e->stream))) return rc;
        HIPCHK(hipMemcpy(e->d_wptr, t.walk_ptr.data(), t.walk_ptr.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_went, went.data(), t.walk_pix.size() * sizeof(uint2), hipMemcpyHostToDevice));
    }
    lap("build_walk + upload");
    {
        const int seg_len = 32;   // visits per work item: 16/32/64/128 measured 222/224/233/242 us per fused step at 512^3
        build_segments(e->n, e->np, seg_len, t);
        static_assert(sizeof(Tables::SegItem) == sizeof(SegItemD), "segment item layout");
        e->h_seg_exec_ptr = t.seg_exec_ptr;
        e->max_items = t.max_items_per_angle;
        if ((rc = dev_alloc((void **)&e->d_seg_exec, std::max<size_t>(1, t.seg_exec.size()) * sizeof(SegItemD), false, e->stream))) return rc;
        if ((rc = dev_alloc((void **)&e->d_row_first, t.row_first.size() * 4, false, e->stream))) return rc;
        if ((rc = dev_alloc((void **)&e->d_row_nseg, t.row_nseg.size() * 4, false, e->stream))) return rc;
        HIPCHK(hipMemcpy(e->d_seg_exec, t.seg_exec.data(), t.seg_exec.size() * sizeof(SegItemD), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_row_first, t.row_first.data(), t.row_first.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_row_nseg, t.row_nseg.data(), t.row_nseg.size() * 4, hipMemcpyHostToDevice));
        release(t.walk_pix); release(t.walk_w); release(t.walk_ptr); release(t.seg_exec); release(t.row_first); release(t.row_nseg);
    }
    lap("build_segments + upload");
    {
        build_tiles(m, e->n, e->np, FT_TY, FT_TZ, 256, t);
        static_assert(Tables::TILE_SLOTS == FT_SLOTS && Tables::TILE_BATCH == FT_BATCH, "tile stream shape");
        e->ft_tiles_z = t.tiles_z; e->ft_ntiles = t.tiles_y * t.tiles_z;
        e->ft_nseg = t.tile_nseg;
        std::vector<uint2> tent(t.tile_off.size());
        for (size_t k = 0; k < tent.size(); ++k) { uint32_t bits; std::memcpy(&bits, &t.tile_w[k], 4); tent[k] = make_uint2(t.tile_off[k], bits); }
        if ((rc = dev_alloc((void **)&e->d_ft_slot_ptr, t.tile_slot_ptr.size() * 4, false, e->stream))) return rc;
        if ((rc = dev_alloc((void **)&e->d_ft_slot_seg0, std::max<size_t>(1, t.tile_slot_seg0.size()) * 4, false, e->stream))) return rc;
        if ((rc = dev_alloc((void **)&e->d_ft_tent, tent.size() * sizeof(uint2), false, e->stream))) return rc;
        if ((rc = dev_alloc((void **)&e->d_ft_rsptr, t.rseg_ptr.size() * 4, false, e->stream))) return rc;
        if ((rc = dev_alloc((void **)&e->d_ft_rsidx, t.rseg_idx.size() * 4, false, e->stream))) return rc;
        HIPCHK(hipMemcpy(e->d_ft_slot_ptr, t.tile_slot_ptr.data(), t.tile_slot_ptr.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_ft_slot_seg0, t.tile_slot_seg0.data(), t.tile_slot_seg0.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_ft_tent, tent.data(), tent.size() * sizeof(uint2), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_ft_rsptr, t.rseg_ptr.data(), t.rseg_ptr.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_ft_rsidx, t.rseg_idx.data(), t.rseg_idx.size() * 4, hipMemcpyHostToDevice));
        release(tent); release(t.tile_off); release(t.tile_w); release(t.rseg_idx); release(t.rseg_ptr); release(t.tile_slot_ptr); release(t.tile_slot_seg0);
        lap("build_tiles + upload");
        {   // sheared-strip tables of the all-angle FP; a geometry they cannot hold (a user matrix whose rays are no lines) keeps the tile form
            std::string why;
            static_assert(Tables::FS_W == FS_W && Tables::FS_H == FS_H && Tables::FS_WAVES == FS_WAVES && Tables::FS_GROUPS == FS_GROUPS, "strip shape");
            static_assert(sizeof(Tables::FsItem) == sizeof(FsItemD), "strip item layout");
            // Measured (round 4, MI355X, FP alone, strips against tiles; strips cut into segments of 16 tiles): 1024 x 512^2 x 90 2.53 / 2.87 ms,
            // 512^3 x 90 1.32 / 1.44, 128 x 1024^2 x 120 1.71 / 1.83, 1024^3 x 120 12.1 / 14.6 -- and 256 x 512^2 0.71 / 0.70, 64 x 512^2 0.235 / 0.196,
            // 256^3 x 60 0.143 / 0.122: an item is a sequential march of up to 16 tiles, and the form pays once a launch has several rounds
            // of them over the 512 resident workgroups: passes (~6.5 with the shear's overhang) x strips x segments x chunks >= 3000.
            // TOMO_FP_STRIP = 0 / 1 overrides the rule (tests build the tables at small sizes).
            const int fs_tiles = (e->n + FS_H - 1) / FS_H, fs_seglen = std::max(4, std::min(16, fs_tiles / 2));
            const double fs_wgs = 6.5 * ((double)e->n / FS_W) * ((fs_tiles + fs_seglen - 1) / fs_seglen) * (e->sxc / 64);
            bool want = e->n >= 384 && fs_wgs >= 3000.0;
            if (const char *env = std::getenv("TOMO_FP_STRIP")) want = std::atoi(env) != 0;
            e->fs_ok = want && build_fp_strips(m, e->n, e->np, 256, e->sxc / 64, t, why);
            if (e->fs_ok) {
                e->fs_nitems = (int)t.fs_item.size(); e->fs_kused = t.fs_kused; e->fs_nseg = t.fs_nseg;
                static_assert(sizeof(uint2) == sizeof(uint64_t), "entry layout");
                if ((rc = dev_alloc((void **)&e->d_fs_items, t.fs_item.size() * sizeof(FsItemD), false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fs_orient, t.fs_orient.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fs_shift, t.fs_shift.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fs_cnt, t.fs_cnt.size(), false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fs_gstart, t.fs_gstart.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fs_gseg0, t.fs_gseg0.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fs_ent, t.fs_ent_n * sizeof(uint2), false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fs_rsptr, t.fs_rseg_ptr.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fs_rsidx, t.fs_rseg_idx.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fs_zero, 256, true, e->stream))) return rc;
                HIPCHK(hipMemcpy(e->d_fs_items, t.fs_item.data(), t.fs_item.size() * sizeof(FsItemD), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fs_orient, t.fs_orient.data(), t.fs_orient.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fs_shift, t.fs_shift.data(), t.fs_shift.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fs_cnt, t.fs_cnt.data(), t.fs_cnt.size(), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fs_gstart, t.fs_gstart.data(), t.fs_gstart.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fs_gseg0, t.fs_gseg0.data(), t.fs_gseg0.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fs_ent, t.fs_ent.get(), t.fs_ent_n * sizeof(uint2), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fs_rsptr, t.fs_rseg_ptr.data(), t.fs_rseg_ptr.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fs_rsidx, t.fs_rseg_idx.data(), t.fs_rseg_idx.size() * 4, hipMemcpyHostToDevice));
            }
            t.fs_ent.reset(); t.fs_ent_n = 0; release(t.fs_cnt); release(t.fs_rseg_idx); release(t.fs_rseg_ptr); release(t.fs_gstart); release(t.fs_gseg0);
        }
        lap("build_fp_strips + upload");
        {   // the same strips as wave-uniform entry lists (k_fp_list); by the same rule (TOMO_FP_LIST = 0 / 1 overrides it)
            std::string why;
            static_assert(Tables::FL_W == FL_W && Tables::FL_TH == FL_TH && Tables::FL_WAVES == FL_WAVES && Tables::FL_BATCH == FL_BATCH && Tables::FL_PIXB == FL_PIXB, "list shape");
            static_assert(sizeof(Tables::FlItem) == sizeof(FlItemD), "list item layout");
            bool want = e->fs_ok;
            if (const char *env = std::getenv("TOMO_FP_LIST")) want = std::atoi(env) != 0;
            e->fl_ok = want && build_fp_lists(m, e->n, e->np, t, why);
            // The waves of a workgroup meet at a barrier after every tile, so a tile costs its busiest wave; build_fp_lists deals the rays
            // to the waves by load (fl_balance = mean / max batches per wave and tile: 0.89-0.90 on the BASELINE geometries).  Measured
            // (MI355X, FP alone, lists against strips): 512^3 x 90 1.12 / 1.22 ms, 512^3 x 70 0.92 / 1.06, 128 x 512^2 x 90 0.35 / 0.39,
            // 256^3 x 60 0.138 / 0.147, 128 x 1024^2 x 120 1.58 / 1.58.  A geometry that cannot be balanced keeps the strips.
            if (e->fl_ok && !std::getenv("TOMO_FP_LIST") && t.fl_balance < 0.8) e->fl_ok = false;
            if (e->fl_ok) {
                e->fl_nitems = (int)t.fl_item.size(); e->fl_nseg = t.fl_nseg;
                if ((rc = dev_alloc((void **)&e->d_fl_items, t.fl_item.size() * sizeof(FlItemD), false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fl_orient, t.fl_orient.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fl_shift, t.fl_shift.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fl_ent, t.fl_ent_n * sizeof(uint2), false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fl_ptr, t.fl_ptr.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fl_fent, t.fl_flush.size() * sizeof(uint2), false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fl_fptr, t.fl_fptr.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fl_rsptr, t.fl_rseg_ptr.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fl_rsidx, t.fl_rseg_idx.size() * 4, false, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->d_fl_zero, 512, true, e->stream))) return rc;
                HIPCHK(hipMemcpy(e->d_fl_items, t.fl_item.data(), t.fl_item.size() * sizeof(FlItemD), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fl_orient, t.fl_orient.data(), t.fl_orient.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fl_shift, t.fl_shift.data(), t.fl_shift.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fl_ent, t.fl_ent.get(), t.fl_ent_n * sizeof(uint2), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fl_ptr, t.fl_ptr.data(), t.fl_ptr.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fl_fent, t.fl_flush.data(), t.fl_flush.size() * sizeof(uint2), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fl_fptr, t.fl_fptr.data(), t.fl_fptr.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fl_rsptr, t.fl_rseg_ptr.data(), t.fl_rseg_ptr.size() * 4, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(e->d_fl_rsidx, t.fl_rseg_idx.data(), t.fl_rseg_idx.size() * 4, hipMemcpyHostToDevice));
            }
            t.fl_ent.reset(); t.fl_ent_n = 0; release(t.fl_flush); release(t.fl_ptr); release(t.fl_fptr); release(t.fl_rseg_ptr); release(t.fl_rseg_idx); release(t.fl_item); release(t.fl_shift);
        }
        lap("build_fp_lists + upload");
        build_sart_tiles(m, e->n, e->np, ST_TY, ST_TZ, ST_MAXR, 256, t);
        static_assert(Tables::ST_MAXSEG == ST_MAXSEG, "segment slots per tile");
        e->st_ok = t.st_ok;
        if (e->st_ok) {
            e->st_ntiles = t.st_tiles; e->st_tiles_z = t.st_tiles_z; e->st_max_ids = t.st_max_ids;
            std::vector<uint2> sent(t.st_off.size());
            for (size_t k = 0; k < sent.size(); ++k) { uint32_t bits; std::memcpy(&bits, &t.st_w[k], 4); sent[k] = make_uint2(t.st_off[k], bits); }
            if ((rc = dev_alloc((void **)&e->d_st_cell, t.st_cell.size() * sizeof(uint4), false, e->stream))) return rc;
            if ((rc = dev_alloc((void **)&e->d_st_win, t.st_win.size() * 4, false, e->stream))) return rc;
            if ((rc = dev_alloc((void **)&e->d_st_segid, t.st_segid.size() * 4, false, e->stream))) return rc;
            if ((rc = dev_alloc((void **)&e->d_st_seg, t.st_seg.size() * 4, false, e->stream))) return rc;
            if ((rc = dev_alloc((void **)&e->d_st_ent, sent.size() * sizeof(uint2), false, e->stream))) return rc;
            if ((rc = dev_alloc((void **)&e->d_st_row_first, t.st_row_first.size() * 4, false, e->stream))) return rc;
            if ((rc = dev_alloc((void **)&e->d_st_row_nseg, t.st_row_nseg.size() * 4, false, e->stream))) return rc;
            HIPCHK(hipMemcpy(e->d_st_cell, t.st_cell.data(), t.st_cell.size() * sizeof(uint4), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_st_win, t.st_win.data(), t.st_win.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_st_segid, t.st_segid.data(), t.st_segid.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_st_seg, t.st_seg.data(), t.st_seg.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_st_ent, sent.data(), sent.size() * sizeof(uint2), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_st_row_first, t.st_row_first.data(), t.st_row_first.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_st_row_nseg, t.st_row_nseg.data(), t.st_row_nseg.size() * 4, hipMemcpyHostToDevice));
        }
        release(t.st_cell); release(t.st_off); release(t.st_w); release(t.st_seg); release(t.st_segid); release(t.st_win);
        lap("build_sart_tiles + upload");
        {   // tables of the volume-resident sweep (resident.cpp); TOMO_SART_RESIDENT = 0 leaves them out
            static_assert(Resident::T == RS_T && Resident::WAVES == RS_WAVES && Resident::MAXWIN == RS_MAXWIN && Resident::RL == RS_RL && Resident::USABLE == RS_USABLE &&
                          Resident::TSN == 8 && Resident::SINK == 14 && sizeof(Resident::Hdr) == sizeof(RsHdrD), "k_sart_resident geometry (resident.h)");
            bool want = e->n % 8 == 0;
            if (const char *env = std::getenv("TOMO_SART_RESIDENT")) want = want && std::atoi(env) != 0;
            e->rs_ok = false;
            if (want) {
                hipDeviceProp_t prop;
                HIPCHK(hipGetDeviceProperties(&prop, e->device));
                e->rs_cus = prop.multiProcessorCount;
                Resident R;
                build_sart_resident(e->n, e->np, t, e->rs_cus, R);
                // every workgroup of a launch must be on the chip at once: the runtime's own count of workgroups per CU for this kernel
                // (registers, LDS) has to cover the launch; and the chunks of a sweep that could not finish go to the tile chain
                int per_cu = 0;
                if (R.ok) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_sart_resident, RS_THREADS, 0));
                if (R.ok && e->st_ok && (int64_t)per_cu * e->rs_cus >= R.ntiles) {
                    e->rs_ntiles = R.ntiles; e->rs_tiles = R.tiles; e->rs_rpt = R.rpt;
                    e->rs_groups = std::max(1, std::min(per_cu * e->rs_cus / R.ntiles, e->sxc / 64));
                    e->rs_pb_bytes = (size_t)e->rs_groups * R.ntiles * RS_MAXWIN * 64 * sizeof(rs_u64);
                    e->rs_rb_bytes = (size_t)e->rs_groups * e->np * e->n * 64 * sizeof(rs_u64);
                    if ((rc = dev_alloc((void **)&e->d_rs_hdr, R.hdr.size() * sizeof(RsHdrD), false, e->stream))) return rc;
                    if ((rc = dev_alloc((void **)&e->d_rs_cell, R.cell.size() * 4, false, e->stream))) return rc;
                    if ((rc = dev_alloc((void **)&e->d_rs_ts, R.ts.size(), false, e->stream))) return rc;
                    if ((rc = dev_alloc((void **)&e->d_rs_rl, R.rl.size() * 2, false, e->stream))) return rc;
                    if ((rc = dev_alloc((void **)&e->rs_pb, e->rs_pb_bytes, true, e->stream))) return rc;     // tag 0 = never written
                    if ((rc = dev_alloc((void **)&e->rs_rb, e->rs_rb_bytes, true, e->stream))) return rc;
                    HIPCHK(hipMemcpy(e->d_rs_hdr, R.hdr.data(), R.hdr.size() * sizeof(RsHdrD), hipMemcpyHostToDevice));
                    HIPCHK(hipMemcpy(e->d_rs_cell, R.cell.data(), R.cell.size() * 4, hipMemcpyHostToDevice));
                    HIPCHK(hipMemcpy(e->d_rs_ts, R.ts.data(), R.ts.size(), hipMemcpyHostToDevice));
                    HIPCHK(hipMemcpy(e->d_rs_rl, R.rl.data(), R.rl.size() * 2, hipMemcpyHostToDevice));
                    if (!e->rs_abort) { HIPCHK(hipHostMalloc((void **)&e->rs_abort, sizeof(int), hipHostMallocMapped)); *e->rs_abort = 0; }
                    if ((rc = dev_alloc((void **)&e->d_rs_abort, sizeof(int), true, e->stream))) return rc;
                    if ((rc = dev_alloc((void **)&e->d_rs_commit, (size_t)(e->sxc / 64) * sizeof(unsigned), true, e->stream))) return rc;
                    if (!e->rs_done) { HIPCHK(hipHostMalloc((void **)&e->rs_done, (size_t)(e->sxc / 64) * sizeof(int), hipHostMallocMapped)); }
                    std::memset(e->rs_done, 0, (size_t)(e->sxc / 64) * sizeof(int));
                    e->rs_epoch = 0; e->rs_seq = 0; e->rs_commit_base = 0; e->rs_commit_dirty = false;
                    e->rs_ok = true;
                }
            }
        }
        lap("build_sart_resident + upload");
        build_bp_tiles(e->n, e->np, FT_TY, FT_TZ, FB_A, FB_MAXR, 256, 2 * FB_A, t);   // the cell ring prefetches up to angle P + 2*FB_A - 2
        static_assert(sizeof(Tables::TileCell) == sizeof(uint4), "tile cell layout");
        e->fb_ok = t.bp_tile_ok && e->np <= FB_MAX_PROJ;
        e->bl_ok = false;
        if (e->fb_ok) {
            if ((rc = dev_alloc((void **)&e->d_fb_cell, t.bp_cell.size() * sizeof(uint4), false, e->stream))) return rc;
            if ((rc = dev_alloc((void **)&e->d_fb_win, t.bp_win.size() * 4, false, e->stream))) return rc;
            HIPCHK(hipMemcpy(e->d_fb_cell, t.bp_cell.data(), t.bp_cell.size() * sizeof(uint4), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_fb_win, t.bp_win.data(), t.bp_win.size() * 4, hipMemcpyHostToDevice));
            release(t.bp_cell);
        }
        lap("build_bp_tiles + upload");
        static_assert(BL_TY == Tables::BL_TY && BL_TZ == Tables::BL_TZ && BL_WAVES == Tables::BL_WAVES && BL_A == Tables::BL_A && BL_MAXR == Tables::BL_MAXR &&
                      BL_ROWB == Tables::BL_ROWB && BL_BATCH == Tables::BL_BATCH, "k_bp_list geometry (sysmat.h)");
        build_bp_lists(e->n, e->np, BL_TY, BL_TZ, BL_A, BL_MAXR, BL_ROWB, BL_WAVES, BL_BATCH, Tables::BL_REGS, t);
        // (k_bp_list keeps a stage's list bounds and window words per lane; its staging offsets are 32-bit)
        e->bl_ok = t.bl_ok && (e->np + BL_A - 1) / BL_A <= 64;
        if (e->bl_ok) {
            const size_t nent = (size_t)(t.bl_nbatch + 1) * BL_BATCH;
            e->bl_tiles_z = (e->n + BL_TZ - 1) / BL_TZ;
            e->bl_ntiles = ((e->n + BL_TY - 1) / BL_TY) * e->bl_tiles_z;
            if ((rc = dev_alloc((void **)&e->d_bl_ent, nent * sizeof(uint4), false, e->stream))) return rc;
            if ((rc = dev_alloc((void **)&e->d_bl_ptr, t.bl_ptr.size() * 4, false, e->stream))) return rc;
            if ((rc = dev_alloc((void **)&e->d_bl_win, t.bl_win.size() * 4, false, e->stream))) return rc;
            HIPCHK(hipMemcpy(e->d_bl_ent, t.bl_ent.get(), nent * sizeof(uint4), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_bl_ptr, t.bl_ptr.data(), t.bl_ptr.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(e->d_bl_win, t.bl_win.data(), t.bl_win.size() * 4, hipMemcpyHostToDevice));
            t.bl_ent.reset();
        }
        release(t.bl_win); release(t.bl_ptr);
    }
    lap("build_bp_lists + upload");
    static_assert(sizeof(Cell) == sizeof(CellD), "cell layout");
    HIPCHK(hipMemcpy(e->d_cell, t.cell.data(), t.cell.size() * sizeof(CellD), hipMemcpyHostToDevice));
    g_alloc_meter = nullptr;                       // what follows are fields, not tables
    if ((rc = dev_alloc((void **)&e->d_scal_own, TOMO_S_COUNT * sizeof(double), true, e->stream))) return rc;
    if ((rc = dev_alloc((void **)&e->d_part, NPART * sizeof(double), true, e->stream))) return rc;
    e->d_scal = e->d_scal_own;
    if ((rc = dev_alloc((void **)&e->halo_lo_own, e->npix * sizeof(float), true, e->stream))) return rc;
    if ((rc = dev_alloc((void **)&e->halo_hi_own, e->npix * sizeof(float), true, e->stream))) return rc;
    e->halo_lo = e->halo_lo_own; e->halo_hi = e->halo_hi_own;
    float *tmp;
    if ((rc = get_vol(e, TOMO_VOL_RECON, &tmp))) return rc;
    if ((rc = get_sino(e, &e->sino[TOMO_SINO_B], &tmp))) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    *out = e;
    return TOMO_OK;
}

extern "C" int tomo_destroy(tomo_engine *e);

// a half-built engine is torn down again (device buffers, stream) and the first error is the one reported
static int finish_create(tomo_engine *e, Coo &m, tomo_engine **out, double t_begin)
{
    int rc = finish_create_impl(e, m, out);
    e->create_ms = (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_begin) * 1e3;
    if (rc) {
        std::string first = g_err;
        tomo_destroy(e);
        g_err = first;
    }
    return rc;
}

static tomo_engine *new_engine(int nslice, int nray, int nproj, int device)
{
    tomo_engine *e = new tomo_engine();
    e->nx = nslice; e->n = nray; e->np = nproj; e->device = device;
    e->sxc = ((nslice + 63) / 64) * 64;
    e->vec = (e->sxc % 256 == 0) ? 4 : (e->sxc % 128 == 0) ? 2 : 1;
    // Row pitch = computed width.  A pixel's row of slices that is a multiple of 4 KB (1024 slices) makes two sweep chains striding over
    // alternate halves of the rows alias the memory channels (round 2: 51.3 against 40.6 ms per sweep at 1024 slices), so such a slab
    // runs ONE chain (chain_count).  Round 4 measured 64 slices of padding on the pitch so that it can run two: 1024^3 x 120 ASD-POCS
    // step 208.7 ms against 206.6 ms with one chain on the plain pitch -- no gain; the environment hook of that experiment is gone
    // (round 5, ADVICE r4: several launches size their work from the pitch, so a padded pitch was not safe to ship).
    e->sx = e->sxc;
    e->npix = (int64_t)nray * nray;
    e->nrows = (int64_t)nray * nproj;
    return e;
}

static int check_dims(int nslice, int nray, int nproj)
{
    if (nslice <= 0 || nray <= 0 || nproj <= 0) return fail(TOMO_ERR_ARG, "Nslice, Nray and Nproj must be positive");
    if (nray > 4096) return fail(TOMO_ERR_ARG, "Nray > 4096 not supported (float32 index storage limit of parallelRay)");
    return TOMO_OK;
}

// A sweep over a slab as one chain of launches on the engine's stream, or as two chains over two sub-slabs (64-slice chunks,
// equal halves) on two streams: slices are independent, and each stream's short kernels and launch boundaries disappear under
// the other stream's tile step.  A sweep is then 2 x ~180 launches of ~100 us kernels: one host thread cannot enqueue both chains
// fast enough to keep both streams fed (measured: 19.4 ms per sweep against 19.9 on one stream), so the second chain is enqueued
// by a second host thread (18.0 ms: what two independent engines on two Python threads reach).
// "sart_streams": 2 = always (when the slab has two chunks), 1 = never, 0 = auto: equal halves, and not when a pixel's row of
// slices is a multiple of 4 KB -- the two halves of such rows land on the same memory channels (1024 slices: 48.5 against 42.9 ms
// per ASD-POCS step; 128 / 256 / 512 / 768 slices: -2.6 / -3.3 / -5.5 / -5.3 %).
// chains a sweep of this engine's slab runs as, under the current "sart_streams" (also what tomo_sart_chain_count reports)
static int chain_count(const tomo_engine *e)
{
    const int units = e->sxc / 64;
    const bool two = e->sart_streams >= 2 || (e->sart_streams == 0 && units % 2 == 0 && (e->sx * sizeof(float)) % 4096 != 0);
    if (!(two && units >= 2)) return 1;
    if (e->sart_streams > 2) return std::min(std::min(e->sart_streams, (int)tomo_engine::MAX_CHAINS), units);
    return 2;   // "sart_streams" = 3 / 4: that many chains (measured at 512 slices: 2 chains 22.4 ms per step, 3: 24.1, 4: 22.7)
}

template <class Chain>
static int run_chains(tomo_engine *e, const Chain &chain)
{
    const int units = e->sxc / 64;                       // 64-slice chunks; a sub-slab's per-row kernels use the widest vector that fits
    auto vec_of = [](int c0, int nc) { return (c0 % 4 == 0 && nc % 4 == 0) ? 4 : (c0 % 2 == 0 && nc % 2 == 0) ? 2 : 1; };
    const int nch = chain_count(e);
    if (nch < 2) return chain(whole(e));
    if (!e->ev_sfork) HIPCHK(hipEventCreateWithFlags(&e->ev_sfork, hipEventDisableTiming));
    for (int u = 0; u < nch; ++u)
        if (!e->sub_stream[u]) {
            HIPCHK(hipStreamCreateWithFlags(&e->sub_stream[u], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&e->ev_sjoin[u], hipEventDisableTiming));
        }
    HIPCHK(hipEventRecord(e->ev_sfork, e->stream));
    Sub sbs[tomo_engine::MAX_CHAINS];
    for (int u = 0, c0 = 0; u < nch; ++u) {
        const int nc = units / nch + (u < units % nch ? 1 : 0);
        sbs[u] = Sub{e->sub_stream[u], c0, nc, vec_of(c0, nc)};
        c0 += nc;
    }
    for (int u = 0; u < nch; ++u) HIPCHK(hipStreamWaitEvent(e->sub_stream[u], e->ev_sfork, 0));
    int rcs[tomo_engine::MAX_CHAINS] = {TOMO_OK, TOMO_OK, TOMO_OK, TOMO_OK};
    std::string errs[tomo_engine::MAX_CHAINS];
    for (int u = 1; u < nch; ++u) {                        // chains 1.. are enqueued by the engine's persistent helper threads
        if (!e->chain_helper[u]) {
            e->chain_helper[u].reset(new ChainHelper());
            e->chain_helper[u]->start(e->device);
        }
        e->chain_helper[u]->submit([&, u]() {
            rcs[u] = chain(sbs[u]);
            if (rcs[u]) errs[u] = g_err;                  // the error text is thread-local
        });
    }
    rcs[0] = chain(sbs[0]);
    for (int u = 1; u < nch; ++u) e->chain_helper[u]->wait();
    for (int u = 1; u < nch; ++u) if (rcs[u] && !rcs[0]) { rcs[0] = rcs[u]; g_err = errs[u]; }
    for (int u = 0; u < nch; ++u)
        if (hipEventRecord(e->ev_sjoin[u], e->sub_stream[u]) != hipSuccess && !rcs[0]) rcs[0] = fail(TOMO_ERR_HIP, "hipEventRecord(sub-slab join)");
    for (int u = 0; u < nch; ++u) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_sjoin[u], 0));
    return rcs[0];
}

extern "C" {

const char *tomo_last_error(void) { return g_err.c_str(); }

int tomo_device_count(int *count)
{
    if (!count) return fail(TOMO_ERR_ARG, "null count");
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) { n = 0; (void)hipGetLastError(); }
    *count = n;
    return TOMO_OK;
}

int tomo_system_matrix(int nray, int nproj, const double *angles_rad, int64_t cap, float *rows, float *cols,
                       float *vals, int64_t *nnz)
{
    if (!angles_rad || !nnz) return fail(TOMO_ERR_ARG, "null argument");
    int rc = check_dims(1, nray, nproj);
    if (rc) return rc;
    Coo m;
    build_parallel_ray(nray, nproj, angles_rad, m);
    *nnz = m.ptr[m.nrow];
    if (cap == 0) return TOMO_OK;
    if (cap < *nnz || !rows || !cols || !vals) return fail(TOMO_ERR_ARG, "output capacity too small");
    for (int64_t r = 0; r < m.nrow; ++r)
        for (int64_t k = m.ptr[r]; k < m.ptr[r + 1]; ++k) { rows[k] = (float)r; cols[k] = (float)m.col[k]; vals[k] = m.val[k]; }
    return TOMO_OK;
}

int tomo_create(int nslice, int nray, int nproj, const double *angles_rad, int device, tomo_engine **out)
{
    if (!angles_rad || !out) return fail(TOMO_ERR_ARG, "null argument");
    int rc = check_dims(nslice, nray, nproj);
    if (rc) return rc;
    tomo_engine *e = new_engine(nslice, nray, nproj, device);
    const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    Coo m;
    build_parallel_ray(nray, nproj, angles_rad, m);
    return finish_create(e, m, out, t_begin);
}

int tomo_create_from_matrix(int nslice, int nray, int nproj, int64_t nnz, const float *rows, const float *cols,
                            const float *vals, int device, tomo_engine **out)
{
    if (!rows || !cols || !vals || !out) return fail(TOMO_ERR_ARG, "null argument");
    int rc = check_dims(nslice, nray, nproj);
    if (rc) return rc;
    const double t_begin = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    Coo m;
    std::string err;
    if (!coo_from_triplets((int64_t)nray * nproj, (int64_t)nray * nray, nnz, rows, cols, vals, m, err))
        return fail(TOMO_ERR_ARG, err);
    tomo_engine *e = new_engine(nslice, nray, nproj, device);
    return finish_create(e, m, out, t_begin);
}

// every device buffer that depends on the tilt geometry (tables, sinograms, scratch sized by it); volumes stay
static void free_geometry(tomo_engine *e)
{
    void **ptrs[] = {(void **)&e->d_st_cell, (void **)&e->d_st_win, (void **)&e->d_st_segid, (void **)&e->d_st_seg, (void **)&e->d_st_ent,
                     (void **)&e->d_st_row_first, (void **)&e->d_st_row_nseg, (void **)&e->st_partial, (void **)&e->st_partial2, (void **)&e->st_flags, (void **)&e->d_fb_cell, (void **)&e->d_fb_win, (void **)&e->d_bl_ent, (void **)&e->d_bl_ptr, (void **)&e->d_bl_win,
                     (void **)&e->d_ft_slot_ptr, (void **)&e->d_ft_slot_seg0, (void **)&e->d_ft_tent, (void **)&e->d_ft_rsptr, (void **)&e->d_ft_rsidx,
                     (void **)&e->ft_part, (void **)&e->ft_part_aux, (void **)&e->cg_w, (void **)&e->fbp_h, (void **)&e->d_seg_exec,
                     (void **)&e->d_row_first, (void **)&e->d_row_nseg, (void **)&e->seg_partial, (void **)&e->d_wptr, (void **)&e->d_went,
                     (void **)&e->d_rptr, (void **)&e->d_rent, (void **)&e->d_rowsum, (void **)&e->d_rowinner, (void **)&e->d_colsum_all,
                     (void **)&e->d_rowcross, (void **)&e->d_cell,
                     (void **)&e->d_fs_items, (void **)&e->d_fs_orient, (void **)&e->d_fs_shift, (void **)&e->d_fs_cnt, (void **)&e->d_fs_gstart,
                     (void **)&e->d_fs_gseg0, (void **)&e->d_fs_ent, (void **)&e->d_fs_zero, (void **)&e->d_fs_rsptr, (void **)&e->d_fs_rsidx, (void **)&e->fs_part, (void **)&e->fs_part_aux,
                     (void **)&e->d_fl_items, (void **)&e->d_fl_orient, (void **)&e->d_fl_shift, (void **)&e->d_fl_ent, (void **)&e->d_fl_ptr, (void **)&e->d_fl_fent, (void **)&e->d_fl_fptr,
                     (void **)&e->d_fl_rsptr, (void **)&e->d_fl_rsidx, (void **)&e->d_fl_zero, (void **)&e->fl_part, (void **)&e->fl_part_aux,
                     (void **)&e->d_rs_hdr, (void **)&e->d_rs_cell, (void **)&e->d_rs_ts, (void **)&e->d_rs_rl, (void **)&e->rs_pb, (void **)&e->rs_rb, (void **)&e->d_rs_angs, (void **)&e->d_rs_abort, (void **)&e->d_rs_commit};
    for (void **p : ptrs) if (*p) { (void)hipFree(*p); *p = nullptr; }
    e->rs_ok = false; e->rs_angs_cap = 0; e->rs_angs_host.clear();
    for (int i = 0; i < TOMO_SINO_SLOTS; ++i) if (e->sino[i]) { (void)hipFree(e->sino[i]); e->sino[i] = nullptr; }
    if (e->g_prev) { (void)hipFree(e->g_prev); e->g_prev = nullptr; }
    if (e->g_yk) { (void)hipFree(e->g_yk); e->g_yk = nullptr; }
    e->yk_claim.valid = false;
    e->g_prev_valid = e->mom_p_ok = e->mom.set = false;
    e->geometry_released = true;
}

int tomo_destroy(tomo_engine *e)
{
    if (!e) return TOMO_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->aux) { (void)hipStreamSynchronize(e->aux); (void)hipStreamDestroy(e->aux); (void)hipEventDestroy(e->ev_fork); (void)hipEventDestroy(e->ev_join); }
    for (int u = 0; u < tomo_engine::MAX_CHAINS; ++u) if (e->sub_stream[u]) { (void)hipStreamSynchronize(e->sub_stream[u]); (void)hipStreamDestroy(e->sub_stream[u]); (void)hipEventDestroy(e->ev_sjoin[u]); }
    for (int w = 0; w < 2; ++w) if (e->fp_red_stream[w]) { (void)hipStreamSynchronize(e->fp_red_stream[w]); (void)hipStreamDestroy(e->fp_red_stream[w]); for (int h = 0; h < 2; ++h) { (void)hipEventDestroy(e->ev_fp_tile[w][h]); (void)hipEventDestroy(e->ev_fp_red[w][h]); } }
    comm_release(e);
    if (e->ev_sfork) (void)hipEventDestroy(e->ev_sfork);
    if (e->ev_peer) (void)hipEventDestroy(e->ev_peer);
    if (e->ev_snap) (void)hipEventDestroy(e->ev_snap);
    if (e->h_snap) (void)hipHostFree(e->h_snap);
    if (e->rs_abort) (void)hipHostFree(e->rs_abort);
    if (e->rs_done) (void)hipHostFree(e->rs_done);
    free_geometry(e);
    void *ptrs[] = {e->tv_alt, e->halo_lo_alt, e->halo_hi_alt, e->d_part_tv, e->d_part_aux, e->fgp_q[0], e->fgp_q[1], e->fgp_q[2], e->cg_p, e->cg_z, e->cg_sums, e->cg_part, e->cg_coef, e->sart_alt,
                    e->tvg, e->fgp_p[0], e->fgp_p[1], e->fgp_p[2], e->stage, e->d_scal_own, e->d_part, e->halo_lo_own, e->halo_hi_own};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (int i = 0; i < TOMO_VOL_SLOTS; ++i) if (e->vol[i]) (void)hipFree(e->vol[i]);
    for (auto &p : e->prof) { for (auto ev : p.ev) (void)hipEventDestroy(ev); if (p.ref) (void)hipEventDestroy(p.ref); }
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return TOMO_OK;
}

// Rebuilding the tilt geometry with the reconstruction kept (tomoengine::update_projection_angles, tomoengine.cpp:128-149;
// ctvlib::update_proj_angles, ctvlib.cpp:317-333): release the old engine's tables and sinograms, create the new engine,
// let it adopt the old engine's volumes (pointers move, nothing is copied), destroy the old engine.
int tomo_release_geometry(tomo_engine *e)
{
    NEED(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->aux) HIPCHK(hipStreamSynchronize(e->aux));
    for (int u = 0; u < tomo_engine::MAX_CHAINS; ++u) if (e->sub_stream[u]) HIPCHK(hipStreamSynchronize(e->sub_stream[u]));
    e->async_pending = false;
    free_geometry(e);
    return TOMO_OK;
}

int tomo_adopt_volumes(tomo_engine *dst, tomo_engine *src)
{
    if (!dst || !src) return fail(TOMO_ERR_ARG, "null engine");
    if (dst->nx != src->nx || dst->n != src->n || dst->sx != src->sx || dst->device != src->device) return fail(TOMO_ERR_ARG, "engines differ in slab shape or device");
    HIPCHK(hipSetDevice(dst->device));
    HIPCHK(hipStreamSynchronize(src->stream));
    if (src->aux) HIPCHK(hipStreamSynchronize(src->aux));       // nothing of src may still be reading the volumes that move
    for (int u = 0; u < tomo_engine::MAX_CHAINS; ++u) if (src->sub_stream[u]) HIPCHK(hipStreamSynchronize(src->sub_stream[u]));
    src->async_pending = false;
    HIPCHK(hipStreamSynchronize(dst->stream));
    for (int i = 0; i < TOMO_VOL_SLOTS; ++i) {
        if (!src->vol[i]) continue;
        if (dst->vol[i]) HIPCHK(hipFree(dst->vol[i]));
        dst->vol[i] = src->vol[i];
        src->vol[i] = nullptr;
    }
    dst->old_is_recon = src->old_is_recon;
    src->old_is_recon = false;
    for (int i = 0; i < TOMO_VOL_SLOTS; ++i) ++dst->vol_version[i];
    g_clear(dst);
    return TOMO_OK;
}

int tomo_set_stream(tomo_engine *e, void *hip_stream)
{
    NEED(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->own_stream) { HIPCHK(hipStreamDestroy(e->stream)); e->own_stream = false; }
    e->stream = (hipStream_t)hip_stream;
    return TOMO_OK;
}

int tomo_synchronize(tomo_engine *e) { NEED(e); HIPCHK(hipStreamSynchronize(e->stream)); return TOMO_OK; }
int tomo_get_device(tomo_engine *e, int *device) { if (!e || !device) return fail(TOMO_ERR_ARG, "null"); *device = e->device; return TOMO_OK; }
int tomo_get_dims(tomo_engine *e, int *nslice, int *nray, int *nproj, int64_t *nnz)
{
    if (!e) return fail(TOMO_ERR_ARG, "null engine");
    if (nslice) *nslice = e->nx;
    if (nray) *nray = e->n;
    if (nproj) *nproj = e->np;
    if (nnz) *nnz = e->nnz;
    return TOMO_OK;
}

int tomo_sart_chain_count(tomo_engine *e, int *count)
{
    if (!e || !count) return fail(TOMO_ERR_ARG, "null");
    *count = chain_count(e);
    return TOMO_OK;
}

// ---- data in / out ---------------------------------------------------------------------------------------
static int upload(tomo_engine *e, const float *host, float *dst, int64_t m)
{
    size_t bytes = (size_t)e->nx * m * sizeof(float);
    int rc = ensure_stage(e, bytes);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(e->stage, host, bytes, hipMemcpyHostToDevice, e->stream));
    dim3 grid((unsigned)((m + 31) / 32), (unsigned)((e->sx + 31) / 32)), block(256);
    hipLaunchKernelGGL(k_transpose_in, grid, block, 0, e->stream, e->stage, dst, e->nx, m, e->sx);
    LAUNCHCHK();
    HIPCHK(hipStreamSynchronize(e->stream));
    return TOMO_OK;
}

static int download(tomo_engine *e, const float *src, float *host, int64_t m)
{
    size_t bytes = (size_t)e->nx * m * sizeof(float);
    int rc = ensure_stage(e, bytes);
    if (rc) return rc;
    dim3 grid((unsigned)((m + 31) / 32), (unsigned)((e->sx + 31) / 32)), block(256);
    hipLaunchKernelGGL(k_transpose_out, grid, block, 0, e->stream, src, e->stage, e->nx, m, e->sx);
    LAUNCHCHK();
    HIPCHK(hipMemcpyAsync(host, e->stage, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TOMO_OK;
}

int tomo_set_sinogram(tomo_engine *e, int which, const float *b)
{
    NEED(e);
    if (!b) return fail(TOMO_ERR_ARG, "null sinogram");
    float *dst; int rc = sino_slot(e, which, &dst); if (rc) return rc;
    return upload(e, b, dst, e->nrows);
}

int tomo_set_tilt_series(tomo_engine *e, const float *b) { return tomo_set_sinogram(e, TOMO_SINO_B, b); }

int tomo_get_sinogram(tomo_engine *e, int which, float *out)
{
    NEED(e);
    if (!out) return fail(TOMO_ERR_ARG, "null output");
    if (which == TOMO_SINO_YK_MODEL) {                   // the extrapolated point's projection, while it is one
        const float *yk = projection_in_hand(e, TOMO_VOL_YK);
        if (!yk) return fail(TOMO_ERR_STATE, "no projection of the extrapolated point in hand (tomo_fista_project_yk)");
        return download(e, yk, out, e->nrows);
    }
    float *src;
    int rc = sino_slot(e, which, &src);
    if (rc) return rc;
    return download(e, src, out, e->nrows);
}

int tomo_set_volume(tomo_engine *e, int vol, const float *data)
{
    NEED(e);
    { int rc_ = order_after_async(e); if (rc_) return rc_; }
    float *dst; int rc = get_vol(e, vol, &dst); if (rc) return rc;
    if (!data) return fail(TOMO_ERR_ARG, "null volume");
    return upload(e, data, dst, e->npix);
}

int tomo_get_volume(tomo_engine *e, int vol, float *data)
{
    NEED(e);
    float *src; int rc = get_vol_ro(e, vol, &src); if (rc) return rc;
    if (!data) return fail(TOMO_ERR_ARG, "null volume");
    return download(e, src, data, e->npix);
}

int tomo_set_slice(tomo_engine *e, int vol, int s, const float *img)
{
    NEED(e);
    { int rc_ = order_after_async(e); if (rc_) return rc_; }
    float *dst; int rc = get_vol(e, vol, &dst); if (rc) return rc;
    if (s < 0 || s >= e->nx || !img) return fail(TOMO_ERR_ARG, "slice index out of range");
    if ((rc = ensure_stage(e, e->npix * sizeof(float)))) return rc;
    HIPCHK(hipMemcpyAsync(e->stage, img, e->npix * sizeof(float), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_scatter_slice, dim3((unsigned)((e->npix + 255) / 256)), dim3(256), 0, e->stream, e->stage, dst, e->npix, e->sx, s);
    LAUNCHCHK();
    HIPCHK(hipStreamSynchronize(e->stream));
    return TOMO_OK;
}

int tomo_get_slice(tomo_engine *e, int vol, int s, float *img)
{
    NEED(e);
    float *src; int rc = get_vol_ro(e, vol, &src); if (rc) return rc;
    if (s < 0 || s >= e->nx || !img) return fail(TOMO_ERR_ARG, "slice index out of range");
    if ((rc = ensure_stage(e, e->npix * sizeof(float)))) return rc;
    hipLaunchKernelGGL(k_gather_slice, dim3((unsigned)((e->npix + 255) / 256)), dim3(256), 0, e->stream, src, e->stage, e->npix, e->sx, s);
    LAUNCHCHK();
    HIPCHK(hipMemcpyAsync(img, e->stage, e->npix * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TOMO_OK;
}

int tomo_restart_recon(tomo_engine *e)
{
    NEED(e);
    { int rc_ = order_after_async(e); if (rc_) return rc_; }
    HIPCHK(hipMemsetAsync(e->vol[TOMO_VOL_RECON], 0, e->vol_elems() * sizeof(float), e->stream));
    if (e->vol[TOMO_VOL_YK]) HIPCHK(hipMemsetAsync(e->vol[TOMO_VOL_YK], 0, e->vol_elems() * sizeof(float), e->stream));
    if (e->vol[TOMO_VOL_RECON_OLD]) HIPCHK(hipMemsetAsync(e->vol[TOMO_VOL_RECON_OLD], 0, e->vol_elems() * sizeof(float), e->stream));
    e->old_is_recon = false;                       // every buffer is physically zero
    ++e->vol_version[TOMO_VOL_RECON]; ++e->vol_version[TOMO_VOL_YK]; ++e->vol_version[TOMO_VOL_RECON_OLD];
    return TOMO_OK;
}

int tomo_copy_volume(tomo_engine *e, int dst, int src)
{
    NEED(e);
    { int rc_ = order_after_async(e); if (rc_) return rc_; }
    float *d, *s; int rc;
    const bool g_src = g_is_projection_of(e, src);
    if ((rc = get_vol(e, dst, &d)) || (rc = get_vol_ro(e, src, &s))) return rc;
    if (d != s) HIPCHK(hipMemcpyAsync(d, s, e->vol_elems() * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    if (g_src && dst != src) { e->g_valid[1].vol = dst; e->g_valid[1].ver = e->vol_version[dst]; }   // G = A * src = A * dst
    return TOMO_OK;
}

// ---- projector ------------------------------------------------------------------------------------------------
int tomo_forward_projection(tomo_engine *e, int vol, int sino)
{
    NEED(e);
    { int rc_ = order_after_async(e); if (rc_) return rc_; }
    float *x, *g; int rc;
    if ((rc = get_vol_ro(e, vol, &x))) return rc;
    if ((rc = sino_slot(e, sino, &g))) return rc;
    if ((rc = launch_fp_all<FP_STORE>(e, x, nullptr, g))) return rc;
    if (sino == TOMO_SINO_G) g_set(e, vol);
    return TOMO_OK;
}

int tomo_back_projection(tomo_engine *e, int sino, int vol)
{
    NEED(e);
    float *x, *g; int rc;
    if ((rc = get_vol(e, vol, &x))) return rc;
    if ((rc = sino_slot(e, sino, &g))) return rc;
    return launch_bp_all(e, x, g, nullptr, 0.f, 1.f, 0);
}

int tomo_lipschitz(tomo_engine *e, float *L) { if (!e || !L) return fail(TOMO_ERR_ARG, "null"); *L = e->lipschitz; return TOMO_OK; }
int tomo_lipschitz_cimmino(tomo_engine *e, float *L) { if (!e || !L) return fail(TOMO_ERR_ARG, "null"); *L = e->lipschitz_cimmino; return TOMO_OK; }
int tomo_row_inner_product(tomo_engine *e) { if (!e) return fail(TOMO_ERR_ARG, "null engine"); return TOMO_OK; /* built with the tables */ }

// ---- reconstruction steps -----------------------------------------------------------------------------------------
int tomo_sirt_landweber(tomo_engine *e, int vol, float beta, int niter)
{
    NEED(e);
    float *x, *r; int rc;
    if ((rc = get_vol(e, vol, &x)) || (rc = get_sino(e, &e->sino[TOMO_SINO_R], &r))) return rc;
    for (int it = 0; it < niter; ++it) {
        if ((rc = launch_fp_all<FP_RESID>(e, x, e->sino[TOMO_SINO_B], r))) return rc;
        if ((rc = launch_bp_all(e, x, r, nullptr, 1.f, beta, 1))) return rc;
    }
    return TOMO_OK;
}

// ctvlib::SIRT(beta) with cimminos_method() active: x += A^T M (b - A x) * beta/Nrow, M = diag(|A_i|^2) (sic: the
// reference multiplies by the row norms, quirk Q10), then positivity   ctvlib.cpp:212-216, 245-251
int tomo_sirt_cimmino(tomo_engine *e, int vol, float beta, int niter)
{
    NEED(e);
    float *x, *r; int rc;
    if ((rc = get_vol(e, vol, &x)) || (rc = get_sino(e, &e->sino[TOMO_SINO_R], &r))) return rc;
    const float *rs = e->d_rowsum;
    for (int it = 0; it < niter; ++it) {
        e->d_rowsum = e->d_rowinner;                     // the FP epilogue reads its per-row factor from this argument
        rc = launch_fp_all<FP_RESID_MUL>(e, x, e->sino[TOMO_SINO_B], r);
        e->d_rowsum = const_cast<float *>(rs);
        if (rc) return rc;
        if ((rc = launch_bp_all(e, x, r, nullptr, 1.f, beta / (float)e->nrows, 1))) return rc;
    }
    return TOMO_OK;
}

int tomo_sirt(tomo_engine *e, int vol, int niter) { return tomo_sirt_data(e, vol, TOMO_SINO_B, niter); }

int tomo_sirt_data(tomo_engine *e, int vol, int sino_b, int niter)
{
    NEED(e);
    float *x, *r, *b; int rc;
    const float *have = (niter > 0 && sino_b != TOMO_SINO_G) ? projection_in_hand(e, vol) : nullptr;   // A * this volume, as it stands
    const bool reuse = have != nullptr;
    if (reuse) { if ((rc = order_after_async(e))) return rc; }                            // (an evaluation on the second stream made it)
    if ((rc = get_vol(e, vol, &x)) || (rc = get_sino(e, &e->sino[TOMO_SINO_R], &r)) || (rc = sino_slot(e, sino_b, &b))) return rc;
    for (int it = 0; it < niter; ++it) {
        if (it == 0 && reuse) { if ((rc = launch_sino_resid<FP_RESID_NORM>(e, b, have, r))) return rc; }
        else if ((rc = launch_fp_all<FP_RESID_NORM>(e, x, b, r))) return rc;
        if ((rc = launch_bp_all(e, x, r, e->d_colsum_all, 1.f, 1.f, 1))) return rc;
    }
    return TOMO_OK;
}

// ---- the SART sweep as one launch of the volume-resident kernel (sart_resident.hip.h) ----------------------------------------------
// Every workgroup of the launch must be on the chip at once (they wait for one another's ray sums), and a workgroup fills a CU:
// two such launches side by side -- two engines on one device, on two streams -- could each get part of the chip and wait for the
// rest until their spins run out.  So the resident launches of one device form a chain: each waits for the event the one before
// it (any engine, any stream) recorded.  Other kernels may overlap freely: they finish by themselves.
static std::mutex g_rs_mu;
static hipEvent_t g_rs_last[64] = {};

// failed: the 64-slice chunk ranges {c0, nc} the launch did not store (empty = the whole sweep is in x); the call returns after the
// launch has finished -- one host wait per sweep (~10 us against a sweep of milliseconds) is what knowing costs
static int launch_sart_resident(tomo_engine *e, float *x, float beta, int64_t steps, const std::function<int(int64_t)> &angle_at, float *track,
                                std::vector<std::pair<int, int>> &failed)
{
    failed.clear();
    if (steps > (int64_t)1 << 24) return fail(TOMO_ERR_ARG, "too many SART steps in one call");
    const int c0 = e->sub_nc ? e->sub_c0 : 0, nc = e->sub_nc ? e->sub_nc : e->sxc / 64;
    const int groups = std::max(1, std::min(e->rs_groups, nc)), rounds = (nc + groups - 1) / groups;
    {   // the angle of every step, on the device (an unchanged sequence stays where it is)
        std::vector<int> seq((size_t)steps);
        for (int64_t k = 0; k < steps; ++k) seq[(size_t)k] = angle_at(k);
        if (seq != e->rs_angs_host) {
            HIPCHK(hipStreamSynchronize(e->stream));          // (a sweep in flight may still read the old sequence; rare: the first sweep, a new order)
            if ((size_t)steps > e->rs_angs_cap) {
                if (e->d_rs_angs) { HIPCHK(hipFree(e->d_rs_angs)); e->d_rs_angs = nullptr; e->rs_angs_cap = 0; }
                HIPCHK(hipMalloc((void **)&e->d_rs_angs, (size_t)steps * sizeof(int)));
                e->rs_angs_cap = (size_t)steps;
            }
            HIPCHK(hipMemcpy(e->d_rs_angs, seq.data(), (size_t)steps * sizeof(int), hipMemcpyHostToDevice));
            e->rs_angs_host.swap(seq);
        }
    }
    const uint64_t need = (uint64_t)rounds * (uint64_t)steps;
    if ((uint64_t)e->rs_epoch + need + 16 > 0xFFFFFFFFull) {     // the tags wrap: back to the state after creation (tag 0 = never written)
        HIPCHK(hipMemsetAsync(e->rs_pb, 0, e->rs_pb_bytes, e->stream));
        HIPCHK(hipMemsetAsync(e->rs_rb, 0, e->rs_rb_bytes, e->stream));
        e->rs_epoch = 0;
    }
    // commit words: all at rs_commit_base (every launch covers every chunk and adds the number of tiles to each word when it commits);
    // cleared after a launch that did not commit, and before the count could reach the poison bit
    if (c0 != 0 || nc != e->sxc / 64) return fail(TOMO_ERR_STATE, "the resident sweep covers the whole slab");
    if (e->rs_commit_dirty || e->rs_commit_base > 0x7F000000u) {
        HIPCHK(hipMemsetAsync(e->d_rs_commit, 0, (size_t)(e->sxc / 64) * sizeof(unsigned), e->stream));
        e->rs_commit_base = 0; e->rs_commit_dirty = false;
    }
    if (++e->rs_seq >= 0x7FFFFFF0u) { std::memset(e->rs_done, 0, (size_t)(e->sxc / 64) * sizeof(int)); e->rs_seq = 1; }
    RsArgs A{};
    A.x = x; A.b = e->cur_b; A.rowsum = e->d_rowsum; A.hdr = e->d_rs_hdr; A.cell = e->d_rs_cell; A.ts = e->d_rs_ts; A.rl = e->d_rs_rl;
    A.pb = e->rs_pb; A.rb = e->rs_rb; A.angs = e->d_rs_angs; A.track = track; A.part = e->d_part; A.abort_word = e->d_rs_abort; A.abort_host = e->rs_abort;
    A.commit = e->d_rs_commit; A.done_host = e->rs_done; A.seq = e->rs_seq; A.commit_base = e->rs_commit_base; A.test_fail = e->rs_test_fail;
    A.n = e->n; A.sx = e->sx; A.np = e->np; A.ntiles = e->rs_ntiles; A.tiles = e->rs_tiles; A.rpt = e->rs_rpt; A.steps = (int)steps; A.chunk0 = c0; A.nchunk = nc;
    A.epoch0 = e->rs_epoch; A.spin_limit = e->rs_spin_limit; A.beta = beta; A.prof = nullptr;
    e->rs_epoch += (uint32_t)need;
    if (e->device < 0 || e->device >= 64) return fail(TOMO_ERR_ARG, "device index");
    {
        std::lock_guard<std::mutex> lk(g_rs_mu);
        hipEvent_t &last = g_rs_last[e->device];
        if (!last) HIPCHK(hipEventCreateWithFlags(&last, hipEventDisableTiming));
        else HIPCHK(hipStreamWaitEvent(e->stream, last, 0));
        {
            ProfScope ps(e, TOMO_K_SART_RESIDENT);
            hipLaunchKernelGGL(k_sart_resident, dim3((unsigned)(e->rs_ntiles * groups)), dim3(RS_THREADS), 0, e->stream, A);
            LAUNCHCHK();
        }
        HIPCHK(hipEventRecord(last, e->stream));
    }
    // the verdict: which chunks carry this launch's sequence number
    HIPCHK(hipStreamSynchronize(e->stream));
    int nfailed = 0;
    for (int c = c0; c < c0 + nc; ++c) {
        if (e->rs_done[c] == (int)e->rs_seq) continue;
        ++nfailed;
        if (!failed.empty() && failed.back().first + failed.back().second == c) ++failed.back().second;
        else failed.emplace_back(c, 1);
    }
    if (nfailed) {
        e->rs_commit_dirty = true;
        e->rs_last_code = *e->rs_abort;
        *e->rs_abort = 0;
        HIPCHK(hipMemsetAsync(e->d_rs_abort, 0, sizeof(int), e->stream));
        ++e->rs_fallbacks;
        e->rs_fallback_chunks += nfailed;
        e->rs_backoff = std::min(64, std::max(1, 2 * e->rs_backoff));
        e->rs_skip = e->rs_backoff;
    } else {
        e->rs_commit_base += (unsigned)e->rs_ntiles;
        e->rs_backoff = 0;
    }
    return TOMO_OK;
}

int tomo_sart(tomo_engine *e, int vol, float beta, int niter, const int32_t *order)
{
    return tomo_sart_data(e, vol, TOMO_SINO_B, beta, niter, order);
}

// track_vol >= 0: the last back-projection of the sweep also leaves ||x_new - track||^2 in scalar `slot` and copies x_new
// into track_vol (tomo_sart_tracked)
static int sart_impl(tomo_engine *e, int vol, int sino_b, float beta, int niter, const int32_t *order, int track_vol, int slot)
{
    NEED(e);
    float *x, *r, *track = nullptr; int rc;
    if ((rc = get_vol(e, vol, &x)) || (rc = get_sino(e, &e->sino[TOMO_SINO_R], &r)) || (rc = sino_slot(e, sino_b, &e->cur_b))) return rc;
    if (track_vol >= 0) {
        if (track_vol == vol) return fail(TOMO_ERR_ARG, "the tracked volume must differ from the swept one");
        if (slot < 0 || slot >= TOMO_S_COUNT) return fail(TOMO_ERR_ARG, "bad scalar slot");
        if ((rc = get_vol(e, track_vol, &track))) return rc;
        if ((int64_t)niter * e->np <= 0) {   // nothing is swept: the plain pair of passes
            if ((rc = tomo_diff_norm_sq(e, vol, track_vol, slot))) return rc;
            return tomo_copy_volume(e, track_vol, vol);
        }
        if ((rc = reduce_begin(e))) return rc;
    }
    auto finish = [&]() -> int { return track ? reduce_end(e, slot) : TOMO_OK; };
    if (order) {
        std::vector<char> seen(e->np, 0);
        for (int q = 0; q < e->np; ++q) {
            if (order[q] < 0 || order[q] >= e->np || seen[order[q]]) return fail(TOMO_ERR_ARG, "SART order is not a permutation of the angles");
            seen[order[q]] = 1;
        }
    }
    const int64_t steps = (int64_t)niter * e->np;
    auto angle_at = [&](int64_t k) { int q = (int)(k % e->np); return order ? order[q] : q; };
    if (!e->sart_fused) {
        // reference structure: one forward projection + one back-projection update per angle
        for (int64_t k = 0; k < steps; ++k) {
            int i = angle_at(k);
            {
                ProfScope ps(e, TOMO_K_FP_ANGLE);
                if ((rc = launch_fp<FP_RESID_NORM>(e, x, i * e->n, e->n, e->cur_b, r))) return rc;
            }
            if ((rc = launch_bp_angle(e, x, i, r + (size_t)i * e->n * e->sx, beta, k == steps - 1 ? track : nullptr))) return rc;
        }
        return finish();
    }
    // fused chain: FP(a0) ; [BP(a_k) + FP(a_k+1)] for every consecutive pair ; BP(a_last)
    if (steps <= 0) return TOMO_OK;
    int form = select_forms(e).sart;
    if (form == TOMO_FORM_SART_RESIDENT && e->rs_skip > 0 && e->sart_resident != 1) {   // sitting out after a sweep that could not finish
        --e->rs_skip;
        form = TOMO_FORM_SART_TILE;
    }
    // the streamed tile form, in place: over the whole slab (as one chain or several), or over the chunk ranges a resident launch left
    auto sweep_tiles = [&](const std::vector<std::pair<int, int>> *ranges) -> int {
        // cooperative chain (k_sart_tile COOP): needs consecutive angles to differ (np >= 2) and whole 64-slice chunks
        const bool coop = e->sart_coop && e->np >= 2 && steps >= 2 && !ranges;
        int rc2;
        if ((rc2 = sart_tile_prepare(e, coop))) return rc2;
        const uint32_t epoch0 = e->st_epoch + 1;             // link k publishes with epoch0 + k (both sub-slab chains alike)
        if (coop) e->st_epoch += (uint32_t)(steps + 1);
        // link k of the chain: 0 = FP(a0); 1..steps-1 = BP(a_k-1) + FP(a_k); steps = BP(a_last)
        auto link = [&](int64_t k, const Sub &sb) -> int {
            int rc3;
            if (coop) {
                float *pk = (k & 1) ? e->st_partial2 : e->st_partial, *pk1 = (k & 1) ? e->st_partial : e->st_partial2;   // P[k&1], P[(k-1)&1]
                if (k == 0) return launch_sart_tile<false>(e, sb, x, 0, angle_at(0), r, beta, pk, false);
                if (k == steps) {
                    int last = angle_at(steps - 1);
                    if ((rc3 = launch_resid_finish_tile(e, sb, pk1, last, r))) return rc3;
                    return launch_bp_angle(e, sb, x, last, r + (size_t)last * e->n * e->sx, beta, track);
                }
                return launch_sart_coop(e, sb, x, angle_at(k - 1), angle_at(k), r, beta, pk1, pk, epoch0 + (uint32_t)k, k);
            }
            if (k == 0) return launch_sart_tile<false>(e, sb, x, 0, angle_at(0), r, beta);
            if (k == steps) { int last = angle_at(steps - 1); return launch_bp_angle(e, sb, x, last, r + (size_t)last * e->n * e->sx, beta, track); }
            int prev = angle_at(k - 1), next = angle_at(k);
            if (prev == next) {
                if ((rc3 = launch_bp_angle(e, sb, x, prev, r + (size_t)prev * e->n * e->sx, beta))) return rc3;
                return launch_sart_tile<false>(e, sb, x, 0, next, r, beta);
            }
            return launch_sart_tile<true>(e, sb, x, prev, next, r, beta, nullptr, true, k);
        };
        auto chain = [&](const Sub &sb) -> int {
            for (int64_t k = 0; k <= steps; ++k) { int rc3 = link(k, sb); if (rc3) return rc3; }
            return TOMO_OK;
        };
        if (!ranges) return run_chains(e, chain);
        for (const auto &rg : *ranges) {     // one after the other on the engine's stream (they share the partial-sum buffer)
            const int c0 = rg.first, nc = rg.second;
            const int vec = (c0 % 4 == 0 && nc % 4 == 0) ? 4 : (c0 % 2 == 0 && nc % 2 == 0) ? 2 : 1;
            if ((rc2 = chain(Sub{e->stream, c0, nc, vec}))) return rc2;
        }
        return TOMO_OK;
    };
    if (form == TOMO_FORM_SART_RESIDENT) {      // the volume-resident sweep: one launch, the slab read and written once
        std::vector<std::pair<int, int>> failed;
        if ((rc = launch_sart_resident(e, x, beta, steps, angle_at, track, failed))) return rc;
        // chunks whose workgroups could not all finish (the device was shared) were not stored: the streamed chain sweeps them
        if (!failed.empty() && (rc = sweep_tiles(&failed))) return rc;
        return finish();
    }
    if (e->sart_resident == 1 && !e->rs_ok) return fail(TOMO_ERR_STATE, "\"sart_resident\" = 1, but this engine has no tables of the resident sweep (N not a multiple of 8, more 32 x 32 tiles than CUs, or a matrix whose ray windows do not fit)");
    if (form == TOMO_FORM_SART_TILE) {   // tile form, in place
        if ((rc = sweep_tiles(nullptr))) return rc;
        return finish();
    }
    float *alt;
    if ((rc = get_scratch(e, &e->sart_alt, &alt))) return rc;
    float *cur = x;
    if ((rc = launch_sart_seg<false>(e, cur, nullptr, 0, angle_at(0), r, beta))) return rc;
    for (int64_t k = 1; k < steps; ++k) {
        int prev = angle_at(k - 1), next = angle_at(k);
        if (prev == next) {   // single-angle geometry: the residual rows read and written would be the same
            if ((rc = launch_bp_angle(e, cur, prev, r + (size_t)prev * e->n * e->sx, beta))) return rc;
            if ((rc = launch_fp<FP_RESID_NORM>(e, cur, next * e->n, e->n, e->cur_b, r))) return rc;
            continue;
        }
        if ((rc = launch_sart_seg<true>(e, cur, alt, prev, next, r, beta))) return rc;
        std::swap(cur, alt);
    }
    int last = angle_at(steps - 1);
    if ((rc = launch_bp_angle(e, cur, last, r + (size_t)last * e->n * e->sx, beta, track))) return rc;
    if (cur != x) { e->vol[vol] = cur; e->sart_alt = x; }   // the swept volume now lives in the partner buffer
    return finish();
}

int tomo_sart_data(tomo_engine *e, int vol, int sino_b, float beta, int niter, const int32_t *order)
{
    return sart_impl(e, vol, sino_b, beta, niter, order, -1, 0);
}

int tomo_sart_tracked(tomo_engine *e, int vol, int sino_b, float beta, int niter, const int32_t *order, int track_vol, int slot)
{
    return sart_impl(e, vol, sino_b, beta, niter, order, track_vol, slot);
}

int tomo_art(tomo_engine *e, float beta) { return tomo_art_order(e, beta, nullptr); }

int tomo_art_order(tomo_engine *e, float beta, const int32_t *order_host)
{
    NEED(e);
    float *x;
    { int rc_ = get_vol(e, TOMO_VOL_RECON, &x); if (rc_) return rc_; }
    int32_t *d_order = nullptr;
    if (order_host) {
        std::vector<char> seen(e->nrows, 0);
        for (int64_t q = 0; q < e->nrows; ++q) {
            if (order_host[q] < 0 || order_host[q] >= e->nrows || seen[order_host[q]]) return fail(TOMO_ERR_ARG, "row order is not a permutation");
            seen[order_host[q]] = 1;
        }
        int rc = ensure_stage(e, e->nrows * sizeof(int32_t)); if (rc) return rc;
        HIPCHK(hipMemcpyAsync(e->stage, order_host, e->nrows * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
        d_order = (int32_t *)e->stage;
    }
    if (!order_host && e->art_chain && e->art_chain_ok) {
        // natural order: one angle = forward projection + recurrence along the rays + back-projection (k_art_chain)
        float *d, *a; int rc;
        if ((rc = get_sino(e, &e->sino[TOMO_SINO_G], &d)) || (rc = get_sino(e, &e->sino[TOMO_SINO_R], &a))) return rc;
        const float *b = e->sino[TOMO_SINO_B];
        int nchunk = e->sxc / (64 * e->vec);
        int ngroups = (int)((e->npix + BP_PPW - 1) / BP_PPW);
        dim3 bgrid((unsigned)(((int64_t)ngroups * nchunk + 3) / 4));
        if (e->art_tile && e->sart_tile && e->st_ok && e->np >= 1) {
            // the same fused steps as the SART sweep: FP(a0); [BP_art(a_k-1) + FP(a_k)] ...; BP_art(a_last), the residual rows of
            // each angle formed by k_art_chain from the tile step's row sums.  Per angle 228 instead of 323 us at 512^3.
            if ((rc = sart_tile_prepare(e, false))) return rc;
            e->cur_b = const_cast<float *>(b);
            const int np = e->np, strm = slab_streams(e) ? 1 : 0;
            auto chain = [&](const Sub &sb) -> int {        // one sub-slab's (or the whole slab's) chain of launches
                const int c64 = sb.nc ? sb.c0 : 0, nc64 = sb.nc ? sb.nc : e->sxc / 64;
                for (int i = 0; i < np; ++i) {
                    int rc2 = i == 0 ? launch_sart_tile<false, true>(e, sb, x, 0, 0, a, beta, nullptr, true, -1, d)
                                     : launch_sart_tile<true, true>(e, sb, x, i - 1, i, a, beta, nullptr, true, i, d);
                    if (rc2) return rc2;
                    hipLaunchKernelGGL(k_art_chain, dim3((unsigned)nc64), dim3(64 * ART_CW), 0, sb.stream, d, b, e->d_rowinner, e->d_rowcross, a, beta, i * e->n, e->n, e->sx, c64);
                    LAUNCHCHK();
                }
                const int last = np - 1, vec = sub_vec(e, sb);
                const CellD *cell = e->d_cell + (size_t)last * e->npix;
                const float *ai = a + (size_t)last * e->n * e->sx;
                const int nch = nc64 / vec, ch0 = c64 / vec;
                dim3 grid((unsigned)(((int64_t)ngroups * nch + 3) / 4));
                switch (vec) {
                case 4: hipLaunchKernelGGL((k_bp_art<4, BP_PPW>), grid, dim3(256), 0, sb.stream, x, cell, ai, beta, (int)e->npix, e->sx, ngroups, nch, strm, ch0); break;
                case 2: hipLaunchKernelGGL((k_bp_art<2, BP_PPW>), grid, dim3(256), 0, sb.stream, x, cell, ai, beta, (int)e->npix, e->sx, ngroups, nch, strm, ch0); break;
                default: hipLaunchKernelGGL((k_bp_art<1, BP_PPW>), grid, dim3(256), 0, sb.stream, x, cell, ai, beta, (int)e->npix, e->sx, ngroups, nch, strm, ch0); break;
                }
                LAUNCHCHK();
                return TOMO_OK;
            };
            if ((rc = run_chains(e, chain))) return rc;
            return tomo_positivity(e, TOMO_VOL_RECON);
        }
        for (int i = 0; i < e->np; ++i) {
            if ((rc = launch_fp<FP_STORE>(e, x, i * e->n, e->n, nullptr, d))) return rc;
            hipLaunchKernelGGL(k_art_chain, dim3((unsigned)(e->sx / 64)), dim3(64 * ART_CW), 0, e->stream, d, b, e->d_rowinner, e->d_rowcross, a, beta, i * e->n, e->n, e->sx, 0);
            LAUNCHCHK();
            const CellD *cell = e->d_cell + (size_t)i * e->npix;
            const float *ai = a + (size_t)i * e->n * e->sx;
            switch (e->vec) {
            case 4: hipLaunchKernelGGL((k_bp_art<4, BP_PPW>), bgrid, dim3(256), 0, e->stream, x, cell, ai, beta, (int)e->npix, e->sx, ngroups, nchunk, slab_streams(e) ? 1 : 0, 0); break;
            case 2: hipLaunchKernelGGL((k_bp_art<2, BP_PPW>), bgrid, dim3(256), 0, e->stream, x, cell, ai, beta, (int)e->npix, e->sx, ngroups, nchunk, slab_streams(e) ? 1 : 0, 0); break;
            default: hipLaunchKernelGGL((k_bp_art<1, BP_PPW>), bgrid, dim3(256), 0, e->stream, x, cell, ai, beta, (int)e->npix, e->sx, ngroups, nchunk, slab_streams(e) ? 1 : 0, 0); break;
            }
            LAUNCHCHK();
        }
        return tomo_positivity(e, TOMO_VOL_RECON);
    }
    hipLaunchKernelGGL(k_art, dim3(e->sxc / 64), dim3(64 * ART_WAVES), 0, e->stream, x, e->d_rptr, e->d_rent, e->sino[TOMO_SINO_B], e->d_rowinner, beta, (int)e->nrows, e->sx, d_order);
    LAUNCHCHK();
    int rc = tomo_positivity(e, TOMO_VOL_RECON);
    if (!rc && order_host) HIPCHK(hipStreamSynchronize(e->stream));   // the staging buffer holds the order until the sweep is done
    return rc;
}

int tomo_poisson_ml(tomo_engine *e, float lambda)
{
    int rc;
    if ((rc = tomo_poisson_residual(e, TOMO_VOL_RECON, TOMO_SINO_B, TOMO_SINO_R))) return rc;
    float *x;
    if ((rc = get_vol(e, TOMO_VOL_RECON, &x))) return rc;
    return launch_bp_all(e, x, e->sino[TOMO_SINO_R], nullptr, 1.f, -(lambda / e->lipschitz), 1);
}

int tomo_poisson_residual(tomo_engine *e, int vol, int sino_b, int sino_out)
{
    NEED(e);
    float *x, *b, *r; int rc;
    if ((rc = get_vol(e, vol, &x)) || (rc = sino_slot(e, sino_b, &b)) || (rc = sino_slot(e, sino_out, &r))) return rc;
    if ((rc = reduce_begin(e))) return rc;
    if ((rc = launch_fp_all<FP_POISSON>(e, x, b, r))) return rc;
    return reduce_end(e, TOMO_S_COST);
}

// ---- CGLS (TomoGPU.cgls; ASTRA CCudaCglsAlgorithm in the reference: tomoengine.cpp:207-229) -------------------------
// Standard CGLS on min ||A x - b||, restarted from the current volume at every call exactly like the reference
// (algo_cgls->initialize per call), independently per slice (alpha, beta are per-slice scalars), positivity at the end.
static int slice_sumsq(tomo_engine *e, const float *v, int64_t m, double *sums)
{
    const int cols = e->sx / 4, per = cols >= 256 ? 256 : (cols >= 128 ? 128 : 64);
    const int nby = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (m + 63) / 64));   // workgroups along the rows
    const int rpb = (int)((m + nby - 1) / nby);
    if (!e->cg_part) { int rc = dev_alloc((void **)&e->cg_part, (size_t)1024 * e->sx * sizeof(double), false, e->stream); if (rc) return rc; }
    dim3 grid((unsigned)((cols + per - 1) / per), (unsigned)nby);
    hipLaunchKernelGGL(k_slice_sumsq, grid, dim3(256), 0, e->stream, v, e->cg_part, m, e->sx, rpb);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_slice_sumsq_finish, dim3((e->sx + 255) / 256), dim3(256), 0, e->stream, (const double *)e->cg_part, sums, nby, e->sx);
    LAUNCHCHK();
    return TOMO_OK;
}

// grid of the per-slice axpy kernels: its stride (blocks * 256 float4) must be a multiple of sx/4 so that a thread keeps its
// slice group; sx is a multiple of 64, so sx/4 divides 256 * (sx/4) / gcd -- simply take a block count that is a multiple of sx/64
static unsigned slice_grid(const tomo_engine *e, int64_t n4)
{
    const int64_t unit = std::max(1, e->sx / 64);          // blocks per 4 rows... 256 float4 = 1024 floats = 1024/sx rows
    int64_t b = std::min<int64_t>((n4 + 255) / 256, 8192);
    b = std::max<int64_t>(unit, (b / unit) * unit);
    return (unsigned)b;
}

int tomo_cgls(tomo_engine *e, int vol, int niter)
{
    NEED(e);
    float *x, *r, *b, *w; int rc;
    const bool reuse_g = g_is_projection_of(e, vol);        // the restart's A x is already in G (a data_distance of this volume)
    if (reuse_g) { if ((rc = order_after_async(e))) return rc; }
    if ((rc = get_vol(e, vol, &x)) || (rc = get_sino(e, &e->sino[TOMO_SINO_R], &r)) || (rc = sino_slot(e, TOMO_SINO_B, &b))) return rc;
    if ((rc = get_scratch(e, &e->cg_p, &w)) || (rc = get_scratch(e, &e->cg_z, &w))) return rc;
    if ((rc = get_sino(e, &e->cg_w, &w))) return rc;
    if (!e->cg_sums) { if ((rc = dev_alloc((void **)&e->cg_sums, 2 * e->sx * sizeof(double), true, e->stream))) return rc; }
    if (!e->cg_coef) { if ((rc = dev_alloc((void **)&e->cg_coef, e->sx * sizeof(float), true, e->stream))) return rc; }
    double *gam = e->cg_sums, *tmp = e->cg_sums + e->sx;
    const int64_t nv = (int64_t)e->vol_elems(), ns = (int64_t)e->sino_elems();
    auto ratio = [&](const double *num, const double *den) {
        hipLaunchKernelGGL(k_slice_ratio, dim3((e->sx + 255) / 256), dim3(256), 0, e->stream, num, den, e->cg_coef, e->sx);
    };
    // r = b - A x ; z = A^T r ; p = z ; gamma = |z|^2
    if (reuse_g) { if ((rc = launch_sino_resid<FP_RESID>(e, b, e->sino[TOMO_SINO_G], r))) return rc; }
    else if ((rc = launch_fp_all<FP_RESID>(e, x, b, r))) return rc;
    if ((rc = launch_bp_all(e, e->cg_z, r, nullptr, 0.f, 1.f, 0))) return rc;
    HIPCHK(hipMemcpyAsync(e->cg_p, e->cg_z, nv * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    if ((rc = slice_sumsq(e, e->cg_z, e->npix, gam))) return rc;
    for (int it = 0; it < niter; ++it) {
        // w = A p ; alpha = gamma / |w|^2 ; x += alpha p ; r -= alpha w
        if ((rc = launch_fp_all<FP_STORE>(e, e->cg_p, nullptr, e->cg_w))) return rc;
        if ((rc = slice_sumsq(e, e->cg_w, e->nrows, tmp))) return rc;
        ratio(gam, tmp);
        hipLaunchKernelGGL(k_slice_axpy, dim3(slice_grid(e, nv / 4)), dim3(256), 0, e->stream, (f4 *)x, (const f4 *)e->cg_p, (const f4 *)e->cg_coef, 1.f, nv / 4, e->sx / 4);
        hipLaunchKernelGGL(k_slice_axpy, dim3(slice_grid(e, ns / 4)), dim3(256), 0, e->stream, (f4 *)r, (const f4 *)e->cg_w, (const f4 *)e->cg_coef, -1.f, ns / 4, e->sx / 4);
        // z = A^T r ; beta = |z|^2 / gamma ; gamma = |z|^2 ; p = z + beta p
        if ((rc = launch_bp_all(e, e->cg_z, r, nullptr, 0.f, 1.f, 0))) return rc;
        if ((rc = slice_sumsq(e, e->cg_z, e->npix, tmp))) return rc;
        ratio(tmp, gam);
        HIPCHK(hipMemcpyAsync(gam, tmp, e->sx * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        hipLaunchKernelGGL(k_slice_xpay, dim3(slice_grid(e, nv / 4)), dim3(256), 0, e->stream, (f4 *)e->cg_p, (const f4 *)e->cg_z, (const f4 *)e->cg_coef, nv / 4, e->sx / 4);
        LAUNCHCHK();
    }
    return tomo_positivity(e, vol);
}

// ---- WBP / FBP (TomoGPU.wbp; ASTRA CCudaFilteredBackProjectionAlgorithm: tomoengine.cpp:317-347) ----------------------
// recon = scale * A^T (h * b): h = real-space filter taps h[0..N-1] (symmetric), built by the host for the named filter.
int tomo_fbp(tomo_engine *e, const float *taps_host, float scale, int apply_positivity)
{
    NEED(e);
    if (!taps_host) return fail(TOMO_ERR_ARG, "null filter");
    float *x, *b, *g; int rc;
    if ((rc = get_vol(e, TOMO_VOL_RECON, &x)) || (rc = sino_slot(e, TOMO_SINO_B, &b)) || (rc = get_sino(e, &e->sino[TOMO_SINO_R], &g))) return rc;
    if (!e->fbp_h) { if ((rc = dev_alloc((void **)&e->fbp_h, e->n * sizeof(float), false, e->stream))) return rc; }
    HIPCHK(hipMemcpyAsync(e->fbp_h, taps_host, e->n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    int nchunk = e->sxc / (64 * e->vec);
    int64_t waves = e->nrows * nchunk;
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    switch (e->vec) {
    case 4: hipLaunchKernelGGL((k_filter_rows<4>), grid, block, 0, e->stream, b, g, e->fbp_h, e->n, (int)e->nrows, e->sx, nchunk); break;
    case 2: hipLaunchKernelGGL((k_filter_rows<2>), grid, block, 0, e->stream, b, g, e->fbp_h, e->n, (int)e->nrows, e->sx, nchunk); break;
    default: hipLaunchKernelGGL((k_filter_rows<1>), grid, block, 0, e->stream, b, g, e->fbp_h, e->n, (int)e->nrows, e->sx, nchunk); break;
    }
    LAUNCHCHK();
    return launch_bp_all(e, x, g, nullptr, 0.f, scale, apply_positivity ? 1 : 0);
}

int tomo_scale_volume(tomo_engine *e, int vol, float factor)
{
    NEED(e);
    { int rc_ = order_after_async(e); if (rc_) return rc_; }
    float *x; int rc; if ((rc = get_vol(e, vol, &x))) return rc;
    int64_t n4 = e->vol_elems() / 4;
    hipLaunchKernelGGL(k_scale, dim3(grid_1d(n4)), dim3(256), 0, e->stream, (f4 *)x, factor, n4);
    LAUNCHCHK();
    return TOMO_OK;
}

int tomo_sino_diff_norm_sq(tomo_engine *e, int a, int b, int slot)
{
    NEED(e);
    if (slot < 0 || slot >= TOMO_S_COUNT) return fail(TOMO_ERR_ARG, "bad scalar slot");
    float *pa, *pb; int rc;
    if ((rc = sino_slot(e, a, &pa)) || (rc = sino_slot(e, b, &pb))) return rc;
    if ((rc = reduce_begin(e))) return rc;
    int64_t n4 = e->sino_elems() / 4;
    hipLaunchKernelGGL(k_sqdiff, dim3(grid_1d(n4)), dim3(256), 0, e->stream, (const f4 *)pa, (const f4 *)pb, e->d_part, n4);
    LAUNCHCHK();
    return reduce_end(e, slot);
}

// per-projection maximum over (slices, rays): multimodal::rescale_projections (multimodal.cpp:323-327)
int tomo_sino_proj_max(tomo_engine *e, int sino, float *out_host)
{
    NEED(e);
    float *g; int rc; if ((rc = sino_slot(e, sino, &g))) return rc;
    if (!out_host) return fail(TOMO_ERR_ARG, "null output");
    if ((rc = ensure_stage(e, e->np * sizeof(float)))) return rc;
    hipLaunchKernelGGL(k_proj_max, dim3(e->np), dim3(256), 0, e->stream, g, e->stage, e->n, e->nx, e->sx);
    LAUNCHCHK();
    HIPCHK(hipMemcpyAsync(out_host, e->stage, e->np * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TOMO_OK;
}

int tomo_sino_proj_scale(tomo_engine *e, int sino, const float *div_host, const float *mul_host)
{
    NEED(e);
    float *g; int rc; if ((rc = sino_slot(e, sino, &g))) return rc;
    if (!div_host || !mul_host) return fail(TOMO_ERR_ARG, "null factors");
    if ((rc = ensure_stage(e, 2 * e->np * sizeof(float)))) return rc;
    HIPCHK(hipMemcpyAsync(e->stage, div_host, e->np * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->stage + e->np, mul_host, e->np * sizeof(float), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_proj_scale, dim3(e->np), dim3(256), 0, e->stream, g, e->stage, e->n, e->sx);
    LAUNCHCHK();
    HIPCHK(hipStreamSynchronize(e->stream));
    return TOMO_OK;
}

// copy a volume between two engines of the same slab shape on one device (used when the tilt geometry is rebuilt:
// tomoengine::update_projection_angles, tomoengine.cpp:128-149, keeps the reconstruction)
int tomo_copy_volume_from(tomo_engine *dst, int dst_vol, tomo_engine *src, int src_vol)
{
    if (!dst || !src) return fail(TOMO_ERR_ARG, "null engine");
    if (dst->nx != src->nx || dst->n != src->n || dst->sx != src->sx || dst->device != src->device) return fail(TOMO_ERR_ARG, "engines differ in slab shape or device");
    HIPCHK(hipSetDevice(dst->device));
    float *d, *s; int rc;
    if ((rc = get_vol(dst, dst_vol, &d)) || (rc = get_vol(src, src_vol, &s))) return rc;
    HIPCHK(hipStreamSynchronize(src->stream));
    HIPCHK(hipMemcpyAsync(d, s, dst->vol_elems() * sizeof(float), hipMemcpyDeviceToDevice, dst->stream));
    HIPCHK(hipStreamSynchronize(dst->stream));
    return TOMO_OK;
}

int tomo_get_stream(tomo_engine *e, void **out) { if (!e || !out) return fail(TOMO_ERR_ARG, "null"); *out = (void *)e->stream; return TOMO_OK; }

// ---- multimodal (ChemicalTomo) element-wise steps: two engines of equal slab size on one device/stream ----
static int mm_check(tomo_engine *a, tomo_engine *b, int nel)
{
    if (!a || !b) return fail(TOMO_ERR_ARG, "null engine");
    if (a->nx != b->nx || a->n != b->n || a->sx != b->sx || a->device != b->device) return fail(TOMO_ERR_ARG, "engines differ in slab shape or device");
    if (a->stream != b->stream) return fail(TOMO_ERR_STATE, "engines must share one stream (tomo_set_stream)");
    if (nel < 1 || nel > MM_MAX_EL) return fail(TOMO_ERR_ARG, "element count out of range");
    return TOMO_OK;
}

// model = Sigma * x^gamma = sum_e w_e x_e^gamma      multimodal.cpp:425-427 (fuse), :459-460
int tomo_mm_model(tomo_engine *ce, const int32_t *xvols, int nel, const float *w, float gamma, tomo_engine *he, int model_vol)
{
    int rc = mm_check(ce, he, nel); if (rc) return rc;
    HIPCHK(hipSetDevice(ce->device));
    MMArgs a{};
    a.nel = nel; a.gamma = gamma;
    for (int i = 0; i < nel; ++i) { float *p; if ((rc = get_vol(ce, xvols[i], &p))) return rc; a.x[i] = p; a.w[i] = w[i]; }
    float *m; if ((rc = get_vol(he, model_vol, &m))) return rc;
    int64_t n4 = ce->vol_elems() / 4;
    hipLaunchKernelGGL(k_mm_model, dim3(grid_1d(n4)), dim3(256), 0, ce->stream, a, (f4 *)m, n4);
    LAUNCHCHK();
    return TOMO_OK;
}

// x_e <- max(0, x_e - (lamC_over_L * uC_e - lamH * gamma x_e^(gamma-1) w_e (upd - model)))   multimodal.cpp:435-438,471
int tomo_mm_update(tomo_engine *ce, const int32_t *xvols, const int32_t *uvols, int nel, const float *w, float gamma,
                   float lamC_over_L, float lamH, tomo_engine *he, int upd_vol, int model_vol)
{
    int rc = mm_check(ce, he, nel); if (rc) return rc;
    HIPCHK(hipSetDevice(ce->device));
    MMArgs a{};
    a.nel = nel; a.gamma = gamma;
    for (int i = 0; i < nel; ++i) {
        float *p, *u;
        if ((rc = get_vol(ce, xvols[i], &p)) || (rc = get_vol(ce, uvols[i], &u))) return rc;
        a.x[i] = p; a.u[i] = u; a.w[i] = w[i];
    }
    float *upd = nullptr, *m = nullptr;
    if (lamH != 0.f) { if ((rc = get_vol(he, upd_vol, &upd)) || (rc = get_vol(he, model_vol, &m))) return rc; }
    int64_t n4 = ce->vol_elems() / 4;
    hipLaunchKernelGGL(k_mm_update, dim3(grid_1d(n4)), dim3(256), 0, ce->stream, a, (const f4 *)upd, (const f4 *)m, lamC_over_L, lamH, n4);
    LAUNCHCHK();
    return TOMO_OK;
}

int tomo_positivity(tomo_engine *e, int vol)
{
    NEED(e);
    { int rc_ = order_after_async(e); if (rc_) return rc_; }
    float *x; int rc; if ((rc = get_vol(e, vol, &x))) return rc;
    int64_t n4 = e->vol_elems() / 4;
    hipLaunchKernelGGL(k_clamp, dim3(grid_1d(n4)), dim3(256), 0, e->stream, (f4 *)x, n4);
    LAUNCHCHK();
    return TOMO_OK;
}

int tomo_soft_threshold(tomo_engine *e, int vol, float lambda)
{
    NEED(e);
    float *x; int rc; if ((rc = get_vol(e, vol, &x))) return rc;
    int64_t n4 = e->vol_elems() / 4;
    hipLaunchKernelGGL(k_soft_threshold, dim3(grid_1d(n4)), dim3(256), 0, e->stream, (f4 *)x, lambda, n4);
    LAUNCHCHK();
    return TOMO_OK;
}

int tomo_fista_momentum(tomo_engine *e, float beta)
{
    NEED(e);
    // recon <- yk is a rotation of the two buffers, recon_old <- recon a flag (get_vol), and yk_new = r + beta (r - old) lands in
    // the buffer recon has just left -- which, from the second step on, is also where `old` sits (old == recon then): the step
    // reads two volumes and writes one (round 2: two reads, three stores: 645 us at 512^3).  Same expression, same bits.
    { int rc_ = order_after_async(e); if (rc_) return rc_; }
    // is the saved projection (tomo_fista_project_yk) A * (what recon_old holds now)?  recon_old was set to recon by the last step and
    // neither has been touched since, and the projection was saved from that very recon
    e->mom_p_ok = e->g_prev_valid && e->mom.set && e->vol_version[TOMO_VOL_RECON] == e->mom.ver_recon
                  && e->vol_version[TOMO_VOL_RECON_OLD] == e->mom.ver_old && e->g_prev_recon_ver == e->vol_version[TOMO_VOL_RECON];
    float *x, *yk, *old; int rc;
    const bool aliased = e->old_is_recon;
    if ((rc = get_vol_ro(e, TOMO_VOL_RECON, &x)) || (rc = get_vol_ro(e, TOMO_VOL_YK, &yk))) return rc;
    if (aliased) old = x;                                 // recon_old's content IS recon's
    else if ((rc = get_vol_ro(e, TOMO_VOL_RECON_OLD, &old))) return rc;
    int64_t n4 = e->vol_elems() / 4;
    hipLaunchKernelGGL(k_momentum, dim3(grid_1d(n4)), dim3(256), 0, e->stream, (const f4 *)yk, (const f4 *)old, (f4 *)x, beta, n4);
    LAUNCHCHK();
    ++e->vol_version[TOMO_VOL_RECON]; ++e->vol_version[TOMO_VOL_YK]; ++e->vol_version[TOMO_VOL_RECON_OLD];
    e->vol[TOMO_VOL_RECON] = yk;                          // the prox result r
    e->vol[TOMO_VOL_YK] = x;                              // r + beta (r - old), written over the buffer recon has left
    e->old_is_recon = true;                               // recon_old == r, not stored
    e->mom.beta = beta; e->mom.set = true;
    e->mom.ver_recon = e->vol_version[TOMO_VOL_RECON]; e->mom.ver_yk = e->vol_version[TOMO_VOL_YK]; e->mom.ver_old = e->vol_version[TOMO_VOL_RECON_OLD];
    return TOMO_OK;
}

// FISTA's next gradient step projects yk = r + beta (r - r_old).  The driver has just projected r for its cost
// (gpu/reconstructor.py:121-155: data_distance after every iteration) and projected r_old one iteration earlier, and the projector
// is linear: A yk = (1 + beta) A r - beta A r_old, a pass over two sinograms (94 MB) instead of a projection of the volume.  Call it
// after data_distance(recon); it does nothing unless every piece is provably in place (G = A recon as it stands, recon / yk /
// recon_old untouched since the last tomo_fista_momentum, the saved A r_old from the very iterate recon_old holds), and then leaves
// G = A yk with the claim the next tomo_sirt on yk picks up ("fp_reuse").  *done = 1 when the projection was formed.  Not
// bit-identical to projecting yk (rounding of the combination, ~1e-7 of |b|): parity tests hold it to the oracle at 1e-5 like
// every other path.
int tomo_fista_project_yk(tomo_engine *e, int *done)
{
    NEED(e);
    if (done) *done = 0;
    if (!e->fp_reuse || !g_is_projection_of(e, TOMO_VOL_RECON) || !e->mom.set) return TOMO_OK;
    if (e->vol_version[TOMO_VOL_RECON] != e->mom.ver_recon || e->vol_version[TOMO_VOL_YK] != e->mom.ver_yk) return TOMO_OK;
    { int rc_ = order_after_async(e); if (rc_) return rc_; }
    int rc; float *p, *q;
    if ((rc = get_sino(e, &e->g_prev, &p)) || (rc = get_sino(e, &e->g_yk, &q))) return rc;
    const float *g = e->sino[TOMO_SINO_G];
    const int64_t n4 = (int64_t)e->sino_elems() / 4;
    const bool have_prev = e->mom_p_ok;
    e->yk_claim.valid = false;
    if (have_prev) {   // q = A yk = (1 + beta) A r - beta A r_old; p = A r (the saved projection of the next step): one pass
        hipLaunchKernelGGL(k_sino_extrapolate, dim3(grid_1d(n4)), dim3(256), 0, e->stream, (const VecOf<4>::T *)g, (VecOf<4>::T *)p, (VecOf<4>::T *)q, e->mom.beta, n4);
        LAUNCHCHK();
    } else {
        HIPCHK(hipMemcpyAsync(p, g, e->sino_elems() * sizeof(float), hipMemcpyDeviceToDevice, e->stream));   // first step: only save A r
    }
    e->g_prev_valid = true;
    e->g_prev_recon_ver = e->vol_version[TOMO_VOL_RECON];
    // G is untouched: it stays A * recon, with its claim, and get_model_projections() returns what the reference's would
    if (have_prev) {
        e->yk_claim.valid = true; e->yk_claim.ver = e->vol_version[TOMO_VOL_YK];
        if (done) *done = 1;
    } else if (e->mom.beta == 0.f) {                        // yk is r, bit for bit: G is A * yk as well (the claim a copy inherits)
        e->g_valid[1].vol = TOMO_VOL_YK; e->g_valid[1].ver = e->vol_version[TOMO_VOL_YK];
        if (done) *done = 1;
    }
    return TOMO_OK;
}

// ---- scalars --------------------------------------------------------------------------------------------------------
int tomo_data_distance_sq(tomo_engine *e, int vol)
{
    NEED(e);
    float *x, *g; int rc;
    if ((rc = get_vol_ro(e, vol, &x)) || (rc = get_sino(e, &e->sino[TOMO_SINO_G], &g))) return rc;
    if ((rc = reduce_begin(e))) return rc;
    if ((rc = launch_fp_all<FP_DD>(e, x, e->sino[TOMO_SINO_B], g))) return rc;
    g_set(e, vol);                                      // FP_DD also stores g = A x
    return reduce_end(e, TOMO_S_DD);
}

// The data distance of a volume that the main sequence no longer modifies (e.g. the TEMP copy) can be evaluated on
// a second stream while the main stream goes on (ASD-POCS: the residual of the SART result next to the TV descent).
int tomo_data_distance_sq_async(tomo_engine *e, int vol)
{
    NEED(e);
    if (e->async_pending) return fail(TOMO_ERR_STATE, "an asynchronous evaluation is already in flight");
    if (!e->aux) {
        HIPCHK(hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
        int rc = dev_alloc((void **)&e->d_part_aux, NPART * sizeof(double), true, e->stream);
        if (rc) return rc;
    }
    float *tmp; int rc;
    if ((rc = get_vol_ro(e, vol, &tmp)) || (rc = get_sino(e, &e->sino[TOMO_SINO_G], &tmp))) return rc;   // allocate on the main stream
    HIPCHK(hipEventRecord(e->ev_fork, e->stream));
    HIPCHK(hipStreamWaitEvent(e->aux, e->ev_fork, 0));
    hipStream_t main_stream = e->stream;
    double *main_part = e->d_part;
    e->stream = e->aux; e->d_part = e->d_part_aux;
    rc = tomo_data_distance_sq(e, vol);
    e->stream = main_stream; e->d_part = main_part;
    if (rc) return rc;
    HIPCHK(hipEventRecord(e->ev_join, e->aux));
    e->async_pending = true;
    return TOMO_OK;
}

int tomo_async_wait(tomo_engine *e)
{
    NEED(e);
    if (e->async_pending) {
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join, 0));
        e->async_pending = false;
    }
    return TOMO_OK;
}

int tomo_diff_norm_sq(tomo_engine *e, int a, int b, int slot)
{
    NEED(e);
    if (slot < 0 || slot >= TOMO_S_COUNT) return fail(TOMO_ERR_ARG, "bad scalar slot");
    float *pa, *pb; int rc;
    if ((rc = get_vol_ro(e, a, &pa)) || (rc = get_vol_ro(e, b, &pb))) return rc;
    if ((rc = reduce_begin(e))) return rc;
    int64_t n4 = e->vol_elems() / 4;
    hipLaunchKernelGGL(k_sqdiff, dim3(grid_1d(n4)), dim3(256), 0, e->stream, (const f4 *)pa, (const f4 *)pb, e->d_part, n4);
    LAUNCHCHK();
    return reduce_end(e, slot);
}

int tomo_l1_norm(tomo_engine *e, int vol)
{
    NEED(e);
    float *x; int rc; if ((rc = get_vol_ro(e, vol, &x))) return rc;
    if ((rc = reduce_begin(e))) return rc;
    int64_t n4 = e->vol_elems() / 4;
    hipLaunchKernelGGL(k_l1, dim3(grid_1d(n4)), dim3(256), 0, e->stream, (const f4 *)x, e->d_part, n4);
    LAUNCHCHK();
    return reduce_end(e, TOMO_S_L1);
}

int tomo_read_scalars(tomo_engine *e, double *out, int count)
{
    NEED(e);
    if (!out || count < 0 || count > TOMO_S_COUNT) return fail(TOMO_ERR_ARG, "bad scalar count");
    { int rc = tomo_async_wait(e); if (rc) return rc; }
    HIPCHK(hipMemcpyAsync(out, e->d_scal, count * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TOMO_OK;
}

// The scalars of an iteration without a pipeline bubble: the copy is enqueued behind the kernels that produce them, the host
// goes on enqueueing (the next SART sweep) and collects the values later (tomo_scalars_snapshot_read waits on the event only).
int tomo_scalars_snapshot(tomo_engine *e)
{
    NEED(e);
    { int rc = tomo_async_wait(e); if (rc) return rc; }        // device-side ordering behind the second stream's evaluation
    if (!e->h_snap) {
        HIPCHK(hipHostMalloc((void **)&e->h_snap, TOMO_S_COUNT * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&e->ev_snap, hipEventDisableTiming));
    }
    hipLaunchKernelGGL(k_scalars_to_host, dim3(1), dim3(64), 0, e->stream, (const double *)e->d_scal, e->h_snap, (int)TOMO_S_COUNT);
    LAUNCHCHK();
    HIPCHK(hipEventRecord(e->ev_snap, e->stream));
    e->snap_pending = true;
    return TOMO_OK;
}

int tomo_scalars_snapshot_read(tomo_engine *e, double *out, int count)
{
    NEED(e);
    if (!out || count < 0 || count > TOMO_S_COUNT) return fail(TOMO_ERR_ARG, "bad scalar count");
    if (!e->snap_pending) return fail(TOMO_ERR_STATE, "no scalar snapshot in flight");
    HIPCHK(hipEventSynchronize(e->ev_snap));
    std::memcpy(out, e->h_snap, count * sizeof(double));
    e->snap_pending = false;
    return TOMO_OK;
}

int tomo_bind_scalar_buffer(tomo_engine *e, void *device_doubles)
{
    NEED(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    e->d_scal = device_doubles ? (double *)device_doubles : e->d_scal_own;
    return TOMO_OK;
}

// ---- TV -----------------------------------------------------------------------------------------------------------------
static int field_ptr(tomo_engine *e, int field, float **out)
{
    if (field == TOMO_FIELD_FGP_D) return get_scratch(e, &e->tvg, out);
    if (field == TOMO_FIELD_FGP_P1) return get_scratch(e, &e->fgp_p[0], out);
    return get_vol_ro(e, field, out);   // halo packs and fills only read
}

int tomo_bind_halo(tomo_engine *e, void *device_lo, void *device_hi)
{
    NEED(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    e->halo_lo = device_lo ? (float *)device_lo : e->halo_lo_own;
    e->halo_hi = device_hi ? (float *)device_hi : e->halo_hi_own;
    return TOMO_OK;
}

int tomo_halo_pack(tomo_engine *e, int field, int last, void *device_dst)
{
    NEED(e);
    float *x; int rc; if ((rc = field_ptr(e, field, &x))) return rc;
    if (!device_dst) return fail(TOMO_ERR_ARG, "null destination");
    hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)((e->npix + 255) / 256)), dim3(256), 0, e->stream, x, (float *)device_dst, (int)e->npix, e->sx, last ? e->nx - 1 : 0);
    LAUNCHCHK();
    return TOMO_OK;
}

int tomo_halo_pack_both(tomo_engine *e, int field, void *first_plane, void *last_plane)
{
    NEED(e);
    float *x; int rc; if ((rc = field_ptr(e, field, &x))) return rc;
    if (!first_plane || !last_plane) return fail(TOMO_ERR_ARG, "null destination");
    // k_halo_wrap(x, lo, hi): lo <- last slice, hi <- slice 0
    hipLaunchKernelGGL(k_halo_wrap, dim3((unsigned)((e->npix + 255) / 256)), dim3(256), 0, e->stream, x, (float *)last_plane, (float *)first_plane, (int)e->npix, e->sx, e->nx);
    LAUNCHCHK();
    return TOMO_OK;
}

int tomo_halo_local(tomo_engine *e, int field)
{
    NEED(e);
    float *x; int rc; if ((rc = field_ptr(e, field, &x))) return rc;
    // below slice 0 sits the last slice, above the last slice sits slice 0
    hipLaunchKernelGGL(k_halo_wrap, dim3((unsigned)((e->npix + 255) / 256)), dim3(256), 0, e->stream, x, e->halo_lo, e->halo_hi, (int)e->npix, e->sx, e->nx);
    LAUNCHCHK();
    return TOMO_OK;
}

// ---- several slab engines on ONE device (tomo_tv_amd/engine.py: _GroupBackend) ------------------------------------------------
// A slab can be run as K sub-slabs, each a complete engine with its own allocations and stream: the dependent launch chains of
// the sub-slabs (a SART sweep is 180 of them) then fill each other's launch gaps and kernel tails.  Slices only couple in the 3-D
// TV stencils and in the global sums; these three calls are what the coupling needs.

// e's stream waits for everything enqueued on other's stream so far
int tomo_wait_for(tomo_engine *e, tomo_engine *other)
{
    NEED(e);
    if (!other) return fail(TOMO_ERR_ARG, "null engine");
    if (other->device != e->device) return fail(TOMO_ERR_ARG, "engines on different devices");
    if (other == e || other->stream == e->stream) return TOMO_OK;
    if (!other->ev_peer) HIPCHK(hipEventCreateWithFlags(&other->ev_peer, hipEventDisableTiming));
    HIPCHK(hipEventRecord(other->ev_peer, other->stream));
    HIPCHK(hipStreamWaitEvent(e->stream, other->ev_peer, 0));
    return TOMO_OK;
}

// e's halo planes from its neighbours' volumes: lo = LAST slice of lo_src's field, hi = FIRST slice of hi_src's field (the
// caller orders the streams: tomo_wait_for).  The ring of sub-slabs of one volume gives the periodic wrap of ctvlib.cpp:348,421.
int tomo_halo_from(tomo_engine *e, int field, tomo_engine *lo_src, tomo_engine *hi_src)
{
    NEED(e);
    if (!lo_src || !hi_src) return fail(TOMO_ERR_ARG, "null engine");
    if (lo_src->n != e->n || hi_src->n != e->n || lo_src->device != e->device || hi_src->device != e->device) return fail(TOMO_ERR_ARG, "engines differ in image size or device");
    float *xl, *xh; int rc;
    if ((rc = field_ptr(lo_src, field, &xl)) || (rc = field_ptr(hi_src, field, &xh))) return rc;
    dim3 grid((unsigned)((e->npix + 255) / 256));
    hipLaunchKernelGGL(k_halo_pack, grid, dim3(256), 0, e->stream, xl, e->halo_lo, (int)e->npix, lo_src->sx, lo_src->nx - 1);
    hipLaunchKernelGGL(k_halo_pack, grid, dim3(256), 0, e->stream, xh, e->halo_hi, (int)e->npix, hi_src->sx, 0);
    LAUNCHCHK();
    return TOMO_OK;
}

// scalar dst_slot of e = sum over the engines' src_slot partial sums, on e's stream (caller orders the streams)
int tomo_scalar_sum_from(tomo_engine *e, int dst_slot, tomo_engine **srcs, int n, int src_slot)
{
    NEED(e);
    if (!srcs || n < 1 || n > 8 || dst_slot < 0 || dst_slot >= TOMO_S_COUNT || src_slot < 0 || src_slot >= TOMO_S_COUNT) return fail(TOMO_ERR_ARG, "bad argument");
    SumSrc s{};
    s.n = n;
    for (int i = 0; i < n; ++i) { if (!srcs[i] || srcs[i]->device != e->device) return fail(TOMO_ERR_ARG, "bad source engine"); s.p[i] = srcs[i]->d_scal + src_slot; }
    hipLaunchKernelGGL(k_sum_doubles, dim3(1), dim3(1), 0, e->stream, s, e->d_scal + dst_slot);
    LAUNCHCHK();
    return TOMO_OK;
}

int tomo_set_slab_edges(tomo_engine *e, int is_first, int is_last)
{
    if (!e) return fail(TOMO_ERR_ARG, "null engine");
    e->is_first = is_first ? 1 : 0; e->is_last = is_last ? 1 : 0;
    return TOMO_OK;
}

static int tv_grid(tomo_engine *e)
{
    int64_t items = e->npix * (e->sx / 64);
    return (int)std::min<int64_t>((items + 3) / 4, 256 * 16);
}

// TV of a volume with the halo planes as they are (step form: the caller has exchanged or wrapped them)
int tomo_tv_partial(tomo_engine *e, int vol, float eps)
{
    NEED(e);
    float *x; int rc; if ((rc = get_vol_ro(e, vol, &x))) return rc;
    Halo h{e->halo_lo, e->halo_hi};
    if (e->tv_lds != 8 && e->tv_lds != 1) {   // direct-global stencil (reads x twice)
        if ((rc = reduce_begin(e))) return rc;
        hipLaunchKernelGGL(k_tv_value, dim3(tv_grid(e)), dim3(256), 0, e->stream, x, h, e->d_part, eps, e->n, e->nx, e->sx);
        LAUNCHCHK();
        return reduce_end(e, TOMO_S_TV);
    }
    // the march of the gradient kernels without their gradient half: x is read once
    if (!e->d_part_tv) { if ((rc = dev_alloc((void **)&e->d_part_tv, NPART * sizeof(double), true, e->stream))) return rc; }
    if ((rc = part_begin(e, e->d_part_tv))) return rc;
    const int yseg = 32;
    if (e->tv_lds == 1 && e->tv_march4 && e->nx % 64 == 0 && e->n % 8 == 0) {
        // the value alone from the branch-free march (round 3): the R loop without the gradient half
        hipLaunchKernelGGL((k_tv_march4<8, true, TVM_VALUE, false>), dim3(tv_march_grid(e->n, 8, e->sxc / 64, (e->n + yseg - 1) / yseg)), dim3(256), 0, e->stream, x, h, (double *)nullptr, eps, e->n, e->nx, e->sx, yseg, e->d_part_tv, TvUpd{});
    } else if (e->tv_lds == 1) {
        hipLaunchKernelGGL((k_tv_grad_reg<8, true, false>), dim3(tv_march_grid(e->n, 8, e->sxc / 64, (e->n + yseg - 1) / yseg)), dim3(256), 0, e->stream, x, h, (float *)nullptr, (double *)nullptr, eps, e->n, e->nx, e->sx, yseg, e->d_part_tv, TvUpd{});
    } else {
        dim3 grid((unsigned)(((e->n + 7) / 8) * (e->sxc / 64)), (unsigned)((e->n + yseg - 1) / yseg));
        hipLaunchKernelGGL((k_tv_grad_lds<8, true, false>), grid, dim3(256), 0, e->stream, x, h, (float *)nullptr, (double *)nullptr, eps, e->n, e->nx, e->sx, yseg, e->d_part_tv);
    }
    LAUNCHCHK();
    return part_end(e, e->d_part_tv, TOMO_S_TV);
}

// Rows a wave of the register march walks.  32 at a full slab (8 chunks x 64 z-blocks x 16 segments = 8192 waves at 512^3);
// a thin slab of a multi-GPU run has too few waves at that length (64 slices: 1024 waves, 4 per CU), so the segments
// shrink until there are ~8 waves per SIMD lane group again -- the 2 halo rows a segment re-reads cost less than the idle CUs.
static int tv_rows_per_wave(const tomo_engine *e, int tz)
{
    if (e->tv_yseg > 0) return e->tv_yseg;
    const int64_t cols = (int64_t)((e->n + tz - 1) / tz) * (e->sxc / 64);
    int yseg = 32;
    while (yseg > TV_YSEG_MIN && cols * ((e->n + yseg - 1) / yseg) < TV_WAVES_WANTED) yseg >>= 1;
    return yseg;
}

static int tv_grad_impl(tomo_engine *e, float eps, bool with_tv, float *g_first = nullptr, float *g_last = nullptr)
{
    NEED(e);
    float *x, *g; int rc;
    if ((rc = get_vol(e, e->tv_target, &x))) return rc;
    if ((rc = get_scratch(e, &e->tvg, &g))) return rc;
    if ((rc = reduce_begin(e))) return rc;
    Halo h{e->halo_lo, e->halo_hi};
    if ((g_first || g_last) && !(e->tv_lds == 1 && e->tv_recompute)) return fail(TOMO_ERR_STATE, "gradient planes need the recompute form of the TV march (tv_lds = 1, tv_recompute = 1)");
    if (with_tv && e->tv_lds != 8 && e->tv_lds != 1) with_tv = false;
    if (with_tv) {
        if (!e->d_part_tv) { if ((rc = dev_alloc((void **)&e->d_part_tv, NPART * sizeof(double), true, e->stream))) return rc; }
        if ((rc = part_begin(e, e->d_part_tv))) return rc;
    }
    {
        ProfScope ps(e, TOMO_K_TV_GRAD);
        e->tv_last_eps = eps;
        if (e->tv_lds == 1 && e->tv_recompute) {   // sum g^2 (and TV) only: the update pass re-evaluates g (TVM_UPDATE)
            TvUpd gp{};
            gp.wrap_lo = g_last; gp.wrap_hi = g_first;      // g's last / first slice (slab-sharded descent), or null
            const int yseg = tv_rows_per_wave(e, e->tv_tz == 4 ? 4 : 8);
            if (e->tv_tz == 4) {
                dim3 grid(tv_march_grid(e->n, 4, e->sxc / 64, (e->n + yseg - 1) / yseg));
                if (with_tv) hipLaunchKernelGGL((k_tv_grad_reg<4, true, true, TVM_NORM>), grid, dim3(256), 0, e->stream, x, h, (float *)nullptr, e->d_part, eps, e->n, e->nx, e->sx, yseg, e->d_part_tv, gp);
                else hipLaunchKernelGGL((k_tv_grad_reg<4, false, true, TVM_NORM>), grid, dim3(256), 0, e->stream, x, h, (float *)nullptr, e->d_part, eps, e->n, e->nx, e->sx, yseg, (double *)nullptr, gp);
            } else {
            dim3 grid(tv_march_grid(e->n, 8, e->sxc / 64, (e->n + yseg - 1) / yseg));
            if (e->tv_march4) {
                const bool edge = e->nx % 64 != 0 || e->n % 8 != 0;      // lanes without a voxel exist: the predicated form
#define TV4_NORM(WTV, EDGE, PTV) hipLaunchKernelGGL((k_tv_march4<8, WTV, TVM_NORM, EDGE>), grid, dim3(256), 0, e->stream, x, h, e->d_part, eps, e->n, e->nx, e->sx, yseg, PTV, gp)
                if (with_tv) { if (edge) TV4_NORM(true, true, e->d_part_tv); else TV4_NORM(true, false, e->d_part_tv); }
                else { if (edge) TV4_NORM(false, true, (double *)nullptr); else TV4_NORM(false, false, (double *)nullptr); }
#undef TV4_NORM
            }
            else if (with_tv) hipLaunchKernelGGL((k_tv_grad_reg<8, true, true, TVM_NORM>), grid, dim3(256), 0, e->stream, x, h, (float *)nullptr, e->d_part, eps, e->n, e->nx, e->sx, yseg, e->d_part_tv, gp);
            else hipLaunchKernelGGL((k_tv_grad_reg<8, false, true, TVM_NORM>), grid, dim3(256), 0, e->stream, x, h, (float *)nullptr, e->d_part, eps, e->n, e->nx, e->sx, yseg, (double *)nullptr, gp);
            }
        } else if (e->tv_lds == 1) {   // register march (k_tv_grad_reg): one wave per (z block, chunk, y segment)
            int yseg = 32;   // 8 .. 64 rows per wave measured the same; longer segments leave too few waves
            dim3 grid(tv_march_grid(e->n, 8, e->sxc / 64, (e->n + yseg - 1) / yseg));
            if (with_tv) hipLaunchKernelGGL((k_tv_grad_reg<8, true>), grid, dim3(256), 0, e->stream, x, h, g, e->d_part, eps, e->n, e->nx, e->sx, yseg, e->d_part_tv, TvUpd{});
            else hipLaunchKernelGGL((k_tv_grad_reg<8, false>), grid, dim3(256), 0, e->stream, x, h, g, e->d_part, eps, e->n, e->nx, e->sx, yseg, (double *)nullptr, TvUpd{});
        } else if (e->tv_lds) {
            int yseg = 32;
            if (e->tv_lds == 16) {
                dim3 grid((unsigned)(((e->n + 15) / 16) * (e->sxc / 64)), (unsigned)((e->n + yseg - 1) / yseg));
                hipLaunchKernelGGL((k_tv_grad_lds<16, false>), grid, dim3(256), 0, e->stream, x, h, g, e->d_part, eps, e->n, e->nx, e->sx, yseg, (double *)nullptr);
            } else {
                dim3 grid((unsigned)(((e->n + 7) / 8) * (e->sxc / 64)), (unsigned)((e->n + yseg - 1) / yseg));
                if (with_tv) hipLaunchKernelGGL((k_tv_grad_lds<8, true>), grid, dim3(256), 0, e->stream, x, h, g, e->d_part, eps, e->n, e->nx, e->sx, yseg, e->d_part_tv);
                else hipLaunchKernelGGL((k_tv_grad_lds<8, false>), grid, dim3(256), 0, e->stream, x, h, g, e->d_part, eps, e->n, e->nx, e->sx, yseg, (double *)nullptr);
            }
        } else {
            hipLaunchKernelGGL(k_tv_grad, dim3(tv_grid(e)), dim3(256), 0, e->stream, x, h, g, e->d_part, eps, e->n, e->nx, e->sx);
        }
    }
    LAUNCHCHK();
    if (with_tv && (rc = part_end(e, e->d_part_tv, TOMO_S_TV))) return rc;
    return reduce_end(e, TOMO_S_GNORM);
}

int tomo_tv_set_target(tomo_engine *e, int vol)
{
    NEED(e);
    float *x;
    int rc = get_vol(e, vol, &x);
    if (rc) return rc;
    e->tv_target = vol;
    return TOMO_OK;
}

int tomo_tv_grad(tomo_engine *e, float eps) { return tv_grad_impl(e, eps, false); }
int tomo_tv_grad_tv(tomo_engine *e, float eps)
{
    if (e && e->tv_lds != 8 && e->tv_lds != 1) {   // kernels without the folded value: a separate pass
        int rc = tomo_tv_partial(e, e->tv_target, eps);
        return rc ? rc : tv_grad_impl(e, eps, false);
    }
    return tv_grad_impl(e, eps, true);
}

// Slab-sharded descent, one communication round per inner iteration: the norm pass also leaves the gradient's first and last
// slice in caller buffers (what the neighbours need to advance their halo planes themselves: tomo_tv_halo_apply).
int tomo_tv_grad_planes(tomo_engine *e, float eps, int with_tv, void *g_first, void *g_last)
{
    if (!g_first || !g_last) return fail(TOMO_ERR_ARG, "null plane buffer");
    return tv_grad_impl(e, eps, with_tv != 0, (float *)g_first, (float *)g_last);
}

// halo planes <- the neighbours' update of those slices: halo - (dPOCS g)/||g|| (clamped), with the received gradient planes
// g_lo (the lower neighbour's last slice) and g_hi (the upper neighbour's first slice).  Call it AFTER the update pass (which
// still reads the old planes).
int tomo_tv_halo_apply(tomo_engine *e, float dPOCS, int clamp, const void *g_lo, const void *g_hi)
{
    NEED(e);
    if (!g_lo || !g_hi) return fail(TOMO_ERR_ARG, "null plane buffer");
    hipLaunchKernelGGL(k_halo_apply, dim3((unsigned)((e->npix + 255) / 256)), dim3(256), 0, e->stream, e->halo_lo, e->halo_hi,
                       (const float *)g_lo, (const float *)g_hi, gnorm_ptr(e), dPOCS, clamp, (int)e->npix);
    LAUNCHCHK();
    return TOMO_OK;
}

// wrap: also write the new last / first slice into the engine's halo planes (single slab, periodic); plane_last /
// plane_first: into caller buffers instead (slab-sharded: the planes the ring exchange sends next)
// hg_lo / hg_hi (slab-sharded descent): the gradient planes received from the ring neighbours; the halo planes are advanced with them
// (k_halo_apply's expression) -- inside the update pass where it can write a second pair of planes ("tv_halo_fold", the march4 form, the
// engine's own planes), by a k_halo_apply launch behind it otherwise
static int tv_update_impl(tomo_engine *e, float dPOCS, int clamp, int track_vol, int slot, bool wrap = false,
                          float *plane_last = nullptr, float *plane_first = nullptr, const float *hg_lo = nullptr, const float *hg_hi = nullptr)
{
    NEED(e);
    float *x, *g, *track = nullptr; int rc;
    if ((rc = get_vol(e, e->tv_target, &x))) return rc;
    if ((rc = get_scratch(e, &e->tvg, &g))) return rc;
    if (track_vol >= 0) {
        if (track_vol == e->tv_target) return fail(TOMO_ERR_ARG, "the tracked volume must differ from the one descended");
        if (slot < 0 || slot >= TOMO_S_COUNT) return fail(TOMO_ERR_ARG, "bad scalar slot");
        if ((rc = get_vol(e, track_vol, &track))) return rc;
        // an evaluation in flight on the second stream may still read the tracked volume (ASD-POCS: the data distance
        // of the snapshot): order this write behind it on the device; tomo_async_wait later is still valid
        if ((rc = order_after_async(e))) return rc;
        if ((rc = reduce_begin(e))) return rc;
    }
    if (e->tv_lds == 1 && e->tv_recompute) {
        // recompute-and-update pass: reads x (+ the halo planes the norm pass used), writes x_new into the second buffer,
        // which then becomes the volume.  The wrapped planes of x_new go to a second pair of halo buffers (this pass still
        // reads the old ones), swapped in afterwards; halo buffers bound by the caller are refreshed by a gather launch.
        float *alt;
        if ((rc = get_scratch(e, &e->tv_alt, &alt))) return rc;
        const bool own_halo = e->halo_lo == e->halo_lo_own || e->halo_lo == e->halo_lo_alt;
        float *wl = plane_last, *wh = plane_first;
        if (wrap && own_halo) {
            if (!e->halo_lo_alt) {
                if ((rc = dev_alloc((void **)&e->halo_lo_alt, e->npix * sizeof(float), true, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->halo_hi_alt, e->npix * sizeof(float), true, e->stream))) return rc;
            }
            wl = e->halo_lo == e->halo_lo_own ? e->halo_lo_alt : e->halo_lo_own;
            wh = e->halo_hi == e->halo_hi_own ? e->halo_hi_alt : e->halo_hi_own;
        }
        Halo h{e->halo_lo, e->halo_hi};
        TvUpd up{alt, gnorm_ptr(e), dPOCS, clamp, track, wl, wh, slab_streams(e) ? 1 : 0, nullptr, nullptr, nullptr, nullptr};
        const bool fold = hg_lo && hg_hi && e->tv_halo_fold && own_halo && !wrap && e->tv_tz != 4 && e->tv_march4;
        if (fold) {
            if (!e->halo_lo_alt) {
                if ((rc = dev_alloc((void **)&e->halo_lo_alt, e->npix * sizeof(float), true, e->stream))) return rc;
                if ((rc = dev_alloc((void **)&e->halo_hi_alt, e->npix * sizeof(float), true, e->stream))) return rc;
            }
            up.hg_lo = hg_lo; up.hg_hi = hg_hi;
            up.ho_lo = e->halo_lo == e->halo_lo_own ? e->halo_lo_alt : e->halo_lo_own;
            up.ho_hi = e->halo_hi == e->halo_hi_own ? e->halo_hi_alt : e->halo_hi_own;
        }
        {
            ProfScope ps(e, TOMO_K_TV_UPDATE);
            const int yseg = tv_rows_per_wave(e, e->tv_tz == 4 ? 4 : 8);
            if (e->tv_tz == 4) {
                dim3 grid(tv_march_grid(e->n, 4, e->sxc / 64, (e->n + yseg - 1) / yseg));
                hipLaunchKernelGGL((k_tv_grad_reg<4, false, true, TVM_UPDATE>), grid, dim3(256), 0, e->stream, x, h, (float *)nullptr, e->d_part, e->tv_last_eps, e->n, e->nx, e->sx, yseg, (double *)nullptr, up);
            } else {
            dim3 grid(tv_march_grid(e->n, 8, e->sxc / 64, (e->n + yseg - 1) / yseg));
            if (e->tv_march4) {
                const bool edge = e->nx % 64 != 0 || e->n % 8 != 0, trk = track != nullptr, strm = up.stream != 0;
#define TV4_UPD(EDGE, TRK, STRM) hipLaunchKernelGGL((k_tv_march4<8, false, TVM_UPDATE, EDGE, TRK, STRM>), grid, dim3(256), 0, e->stream, x, h, e->d_part, e->tv_last_eps, e->n, e->nx, e->sx, yseg, (double *)nullptr, up)
                if (edge) { if (trk) { if (strm) TV4_UPD(true, true, true); else TV4_UPD(true, true, false); } else { if (strm) TV4_UPD(true, false, true); else TV4_UPD(true, false, false); } }
                else { if (trk) { if (strm) TV4_UPD(false, true, true); else TV4_UPD(false, true, false); } else { if (strm) TV4_UPD(false, false, true); else TV4_UPD(false, false, false); } }
#undef TV4_UPD
            }
            else hipLaunchKernelGGL((k_tv_grad_reg<8, false, true, TVM_UPDATE>), grid, dim3(256), 0, e->stream, x, h, (float *)nullptr, e->d_part, e->tv_last_eps, e->n, e->nx, e->sx, yseg, (double *)nullptr, up);
            }
        }
        LAUNCHCHK();
        e->vol[e->tv_target] = alt; e->tv_alt = x;              // the updated volume lives in the partner buffer
        if (wrap && own_halo) { e->halo_lo = wl; e->halo_hi = wh; }
        else if (wrap) { if ((rc = tomo_halo_local(e, e->tv_target))) return rc; }
        if (fold) { e->halo_lo = up.ho_lo; e->halo_hi = up.ho_hi; }
        else if (hg_lo && hg_hi && (rc = tomo_tv_halo_apply(e, dPOCS, clamp, hg_lo, hg_hi))) return rc;
        return track ? reduce_end(e, slot) : TOMO_OK;
    }
    int64_t n4 = e->vol_elems() / 4;
    {
        ProfScope ps(e, TOMO_K_TV_UPDATE);
        float *wl = wrap ? e->halo_lo : plane_last, *wh = wrap ? e->halo_hi : plane_first;
        if (track) hipLaunchKernelGGL(k_tv_update<true>, dim3(grid_1d(n4)), dim3(256), 0, e->stream, (f4 *)x, (const f4 *)g, gnorm_ptr(e), dPOCS, clamp, n4, (f4 *)track, e->d_part, wl, wh, e->nx, e->sx / 4);
        else hipLaunchKernelGGL(k_tv_update<false>, dim3(grid_1d(n4)), dim3(256), 0, e->stream, (f4 *)x, (const f4 *)g, gnorm_ptr(e), dPOCS, clamp, n4, (f4 *)nullptr, (double *)nullptr, wl, wh, e->nx, e->sx / 4);
    }
    LAUNCHCHK();
    if (hg_lo && hg_hi && (rc = tomo_tv_halo_apply(e, dPOCS, clamp, hg_lo, hg_hi))) return rc;
    return track ? reduce_end(e, slot) : TOMO_OK;
}

int tomo_tv_update(tomo_engine *e, float dPOCS, int clamp) { return tv_update_impl(e, dPOCS, clamp, -1, 0); }

int tomo_tv_update_planes(tomo_engine *e, float dPOCS, int clamp, void *first_plane, void *last_plane)
{
    if (!first_plane || !last_plane) return fail(TOMO_ERR_ARG, "null plane buffer");
    return tv_update_impl(e, dPOCS, clamp, -1, 0, false, (float *)last_plane, (float *)first_plane);
}

int tomo_tv_update_tracked(tomo_engine *e, float dPOCS, int clamp, int track_vol, int slot)
{
    return tv_update_impl(e, dPOCS, clamp, track_vol, slot);
}

int tomo_fgp_begin(tomo_engine *e) { return tomo_fgp_begin_vol(e, TOMO_VOL_RECON); }

static int fgp_begin_impl(tomo_engine *e, int vol, bool zero_p)
{
    NEED(e);
    float *d, *p; int rc;
    if ((rc = get_vol(e, vol, &d))) return rc;
    e->fgp_target = vol;
    if (zero_p) {   // step form: D and P start as zero fields (tv_fgp.cu:216-227); the fused form needs neither
        if ((rc = get_scratch(e, &e->tvg, &d))) return rc;
        HIPCHK(hipMemsetAsync(d, 0, e->vol_elems() * sizeof(float), e->stream));
    }
    for (int i = 0; i < 3; ++i) {
        if ((rc = get_scratch(e, &e->fgp_p[i], &p))) return rc;
        if (zero_p) HIPCHK(hipMemsetAsync(p, 0, e->vol_elems() * sizeof(float), e->stream));
    }
    return TOMO_OK;
}

int tomo_fgp_begin_vol(tomo_engine *e, int vol) { return fgp_begin_impl(e, vol, true); }

int tomo_fgp_obj(tomo_engine *e, float lambda)
{
    NEED(e);
    if (!e->tvg || !e->fgp_p[2]) return fail(TOMO_ERR_STATE, "tomo_fgp_begin has not been called");
    ProfScope ps(e, TOMO_K_FGP_OBJ);
    hipLaunchKernelGGL(k_fgp_obj, dim3(tv_grid(e)), dim3(256), 0, e->stream, e->vol[e->fgp_target], e->tvg, e->fgp_p[0], e->fgp_p[1], e->fgp_p[2], e->halo_lo, e->is_first, lambda, e->n, e->nx, e->sx);
    LAUNCHCHK();
    return TOMO_OK;
}

int tomo_fgp_grad(tomo_engine *e, float lambda)
{
    NEED(e);
    if (!e->tvg || !e->fgp_p[2]) return fail(TOMO_ERR_STATE, "tomo_fgp_begin has not been called");
    float multip = 1.0f / (26.0f * lambda);
    ProfScope ps(e, TOMO_K_FGP_GRAD);
    hipLaunchKernelGGL(k_fgp_grad, dim3(tv_grid(e)), dim3(256), 0, e->stream, e->tvg, e->fgp_p[0], e->fgp_p[1], e->fgp_p[2], e->halo_hi, e->is_last, multip, e->n, e->nx, e->sx);
    LAUNCHCHK();
    return TOMO_OK;
}

int tomo_fgp_end(tomo_engine *e, int iters)
{
    if (e && e->fgp_target >= 0 && e->fgp_target < TOMO_VOL_SLOTS) ++e->vol_version[e->fgp_target];   // the prox result lands in the volume

    NEED(e);
    if (!e->tvg) return fail(TOMO_ERR_STATE, "tomo_fgp_begin has not been called");
    (void)iters;  // D is the zero-filled buffer when no iteration ran, exactly like d_update (tv_fgp.cu:223,272)
    HIPCHK(hipMemcpyAsync(e->vol[e->fgp_target], e->tvg, e->vol_elems() * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    return TOMO_OK;
}

int tomo_tv(tomo_engine *e, int vol, float eps)
{
    int rc;
    if ((rc = tomo_halo_local(e, vol))) return rc;
    return tomo_tv_partial(e, vol, eps);
}

static int tv_gd_impl(tomo_engine *e, int ng, float dPOCS, float eps, int track_vol, int slot)
{
    int rc;
    if (!e) return fail(TOMO_ERR_ARG, "null engine");
    // the TV value before descent comes out of the first gradient pass (its denominators are the TV integrand)
    const bool fold_tv = ng > 0 && (e->tv_lds == 8 || e->tv_lds == 1);
    if (fold_tv) { if ((rc = tomo_halo_local(e, e->tv_target))) return rc; }
    else if ((rc = tomo_tv(e, e->tv_target, eps))) return rc;
    for (int g = 0; g < ng; ++g) {
        // single slab: every descent step but the last also writes the wrapped halo planes of its result
        if ((rc = tv_grad_impl(e, eps, fold_tv && g == 0))) return rc;
        if ((rc = tv_update_impl(e, dPOCS, g == ng - 1, g == ng - 1 ? track_vol : -1, slot, g < ng - 1))) return rc;
    }
    if (ng <= 0) {
        if ((rc = tomo_positivity(e, e->tv_target))) return rc;
        if (track_vol >= 0) {
            if ((rc = tomo_diff_norm_sq(e, e->tv_target, track_vol, slot))) return rc;
            return tomo_copy_volume(e, track_vol, e->tv_target);
        }
    }
    return TOMO_OK;
}

int tomo_tv_gd(tomo_engine *e, int ng, float dPOCS, float eps) { return tv_gd_impl(e, ng, dPOCS, eps, -1, 0); }

int tomo_tv_gd_tracked(tomo_engine *e, int ng, float dPOCS, float eps, int track_vol, int slot)
{
    if (track_vol < 0) return fail(TOMO_ERR_ARG, "bad tracked volume");
    return tv_gd_impl(e, ng, dPOCS, eps, track_vol, slot);
}

int tomo_tv_fgp(tomo_engine *e, int iters, float lambda) { return tomo_tv_fgp_vol(e, TOMO_VOL_RECON, iters, lambda); }

// ---- fused FGP iteration, step form (single slab and slab-sharded) ------------------------------------------------------
int tomo_bind_fgp_halo(tomo_engine *e, void *lo, void *hi, void *send_first, void *send_last)
{
    NEED(e);
    if (!lo || !hi || !send_first || !send_last) return fail(TOMO_ERR_ARG, "null plane buffer");
    HIPCHK(hipStreamSynchronize(e->stream));
    e->fgp_lo = (float *)lo; e->fgp_hi = (float *)hi; e->fgp_send_first = (float *)send_first; e->fgp_send_last = (float *)send_last;
    e->fgp_planes2 = false;
    return TOMO_OK;
}

// the two-slice-deep set (k_fgp_fused2<.., SHARDED>): lo 5 planes [P1(-1), A(-1), P2(-1), P3(-1), P1(-2)], hi 8 planes
// [A, P1, P2, P3](nx), [..](nx + 1), send_first 8 planes [A, P1, P2, P3](0), [..](1), send_last 5 planes [P1(nx-1), A(nx-1), P2(nx-1),
// P3(nx-1), P1(nx-2)].  The one-deep planes of tomo_bind_fgp_halo are the prefixes (1 / 4 / 4 / 1), so both step forms run on it.
int tomo_bind_fgp_halo2(tomo_engine *e, void *lo, void *hi, void *send_first, void *send_last)
{
    int rc = tomo_bind_fgp_halo(e, lo, hi, send_first, send_last);
    if (rc) return rc;
    e->fgp_planes2 = true;
    return TOMO_OK;
}

// sharded: slab faces that are not global edges read / write the bound planes
static bool fgp_sharded(const tomo_engine *e) { return !(e->is_first && e->is_last); }

int tomo_fgp_fused_begin(tomo_engine *e, int vol)
{
    int rc;
    if ((rc = fgp_begin_impl(e, vol, false))) return rc;     // the first iteration takes P = 0 as known
    float *q;
    for (int i = 0; i < 3; ++i) if ((rc = get_scratch(e, &e->fgp_q[i], &q))) return rc;
    if (fgp_sharded(e)) {
        if (!e->fgp_lo) return fail(TOMO_ERR_STATE, "slab-sharded fused FGP needs tomo_bind_fgp_halo");
        // plane 0 of send_first: the first slice of the prox input (constant over the call)
        hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)((e->npix + 255) / 256)), dim3(256), 0, e->stream, e->vol[vol], e->fgp_send_first, (int)e->npix, e->sx, 0);
        LAUNCHCHK();
        if (e->fgp_planes2 && e->nx >= 2) {      // the deep set: A of slice 1 (send_first plane 4) and of the last slice (send_last plane 1)
            hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)((e->npix + 255) / 256)), dim3(256), 0, e->stream, e->vol[vol], e->fgp_send_first + 4 * e->npix, (int)e->npix, e->sx, 1);
            hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)((e->npix + 255) / 256)), dim3(256), 0, e->stream, e->vol[vol], e->fgp_send_last + e->npix, (int)e->npix, e->sx, e->nx - 1);
            LAUNCHCHK();
        }
    }
    return TOMO_OK;
}

int tomo_fgp_fused_step(tomo_engine *e, float lambda, int first_iteration)
{
    NEED(e);
    if (!e->fgp_q[2] || !e->fgp_p[2]) return fail(TOMO_ERR_STATE, "tomo_fgp_fused_begin has not been called");
    const int yseg = 32;
    const int nzb = (e->n + TVL_TZ - 1) / TVL_TZ, nys = (e->n + yseg - 1) / yseg, nchunk = e->sxc / 64;
    // one workgroup per item; an XCD-aligned grid when the z blocks split evenly over the 8 XCDs (k_fgp_fused's item map)
    dim3 grid((nzb & 7) == 0 ? 8u * (unsigned)((nzb >> 3) * nchunk * nys) : (unsigned)(nzb * nchunk * nys));
    const float multip = 1.0f / (26.0f * lambda);
    FgpEdge ed{};
    ed.first = e->is_first; ed.last = e->is_last;
    if (fgp_sharded(e)) { ed.p1_lo = e->fgp_lo; ed.hi = e->fgp_hi; ed.send_first = e->fgp_send_first; ed.send_last = e->fgp_send_last; }
    {
        ProfScope ps(e, TOMO_K_FGP_GRAD);
        if (fgp_sharded(e))
            hipLaunchKernelGGL(k_fgp_fused<true>, grid, dim3(256), 0, e->stream, e->vol[e->fgp_target], e->fgp_p[0], e->fgp_p[1], e->fgp_p[2],
                               e->fgp_q[0], e->fgp_q[1], e->fgp_q[2], lambda, multip, e->n, e->nx, e->sx, yseg, first_iteration ? 1 : 0, ed);
        else
            hipLaunchKernelGGL(k_fgp_fused<false>, grid, dim3(256), 0, e->stream, e->vol[e->fgp_target], e->fgp_p[0], e->fgp_p[1], e->fgp_p[2],
                               e->fgp_q[0], e->fgp_q[1], e->fgp_q[2], lambda, multip, e->n, e->nx, e->sx, yseg, first_iteration ? 1 : 0, ed);
    }
    LAUNCHCHK();
    for (int k = 0; k < 3; ++k) std::swap(e->fgp_p[k], e->fgp_q[k]);
    return TOMO_OK;
}

// two iterations in one pass (k_fgp_fused2: P stays on chip between them).  A sharded slab needs the two-slice-deep planes
// (tomo_bind_fgp_halo2, exchanged once before the call: tomo_comm_fgp_exchange2) and at least two slices on EVERY slab of the ring.
int tomo_fgp_fused_step2(tomo_engine *e, float lambda, int first_iteration)
{
    NEED(e);
    if (!e->fgp_q[2] || !e->fgp_p[2]) return fail(TOMO_ERR_STATE, "tomo_fgp_fused_begin has not been called");
    if (fgp_sharded(e) && (!e->fgp_planes2 || !e->fgp_lo)) return fail(TOMO_ERR_STATE, "a sharded tomo_fgp_fused_step2 needs the two-slice-deep planes (tomo_bind_fgp_halo2)");
    if (fgp_sharded(e) && e->nx < 2) return fail(TOMO_ERR_STATE, "a sharded tomo_fgp_fused_step2 needs at least two slices per slab");
    const int yseg = 32;                                 // (16 ... 64 within 3 %; 128 and more lose to the tail)
    const int nzb = (e->n + F2_TZ - 1) / F2_TZ, nys = (e->n + yseg - 1) / yseg, nchunk = (e->nx + F2_SC - 1) / F2_SC;
    dim3 grid((nzb & 7) == 0 ? 8u * (unsigned)((nzb >> 3) * nchunk * nys) : (unsigned)(nzb * nchunk * nys));
    const float multip = 1.0f / (26.0f * lambda);
    {
        ProfScope ps(e, TOMO_K_FGP_GRAD);
        if (fgp_sharded(e)) {
            Fgp2Edge ed{e->fgp_lo, e->fgp_hi, e->fgp_send_first, e->fgp_send_last, e->is_first, e->is_last};
            hipLaunchKernelGGL((k_fgp_fused2<false, true>), grid, dim3(256), 0, e->stream, e->vol[e->fgp_target], e->fgp_p[0], e->fgp_p[1], e->fgp_p[2],
                               e->fgp_q[0], e->fgp_q[1], e->fgp_q[2], lambda, multip, e->n, e->nx, e->sx, yseg, first_iteration ? 1 : 0, ed);
        } else
        hipLaunchKernelGGL((k_fgp_fused2<false, false>), grid, dim3(256), 0, e->stream, e->vol[e->fgp_target], e->fgp_p[0], e->fgp_p[1], e->fgp_p[2],
                           e->fgp_q[0], e->fgp_q[1], e->fgp_q[2], lambda, multip, e->n, e->nx, e->sx, yseg, first_iteration ? 1 : 0, Fgp2Edge{});
    }
    LAUNCHCHK();
    for (int k = 0; k < 3; ++k) std::swap(e->fgp_p[k], e->fgp_q[k]);
    return TOMO_OK;
}

// one more iteration AND the call's result in one pass (k_fgp_fused2<true>): D of the iteration's P, which is never stored; the
// result is written to a scratch volume that then changes places with the target's buffer.  Whole-volume slabs only.
int tomo_fgp_fused_last(tomo_engine *e, float lambda, int first_iteration)
{
    NEED(e);
    if (!e->fgp_q[2] || !e->fgp_p[2]) return fail(TOMO_ERR_STATE, "tomo_fgp_fused_begin has not been called");
    if (fgp_sharded(e)) return fail(TOMO_ERR_STATE, "tomo_fgp_fused_last is for a slab that is the whole volume");
    if (e->fgp_target < 0 || e->fgp_target >= TOMO_VOL_SLOTS) return fail(TOMO_ERR_STATE, "no target volume");
    const int yseg = 32;
    const int nzb = (e->n + F2_TZ - 1) / F2_TZ, nys = (e->n + yseg - 1) / yseg, nchunk = (e->nx + F2_SC - 1) / F2_SC;
    dim3 grid((nzb & 7) == 0 ? 8u * (unsigned)((nzb >> 3) * nchunk * nys) : (unsigned)(nzb * nchunk * nys));
    const float multip = 1.0f / (26.0f * lambda);
    {
        ProfScope ps(e, TOMO_K_FGP_OBJ);
        hipLaunchKernelGGL((k_fgp_fused2<true, false>), grid, dim3(256), 0, e->stream, e->vol[e->fgp_target], e->fgp_p[0], e->fgp_p[1], e->fgp_p[2],
                           e->fgp_q[0], e->fgp_q[1], e->fgp_q[2], lambda, multip, e->n, e->nx, e->sx, yseg, first_iteration ? 1 : 0, Fgp2Edge{});
    }
    LAUNCHCHK();
    std::swap(e->vol[e->fgp_target], e->fgp_q[0]);         // the prox result's buffer becomes the volume; the old one is scratch now
    ++e->vol_version[e->fgp_target];
    return TOMO_OK;
}

// the last iteration only needs D (tv_fgp.cu:272), written straight over the target volume
int tomo_fgp_fused_end(tomo_engine *e, float lambda)
{
    if (e && e->fgp_target >= 0 && e->fgp_target < TOMO_VOL_SLOTS) ++e->vol_version[e->fgp_target];   // the prox result lands in the volume

    NEED(e);
    if (!e->fgp_p[2]) return fail(TOMO_ERR_STATE, "tomo_fgp_fused_begin has not been called");
    ProfScope ps(e, TOMO_K_FGP_OBJ);
    float *a = e->vol[e->fgp_target];
    hipLaunchKernelGGL(k_fgp_obj, dim3(tv_grid(e)), dim3(256), 0, e->stream, a, a, e->fgp_p[0], e->fgp_p[1], e->fgp_p[2],
                       fgp_sharded(e) ? e->fgp_lo : e->halo_lo, e->is_first, lambda, e->n, e->nx, e->sx);
    LAUNCHCHK();
    return TOMO_OK;
}

int tomo_tv_fgp_vol(tomo_engine *e, int vol, int iters, float lambda)
{
    int rc;
    if ((rc = tomo_tv(e, vol, 1e-6f))) return rc;    // tv_fgp.cu:170-189,231-238
    const bool fused = e && e->fgp_fused && iters > 1;
    int f = e->is_first, l = e->is_last;
    e->is_first = e->is_last = 1;
    if (fused) {
        // iterations 0..iters-2: one fused kernel each (D stays on chip); the last iteration only needs D
        rc = tomo_fgp_fused_begin(e, vol);
        int i = 0;
        if (e->fgp_pair) for (; i + 2 < iters && !rc; i += 2) rc = tomo_fgp_fused_step2(e, lambda, i == 0);     // pairs, P kept on chip between
        if (e->fgp_pair && i + 2 == iters) {              // an odd iteration out and the result: one pass
            if (!rc) rc = tomo_fgp_fused_last(e, lambda, i == 0);
        } else {
            for (; i + 1 < iters && !rc; ++i) rc = tomo_fgp_fused_step(e, lambda, i == 0);
            if (!rc) rc = tomo_fgp_fused_end(e, lambda);
        }
        e->is_first = f; e->is_last = l;
        return rc;
    }
    rc = fgp_begin_impl(e, vol, true);
    for (int i = 0; i < iters && !rc; ++i) {
        if ((rc = tomo_fgp_obj(e, lambda)) || (rc = tomo_fgp_grad(e, lambda))) break;
    }
    e->is_first = f; e->is_last = l;
    if (rc) return rc;
    return tomo_fgp_end(e, iters);
}


// ---- native communicator: RCCL on the engine's own stream --------------------------------------------------------------------
// The slab-sharded path needs, per ASD-POCS iteration: one ring exchange of halo planes, ten rounds of {all-reduce of ||g||^2 +
// the gradient's boundary planes to the two neighbours}, and one all-reduce of the iteration's scalars (mpi_ctvlib.cpp:400-422
// ring, :455,:547 MPI_Allreduce).  Through torch.distributed each of these is two collectives issued from Python on RCCL's own
// stream with an event hop in and out: measured on a world-1 group ~100 us per round, 1.1 ms of a 5.3 ms step on the 64-slice
// slab of an 8-GPU strong-scaling run.  Here a round is ONE ncclGroup (all-reduce + 2 sends + 2 receives = one RCCL kernel)
// enqueued by this library on the engine's stream, so a whole sharded tv_gd is one C call with every launch stream-ordered and
// no Python, no second stream, no event hop in between.  It also gives a C / C++ host a way to shard (the C ABI had
// no communicator entry).  librccl is opened with dlopen on first use (the copy already in the process if there is one -- torch
// ships its own), so single-GPU users never load it.
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    int version = 0;            // ncclGetVersion's code: major * 10000 + minor * 100 + patch (2.9 and later)
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return TOMO_OK;
    void *h = nullptr;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *nm : names) if (!h) h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);     // the copy already in the process, if any
    for (const char *nm : names) if (!h) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(TOMO_ERR_STATE, std::string("librccl not found: ") + dlerror());
#define RCCL_SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return fail(TOMO_ERR_STATE, "librccl lacks " name); } while (0)
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); RCCL_SYM(CommInitRank, "ncclCommInitRank"); RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(AllReduce, "ncclAllReduce"); RCCL_SYM(Send, "ncclSend"); RCCL_SYM(Recv, "ncclRecv");
    RCCL_SYM(GroupStart, "ncclGroupStart"); RCCL_SYM(GroupEnd, "ncclGroupEnd"); RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    // a round here is ONE group of an all-reduce with point-to-point sends / receives: grouped ncclSend / ncclRecv exist since 2.7
    // (codes below 10000 are the old major * 1000 + minor * 100 scheme, i.e. older than 2.9; built and verified against 2.27.7, ROCm 7.2.0)
    if (auto getv = (ncclResult_t (*)(int *))dlsym(h, "ncclGetVersion")) { int v = 0; if (getv(&v) == ncclSuccess) g_rccl.version = v; }
    if (g_rccl.version && g_rccl.version < 2700) return fail(TOMO_ERR_STATE, "librccl is older than 2.7 (no grouped send / receive): version code " + std::to_string(g_rccl.version));
    g_rccl.lib = h;
    return TOMO_OK;
}
#define NCCLCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail(TOMO_ERR_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); } while (0)

struct CommRef {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    std::atomic<int> refs{1};
};

static void comm_release(tomo_engine *e)
{
    if (!e->comm) return;
    if (e->comm->refs.fetch_sub(1) == 1) {
        if (e->comm->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(e->comm->comm);
        delete e->comm;
    }
    e->comm = nullptr;
    if (e->comm_fgp && e->fgp_lo == e->comm_fgp) { e->fgp_lo = e->fgp_hi = e->fgp_send_first = e->fgp_send_last = nullptr; e->fgp_planes2 = false; }
    void *ptrs[] = {e->comm_send_first, e->comm_send_last, e->comm_g_lo, e->comm_g_hi, e->comm_scal, e->comm_fgp};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    e->comm_fgp = nullptr;
    e->comm_send_first = e->comm_send_last = e->comm_g_lo = e->comm_g_hi = nullptr;
    e->comm_scal = nullptr;
}

static int comm_buffers(tomo_engine *e)
{
    if (e->comm_scal) return TOMO_OK;
    int rc;
    float **planes[] = {&e->comm_send_first, &e->comm_send_last, &e->comm_g_lo, &e->comm_g_hi};
    for (float **p : planes) if ((rc = dev_alloc((void **)p, e->npix * sizeof(float), true, e->stream))) return rc;
    if (!e->fgp_lo) {        // a host that binds no planes of its own (tomo_bind_fgp_halo / _halo2) gets the engine's, the deep set: lo 5, hi 8, send_first 8, send_last 5
        if ((rc = dev_alloc((void **)&e->comm_fgp, 26 * e->npix * sizeof(float), true, e->stream))) return rc;
        e->fgp_lo = e->comm_fgp; e->fgp_hi = e->comm_fgp + 5 * e->npix; e->fgp_send_first = e->comm_fgp + 13 * e->npix; e->fgp_send_last = e->comm_fgp + 21 * e->npix;
        e->fgp_planes2 = true;
    }
    return dev_alloc((void **)&e->comm_scal, TOMO_S_COUNT * sizeof(double), true, e->stream);
}

// ring exchange inside an open group: my last plane(s) -> next's lo, my first plane(s) -> prev's hi.  With prev == next (two
// ranks) the two messages to the one peer match in posting order on both sides; with one rank they are self-sends.
static ncclResult_t comm_ring(tomo_engine *e, const float *first, size_t nfirst, const float *last, size_t nlast, float *lo, float *hi)
{
    const CommRef *c = e->comm;
    const int nxt = (c->rank + 1) % c->world, prv = (c->rank + c->world - 1) % c->world;
    // (inside an open group: no early return -- the caller must reach ncclGroupEnd whatever happens, or every later RCCL call of this
    // thread, torch.distributed's included, is queued into a group that never closes)
    ncclResult_t r[4] = {g_rccl.Send(last, nlast, ncclFloat32, nxt, c->comm, e->stream),
                         g_rccl.Send(first, nfirst, ncclFloat32, prv, c->comm, e->stream),
                         g_rccl.Recv(lo, nlast, ncclFloat32, prv, c->comm, e->stream),
                         g_rccl.Recv(hi, nfirst, ncclFloat32, nxt, c->comm, e->stream)};
    for (ncclResult_t x : r) if (x != ncclSuccess) return x;
    return ncclSuccess;
}
// one group around `body` (which returns the first failing ncclResult_t of what it enqueued): GroupEnd is always reached
static int comm_group(tomo_engine *e, const char *what, const std::function<ncclResult_t()> &body)
{
    ++e->comm_rounds;
    NCCLCHK(g_rccl.GroupStart());
    const ncclResult_t r = body();
    const ncclResult_t rend = g_rccl.GroupEnd();
    if (r != ncclSuccess) return fail(TOMO_ERR_HIP, std::string(what) + ": " + g_rccl.GetErrorString(r));
    if (rend != ncclSuccess) return fail(TOMO_ERR_HIP, std::string(what) + " (ncclGroupEnd): " + g_rccl.GetErrorString(rend));
    return TOMO_OK;
}
#define NEED_COMM(e) do { NEED(e); if (!(e)->comm) return fail(TOMO_ERR_STATE, "engine has no communicator (tomo_comm_init)"); { int rc_ = comm_buffers(e); if (rc_) return rc_; } } while (0)

int tomo_comm_unique_id(void *id128)
{
    if (!id128) return fail(TOMO_ERR_ARG, "null id buffer");
    int rc = rccl_load(); if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return TOMO_OK;
}

int tomo_comm_init(tomo_engine *e, const void *id128, int world, int rank)
{
    NEED(e);
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(TOMO_ERR_ARG, "bad communicator arguments");
    int rc = rccl_load(); if (rc) return rc;
    comm_release(e);
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    CommRef *c = new CommRef();
    c->world = world; c->rank = rank;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);      // collective: every rank of the group calls it
    if (r != ncclSuccess) { delete c; return fail(TOMO_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
    e->comm = c;
    e->is_first = rank == 0; e->is_last = rank == world - 1;
    return TOMO_OK;
}

int tomo_comm_share(tomo_engine *e, tomo_engine *other)
{
    NEED(e);
    if (!other || !other->comm) return fail(TOMO_ERR_ARG, "the other engine has no communicator");
    if (other->device != e->device) return fail(TOMO_ERR_ARG, "engines on different devices cannot share a communicator");
    if (e->comm == other->comm) return TOMO_OK;
    comm_release(e);
    other->comm->refs.fetch_add(1);
    e->comm = other->comm;
    e->is_first = e->comm->rank == 0; e->is_last = e->comm->rank == e->comm->world - 1;
    return TOMO_OK;
}

int tomo_comm_destroy(tomo_engine *e) { if (!e) return fail(TOMO_ERR_ARG, "null engine"); (void)hipSetDevice(e->device); if (e->stream) (void)hipStreamSynchronize(e->stream); comm_release(e); return TOMO_OK; }

int tomo_comm_info(tomo_engine *e, int *world, int *rank)
{
    if (!e) return fail(TOMO_ERR_ARG, "null engine");
    if (world) *world = e->comm ? e->comm->world : 0;
    if (rank) *rank = e->comm ? e->comm->rank : 0;
    return TOMO_OK;
}

// the field's boundary slices to the ring neighbours' halo planes (before a stencil pass): pack + one group
int tomo_comm_exchange_halo(tomo_engine *e, int field)
{
    NEED_COMM(e);
    int rc;
    if ((rc = tomo_halo_pack_both(e, field, e->comm_send_first, e->comm_send_last))) return rc;
    return comm_group(e, "halo exchange", [&] { return comm_ring(e, e->comm_send_first, (size_t)e->npix, e->comm_send_last, (size_t)e->npix, e->halo_lo, e->halo_hi); });
}

// all slots of the scalar buffer summed over the ranks into a COPY (the buffer itself keeps this slab's partial sums), read back:
// blocking form and the snapshot form of tomo_scalars_snapshot (collected by tomo_scalars_snapshot_read)
static int comm_sum_scalars(tomo_engine *e)
{
    { int rc = tomo_async_wait(e); if (rc) return rc; }
    ++e->comm_rounds;       // (out of place: the copy into comm_scal that an in-place all-reduce needed is one launch less per step)
    NCCLCHK(g_rccl.AllReduce(e->d_scal, e->comm_scal, TOMO_S_COUNT, ncclFloat64, ncclSum, e->comm->comm, e->stream));
    return TOMO_OK;
}

int tomo_comm_read_scalars(tomo_engine *e, double *out, int count)
{
    NEED_COMM(e);
    if (!out || count < 0 || count > TOMO_S_COUNT) return fail(TOMO_ERR_ARG, "bad scalar count");
    int rc = comm_sum_scalars(e); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, e->comm_scal, count * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return TOMO_OK;
}

int tomo_comm_scalars_snapshot(tomo_engine *e)
{
    NEED_COMM(e);
    int rc = comm_sum_scalars(e); if (rc) return rc;
    if (!e->h_snap) {
        HIPCHK(hipHostMalloc((void **)&e->h_snap, TOMO_S_COUNT * sizeof(double), hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&e->ev_snap, hipEventDisableTiming));
    }
    hipLaunchKernelGGL(k_scalars_to_host, dim3(1), dim3(64), 0, e->stream, (const double *)e->comm_scal, e->h_snap, (int)TOMO_S_COUNT);
    LAUNCHCHK();
    HIPCHK(hipEventRecord(e->ev_snap, e->stream));
    e->snap_pending = true;
    return TOMO_OK;
}

// slab-sharded tv_gd, whole call: the halo planes once, then per inner iteration the norm pass, ONE group {all-reduce of sum g^2
// in place + the gradient's boundary planes round the ring}, the update pass and the halo planes advanced locally
// (engine.py: _tv_descent_one_round is the same protocol over torch.distributed; bit-identical results).  TV before descent
// stays in TOMO_S_TV as this slab's partial sum; track_vol < 0: plain tv_gd.
int tomo_comm_tv_gd(tomo_engine *e, int ng, float dPOCS, float eps, int track_vol, int slot)
{
    NEED_COMM(e);
    int rc;
    if ((rc = tomo_comm_exchange_halo(e, e->tv_target))) return rc;
    if (ng <= 0) {
        if ((rc = tomo_tv_partial(e, e->tv_target, eps)) || (rc = tomo_positivity(e, e->tv_target))) return rc;
        if (track_vol >= 0) {
            if ((rc = tomo_diff_norm_sq(e, e->tv_target, track_vol, slot))) return rc;
            return tomo_copy_volume(e, track_vol, e->tv_target);
        }
        return TOMO_OK;
    }
    for (int g = 0; g < ng; ++g) {
        if ((rc = tomo_tv_grad_planes(e, eps, g == 0, e->comm_send_first, e->comm_send_last))) return rc;
        // the global sum g^2 lands in comm_scal[GNORM]; TOMO_S_GNORM itself keeps this slab's partial sum, so a later
        // tomo_comm_read_scalars (which sums every slot over the ranks) returns the global norm once, not world times
        rc = comm_group(e, "tv_gd round", [&] {
            ncclResult_t r = g_rccl.AllReduce(e->d_scal + TOMO_S_GNORM, e->comm_scal + TOMO_S_GNORM, 1, ncclFloat64, ncclSum, e->comm->comm, e->stream);
            ncclResult_t r2 = comm_ring(e, e->comm_send_first, (size_t)e->npix, e->comm_send_last, (size_t)e->npix, e->comm_g_lo, e->comm_g_hi);
            return r != ncclSuccess ? r : r2; });
        if (rc) return rc;
        e->gnorm_override = e->comm_scal + TOMO_S_GNORM;
        if (g == ng - 1) {
            rc = track_vol >= 0 ? tomo_tv_update_tracked(e, dPOCS, 1, track_vol, slot) : tomo_tv_update(e, dPOCS, 1);
        } else {
            // the update reads the old halo planes, which then follow the neighbours (inside the pass: a second pair of planes)
            rc = tv_update_impl(e, dPOCS, 0, -1, 0, false, nullptr, nullptr, e->comm_g_lo, e->comm_g_hi);
        }
        e->gnorm_override = nullptr;
        if (rc) return rc;
    }
    return TOMO_OK;
}

// the exchange between two fused FGP iterations: send_last (P1 of my last slice) -> next's lo, send_first (A, P1, P2, P3 of my
// first slice) -> prev's hi (tomo_bind_fgp_halo names the four buffers)
int tomo_comm_fgp_exchange(tomo_engine *e)
{
    NEED_COMM(e);
    if (!e->fgp_lo) return fail(TOMO_ERR_STATE, "slab-sharded fused FGP needs tomo_bind_fgp_halo");
    return comm_group(e, "fgp exchange", [&] { return comm_ring(e, e->fgp_send_first, 4 * (size_t)e->npix, e->fgp_send_last, (size_t)e->npix, e->fgp_lo, e->fgp_hi); });
}

// ... and between two PAIRS of fused iterations (tomo_fgp_fused_step2 on slabs): the two-slice-deep planes, one exchange per two
// iterations: send_last (5 planes) -> next's lo, send_first (8 planes) -> prev's hi
int tomo_comm_fgp_exchange2(tomo_engine *e)
{
    NEED_COMM(e);
    if (!e->fgp_lo || !e->fgp_planes2) return fail(TOMO_ERR_STATE, "the two-deep FGP exchange needs the two-slice-deep planes (tomo_bind_fgp_halo2)");
    return comm_group(e, "fgp exchange (two deep)", [&] { return comm_ring(e, e->fgp_send_first, 8 * (size_t)e->npix, e->fgp_send_last, 5 * (size_t)e->npix, e->fgp_lo, e->fgp_hi); });
}

int tomo_get_option(tomo_engine *e, const char *name, int *value)
{
    if (!e || !name || !value) return fail(TOMO_ERR_ARG, "null argument");
    if (std::strcmp(name, "fp_strip") == 0) { *value = e->fp_strip; return TOMO_OK; }
    if (std::strcmp(name, "fp_strip_ready") == 0) { *value = e->fs_ok ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "fp_strip_slots") == 0) { *value = e->fs_ok ? e->fs_kused : 0; return TOMO_OK; }
    if (std::strcmp(name, "fp_tile") == 0) { *value = e->fp_tile; return TOMO_OK; }
    if (std::strcmp(name, "bp_tile") == 0) { *value = e->bp_tile; return TOMO_OK; }
    if (std::strcmp(name, "bp_list") == 0) { *value = e->bp_list; return TOMO_OK; }
    if (std::strcmp(name, "fgp_pair") == 0) { *value = e->fgp_pair; return TOMO_OK; }
    if (std::strcmp(name, "fp_list") == 0) { *value = e->fp_list; return TOMO_OK; }
    if (std::strcmp(name, "fp_list_ready") == 0) { *value = e->fl_ok ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "bp_list_ready") == 0) { *value = e->bl_ok ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "fp_reuse") == 0) { *value = e->fp_reuse; return TOMO_OK; }
    if (std::strcmp(name, "sart_tile") == 0) { *value = e->sart_tile; return TOMO_OK; }
    if (std::strcmp(name, "sart_resident") == 0) { *value = e->sart_resident; return TOMO_OK; }
    if (std::strcmp(name, "sart_resident_ready") == 0) { *value = e->rs_ok ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "sart_resident_active") == 0) { *value = (e->rs_ok && e->sart_resident != 0) ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "sart_resident_spin") == 0) { *value = (int)std::min<uint32_t>(e->rs_spin_limit, 0x7FFFFFFFu); return TOMO_OK; }
    if (std::strcmp(name, "sart_resident_fallbacks") == 0) { *value = e->rs_fallbacks; return TOMO_OK; }            // sweeps that needed the streamed chain
    if (std::strcmp(name, "sart_resident_fallback_chunks") == 0) { *value = e->rs_fallback_chunks; return TOMO_OK; } // ... and the 64-slice chunks it swept
    if (std::strcmp(name, "sart_resident_skip") == 0) { *value = e->rs_skip; return TOMO_OK; }                       // sweeps the resident form still sits out
    if (std::strcmp(name, "sart_resident_last_code") == 0) { *value = e->rs_last_code; return TOMO_OK; }             // 1 residual rows, 2 tile sums (0: the commit)
    if (std::strcmp(name, "table_kib") == 0) { *value = (int)std::min<size_t>(e->table_bytes >> 10, 0x7FFFFFFF); return TOMO_OK; }   // device tables built at creation
    if (std::strcmp(name, "create_ms") == 0) { *value = (int)std::min(e->create_ms + 0.5, 2147483647.0); return TOMO_OK; }           // what creating this engine took
    if (std::strcmp(name, "form_fp") == 0) { *value = select_forms(e).fp; return TOMO_OK; }
    if (std::strcmp(name, "form_bp") == 0) { *value = select_forms(e).bp; return TOMO_OK; }
    if (std::strcmp(name, "form_sart") == 0) { *value = select_forms(e).sart; return TOMO_OK; }
    if (std::strcmp(name, "rccl_version") == 0) { *value = g_rccl.version; return TOMO_OK; }     // 0 until a communicator has been opened
    if (std::strcmp(name, "comm_rounds") == 0) { *value = (int)std::min<int64_t>(e->comm_rounds, 0x7FFFFFFF); return TOMO_OK; }
    return fail(TOMO_ERR_ARG, std::string("unknown option ") + name);
}

int tomo_set_option(tomo_engine *e, const char *name, int value)
{
    if (!e || !name) return fail(TOMO_ERR_ARG, "null argument");
    if (std::strcmp(name, "fgp_fused") == 0) { e->fgp_fused = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "fgp_pair") == 0) { e->fgp_pair = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "sart_fused") == 0) { e->sart_fused = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "fp_all_lpr") == 0) { e->fp_all_lpr = (value == 16 || value == 32) ? value : 0; return TOMO_OK; }
    if (std::strcmp(name, "art_chain") == 0) { e->art_chain = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "sart_tile") == 0) { e->sart_tile = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "sart_resident") == 0) { e->sart_resident = value < 0 ? -1 : (value ? 1 : 0); e->rs_skip = e->rs_backoff = 0; return TOMO_OK; }
    if (std::strcmp(name, "sart_resident_spin") == 0) { e->rs_spin_limit = value < 0 ? (1u << 21) : (uint32_t)value; return TOMO_OK; }   // polls before a wait gives up (< 0: the default; tests: tiny, 0 = at the first look)
    if (std::strcmp(name, "sart_resident_test_fail") == 0) { e->rs_test_fail = std::max(0, value); return TOMO_OK; }   // tests: chunk + 1 that refuses to commit
    if (std::strcmp(name, "fp_reuse") == 0) { e->fp_reuse = value != 0; g_clear(e); e->yk_claim.valid = false; return TOMO_OK; }
    if (std::strcmp(name, "fp_tile_pipe") == 0) { e->fp_tile_pipe = std::max(0, value); return TOMO_OK; }
    if (std::strcmp(name, "sart_streams") == 0) { e->sart_streams = value >= 2 ? std::min(value, (int)tomo_engine::MAX_CHAINS) : (value == 1 ? 1 : 0); return TOMO_OK; }
    if (std::strcmp(name, "art_tile") == 0) { e->art_tile = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "sart_skip_same") == 0) { e->sart_skip_same = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "sart_nt") == 0) { e->sart_nt = value < 0 ? -1 : (value ? 1 : 0); return TOMO_OK; }
    if (std::strcmp(name, "sart_coop") == 0) { e->sart_coop = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "sart_coop_spin") == 0) { e->sart_coop_spin = value < 0 ? -1 : value; return TOMO_OK; }
    if (std::strcmp(name, "bp_tile") == 0) { e->bp_tile = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "bp_list") == 0) { e->bp_list = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "bp_list_band") == 0) { e->bp_list_band = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "fp_list") == 0) { e->fp_list = value ? 1 : 0; return TOMO_OK; }
    // all-angle FP form: "fp_strip" = 1 (default) sheared strips; asking for "fp_tile" = 1 / 0 explicitly selects the tile-stationary /
    // the ray-driven form (and takes the strips out of the way until "fp_strip" = 1 is set again)
    if (std::strcmp(name, "fp_strip") == 0) { e->fp_strip = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "fp_tile") == 0) { e->fp_tile = value ? 1 : 0; e->fp_strip = 0; return TOMO_OK; }
    if (std::strcmp(name, "fp_tile_chunks_per_pass") == 0) {   // any count >= 1 (0 = from the scratch cap); before the first projection
        if (value < 0 || e->ft_part || e->ft_part_aux || e->fs_part || e->fs_part_aux || e->fl_part || e->fl_part_aux) return fail(TOMO_ERR_STATE, "fp_tile_chunks_per_pass must be set before the first projection");
        e->ft_ncp_forced = value; e->ft_ncp = 0; e->fs_ncp = 0; e->fl_ncp = 0; return TOMO_OK;
    }
    if (std::strcmp(name, "fp_tile_scratch_mib") == 0) {   // cap of the partial-sum scratch; takes effect before the first all-angle FP
        if (value <= 0 || e->ft_part || e->ft_part_aux || e->fs_part || e->fs_part_aux || e->fl_part || e->fl_part_aux) return fail(TOMO_ERR_STATE, "fp_tile_scratch_mib must be positive and set before the first projection");
        e->ft_scratch_cap = (size_t)value << 20; e->ft_ncp = 0; e->fs_ncp = 0; e->fl_ncp = 0; return TOMO_OK;
    }
    if (std::strcmp(name, "tv_gnorm_slot") == 0) {
        if (value < 0 || value >= TOMO_S_COUNT) return fail(TOMO_ERR_ARG, "bad scalar slot");
        e->gnorm_slot = value; return TOMO_OK;
    }
    if (std::strcmp(name, "tv_tz") == 0) { e->tv_tz = value == 4 ? 4 : 8; return TOMO_OK; }
    if (std::strcmp(name, "tv_halo_fold") == 0) { e->tv_halo_fold = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "tv_march4") == 0) { e->tv_march4 = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "tv_yseg") == 0) { e->tv_yseg = value < 0 ? 0 : value; return TOMO_OK; }
    if (std::strcmp(name, "tv_recompute") == 0) { e->tv_recompute = value ? 1 : 0; return TOMO_OK; }
    if (std::strcmp(name, "tv_lds") == 0) { e->tv_lds = value; return TOMO_OK; }   // 1 register march, 8 / 16 LDS march (z-columns per workgroup), 0 direct
    return fail(TOMO_ERR_ARG, std::string("unknown option ") + name);
}

// ---- measurement ------------------------------------------------------------------------------------------------------------
int tomo_profile_enable(tomo_engine *e, int kernel, int on)
{
    NEED(e);
    if (kernel < 0 || kernel >= PROF_MAX_KERNELS) return fail(TOMO_ERR_ARG, "bad kernel id");
    HIPCHK(hipStreamSynchronize(e->stream));
    e->prof[kernel].on = on != 0;
    e->prof[kernel].stride = on > 1 ? (unsigned)on : 1u;     // on = N > 1: bracket every N-th launch only
    e->prof[kernel].seen = 0;
    e->prof[kernel].used = 0;
    e->prof[kernel].dropped = 0;
    if (on) {
        if (!e->prof[kernel].ref) HIPCHK(hipEventCreate(&e->prof[kernel].ref));
        HIPCHK(hipEventRecord(e->prof[kernel].ref, e->stream));
    }
    return TOMO_OK;
}

// busy_ms (may be null): time during which AT LEAST ONE launch of the kernel was executing (union of the launch intervals
// on the common time base) -- with the SART sweep on two streams two launches of one kernel overlap, and sum / launches
// is then the duration of a launch that shares the chip, not the chip's rate
int tomo_profile_read2(tomo_engine *e, int kernel, int64_t *launches, double *total_ms, double *busy_ms)
{
    NEED(e);
    if (kernel < 0 || kernel >= PROF_MAX_KERNELS || !launches || !total_ms) return fail(TOMO_ERR_ARG, "bad argument");
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int u = 0; u < tomo_engine::MAX_CHAINS; ++u) if (e->sub_stream[u]) HIPCHK(hipStreamSynchronize(e->sub_stream[u]));
    if (e->aux) HIPCHK(hipStreamSynchronize(e->aux));
    ProfSlot &p = e->prof[kernel];
    double tot = 0;
    std::vector<std::pair<float, float>> iv;
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
        tot += ms;
        if (busy_ms && p.ref) {
            float t0 = 0;
            HIPCHK(hipEventElapsedTime(&t0, p.ref, p.ev[i]));
            iv.emplace_back(t0, t0 + ms);
        }
    }
    if (busy_ms) {
        std::sort(iv.begin(), iv.end());
        double busy = 0; float cur_a = 0, cur_b = -1;
        for (auto &q : iv) {
            if (cur_b < cur_a || q.first > cur_b) { if (cur_b >= cur_a) busy += cur_b - cur_a; cur_a = q.first; cur_b = q.second; }
            else cur_b = std::max(cur_b, q.second);
        }
        if (cur_b >= cur_a) busy += cur_b - cur_a;
        *busy_ms = busy;
    }
    *launches = (int64_t)(p.used / 2);
    *total_ms = tot;
    p.used = 0;
    if (p.dropped) {   // never report an average over a silently truncated log
        std::string msg = std::to_string(p.dropped) + " launches were not recorded (event log full)";
        p.dropped = 0;
        return fail(TOMO_ERR_STATE, msg);
    }
    return TOMO_OK;
}

// the launch intervals of a kernel [t0, t1) in ms since ref_engine's log of the same kernel was switched on (engines of a slab
// group share one device: their events are on one time base); does not reset the log
int tomo_profile_intervals(tomo_engine *e, int kernel, tomo_engine *ref_engine, double *t0, double *t1, int cap, int *count)
{
    NEED(e);
    if (kernel < 0 || kernel >= PROF_MAX_KERNELS || !ref_engine || !count) return fail(TOMO_ERR_ARG, "bad argument");
    HIPCHK(hipStreamSynchronize(e->stream));
    ProfSlot &p = e->prof[kernel];
    hipEvent_t ref = ref_engine->prof[kernel].ref;
    if (!ref) return fail(TOMO_ERR_STATE, "the reference engine's log is not enabled");
    int n = (int)(p.used / 2);
    *count = n;
    if (!t0 || !t1 || cap < n) return cap == 0 ? TOMO_OK : fail(TOMO_ERR_ARG, "interval buffers too small");
    for (int i = 0; i < n; ++i) {
        float a = 0, d = 0;
        HIPCHK(hipEventElapsedTime(&a, ref, p.ev[2 * i]));
        HIPCHK(hipEventElapsedTime(&d, p.ev[2 * i], p.ev[2 * i + 1]));
        t0[i] = a; t1[i] = a + d;
    }
    return TOMO_OK;
}

int tomo_profile_read(tomo_engine *e, int kernel, int64_t *launches, double *total_ms)
{
    return tomo_profile_read2(e, kernel, launches, total_ms, nullptr);
}

}  // extern "C"
