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
    // slab-sharded descent: the update pass advances the halo planes itself (lanes 1..8 of the packed edge registers: one 32-byte load of
    // the received gradient plane and one store per row and plane) instead of a k_halo_apply launch per inner iteration.  -1 = automatic:
    // where the slab has two chunks or more.  Measured (round 6, bench.py --force-dist, world-1 RCCL group): 128 slices 3.53 against 3.62 ms
    // per step; 64 slices (ONE chunk: every wave of the pass holds both planes) 2.04 against 2.01 -- so a one-chunk slab keeps the launch.
    // (The first form, one lane looping over the row's eight pixels: 2.165 against 2.012 at 64 slices.)  1 / 0 force it.
    int tv_halo_fold = -1;
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
    bool seg_ready = false;                       // the walk / segment tables above are on the device (built at creation or on first use)
    std::vector<double> angles;                   // the tilt angles of an engine created from angles (tomo_create): tables built on first use come from them
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

#include "engine_launch.inc"
#include "engine_create.inc"

extern "C" {

#include "engine_api.inc"
#include "engine_tv.inc"
#include "engine_comm.inc"
#include "engine_options.inc"

}  // extern "C"
