// sysmat.h -- host-side system matrix + derived tables (see sysmat.cpp).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace tomo {

struct Coo {  // CSR-shaped storage; "Coo" because rows may still be in path order before sort_rows()
    int64_t nrow = 0, ncol = 0;
    std::vector<int64_t> ptr;
    std::vector<uint32_t> col;
    std::vector<float> val;
};

struct Cell {  // the (at most two) rays of one angle that cross one pixel
    uint32_t r0; float w0; uint32_t r1; float w1;
};

struct Tables {
    std::vector<float> rowsum, rowinner, colsum_all;
    // rowcross[r] = A_r . A_{r+1} for two rays of one angle (0 for an angle's last ray).  art_chain_ok: every pixel's
    // two rays of an angle are neighbours (r1 = r0 + 1), so rays two apart share no pixel and one angle of a Kaczmarz
    // sweep is a forward projection + a scalar recurrence along the rays + a back-projection (k_art_chain).
    std::vector<float> rowcross;
    bool art_chain_ok = false;
    std::vector<Cell> cell;  // [P][N*N]
    float lipschitz = 0.f, lipschitz_cimmino = 0.f;
    // "walk" lists for the fused SART step: per ray, its matrix entries plus a share of the angle's
    // un-crossed pixels, so that the rays of one angle visit EVERY pixel and exactly one visit owns it.
    // entry = {pixel | owner << 31, weight bits}
    std::vector<uint32_t> walk_ptr;              // [P*N + 1]
    std::vector<uint32_t> walk_pix;
    std::vector<float> walk_w;
    // Equal-sized work items over the walk lists (one wave each): a ray is cut into segments of <= seg_len
    // visits.  Items of RAY_GROUP neighbouring rays are listed together and whole groups are dealt to the 8
    // XCD lists, all padded to the same length, so neighbours share an L2 and every XCD gets the same work.
    struct SegItem { uint32_t id, kbeg, kend, pad; };   // id = row_first[row] + segment number
    int seg_len = 0;
    std::vector<uint32_t> seg_exec_ptr;          // [P + 1] offsets (in items) into seg_exec; per angle 8*L items,
    std::vector<SegItem> seg_exec;               //   XCD-major: item of workgroup b sits at [ (b%8)*L + b/8 ]
    std::vector<uint32_t> row_first;             // [P*N] first partial-sum id of the row (within its angle)
    std::vector<uint32_t> row_nseg;              // [P*N]
    uint32_t max_items_per_angle = 0;
    // Tile-stationary all-angle projector: the matrix re-sorted by (image tile, row).  A "tile segment" is the part of
    // one ray inside one tile; its partial ray sum is computed from the tile held in LDS, the row's segments are then
    // summed in ascending tile order.  A tile's segments are dealt to TILE_SLOTS entry streams of equal length (one
    // per 16-lane group of the workgroup); a stream is a run of batches of TILE_BATCH entries, every segment padded
    // to whole batches with zero weights.  Entry = {byte offset of the pixel in the LDS tile image | last-batch-of-
    // segment flag << 31, weight}.  Partial-sum ids count segments in stream order.
    static constexpr int TILE_SLOTS = 64, TILE_BATCH = 8;
    int tile_ty = 0, tile_tz = 0, tiles_y = 0, tiles_z = 0;
    uint32_t tile_nseg = 0;
    std::vector<uint32_t> tile_slot_ptr;         // [ntiles*TILE_SLOTS + 1] first batch of a stream
    std::vector<uint32_t> tile_slot_seg0;        // [ntiles*TILE_SLOTS]     partial-sum id of its first segment
    std::vector<uint32_t> tile_off;              // [nbatch*TILE_BATCH]
    std::vector<float> tile_w;                   // [nbatch*TILE_BATCH]
    std::vector<uint32_t> rseg_ptr;              // [P*N + 1] segments of a row ...
    std::vector<uint32_t> rseg_idx;              // [nseg]    ... as partial-sum ids, ascending tile
    // Tile-stationary all-angle back-projector: per (tile, angle) the window of rays that cross the tile (staged in
    // LDS by the kernel) and the cell table regrouped by tile, ray numbers replaced by byte offsets into the staged
    // window (slot (angle % stage_angles), row (ray - first ray of the window)); weight 0 -> the zero row that ends a buffer.
    struct TileCell { uint32_t off0; float w0; uint32_t off1; float w1; };
    bool bp_tile_ok = false;                     // false: some window exceeds max_rows (kernel falls back)
    std::vector<uint32_t> bp_win;                // [ntiles * P]  first ray | rays << 16
    std::vector<TileCell> bp_cell;               // [(ntiles * P + pad) * TY*TZ], pixel order inside a tile: y-major
    // The same matrix as entry LISTS of the wave-per-pixel-block form (k_bp_list): a tile's pixels are dealt to `waves` waves in blocks
    // of 8 x 4 (bl_pixel); per (tile, stage of stage_angles angles, wave) the nonzero weights of the wave's pixels, angles ascending,
    // within an angle rows (rays) ascending -- a pixel's first ray is its lower one, so its sum keeps the order of k_bp_all -- as
    // PAIRS that share one read of their row:  {byte offset of the row in the staged windows (a multiple of row_bytes >= 256) |
    // register of pixel q0, weight 0, register of pixel q1, weight 1};  a row with an odd number of weights ends in a pair whose
    // second weight is 0 (into accumulator 0); lists are padded to whole batches of `batch` pairs the same way.  Stage s is staged in
    // LDS buffer s & 1.
    static constexpr int BL_TY = 16, BL_TZ = 16, BL_WAVES = 8, BL_A = 3, BL_MAXR = 26, BL_ROWB = 512, BL_BATCH = 8, BL_REGS = 2;   // the geometry k_bp_list is built for
    // pixel q (0..31) of wave w (0..7) inside the 16 x 16 tile: waves own 8 x 4 blocks (rows of a block share rays: 5.6 weights per ray)
    static void bl_pixel(int w, int q, int &ly, int &lz) { ly = (w >> 2) * 8 + (q >> 2); lz = (w & 3) * 4 + (q & 3); }
    bool bl_ok = false;                          // false: some window exceeds max_rows, or too many batches (the cell form stays)
    std::vector<uint32_t> bl_win;                // [ntiles * P]  first ray | rays << 16 (tiles of the LIST form's own size)
    std::unique_ptr<uint64_t[]> bl_ent;          // [(bl_nbatch + 1) * batch * 2] (one batch of padding behind the last: the kernel prefetches)
    uint64_t bl_nbatch = 0;
    std::vector<uint32_t> bl_ptr;                // [ntiles * nstage * waves + 1] first batch of a list
    // Per-angle tile tables of the fused SART step (k_sart_tile): tiles of st_ty x st_tz pixels, angle-major.
    //   st_cell[(i*ntiles + tile)*T*T + pixel]  rays of angle i through the pixel as byte offsets into the tile's
    //                                           staged ray window (zero row = row st_maxr)
    //   st_win[i*ntiles + tile]                 first ray | rays << 16 of that window
    //   st_seg[(i*ntiles + tile)*ST_MAXSEG + k] {first batch, batches} of the k-th ray segment of angle i in the tile
    //   st_segid[same index]                    its partial-sum id within the angle: a row's segments (ascending tile)
    //                                           have consecutive ids, st_row_first[row] + 0 .. st_row_nseg[row] - 1
    //   st_off / st_w                           entry batches (TILE_BATCH entries, zero-weight padding -> zero pixel)
    static constexpr int ST_MAXSEG = 32;
    bool st_ok = false;
    int st_ty = 0, st_tz = 0, st_tiles = 0, st_tiles_z = 0, st_maxr = 0;
    uint32_t st_max_ids = 0;                     // most partial sums of one angle
    std::vector<TileCell> st_cell;
    std::vector<uint32_t> st_win, st_segid, st_off, st_row_first, st_row_nseg;
    std::vector<uint32_t> st_seg;                // 2 words per slot
    std::vector<float> st_w;
    // Sheared-strip all-angle forward projector (k_fp_strip, round 4).  The angles are split into "passes" of rays that run in
    // nearly the same direction; inside a pass the image is cut into strips of FS_W pixels across the rays' mean direction
    // (columns for rays closer to the y axis, rows otherwise), sheared with that direction: fs_shift[pass][u] is the integer
    // offset of the strip pattern at march coordinate u, so a ray stays inside ONE strip for a long stretch.  A workgroup
    // marches one strip tile by tile (FS_H march steps x FS_W pixels x 64 slices in LDS) with the ray sums of the rays inside
    // the strip RESIDENT IN REGISTERS: a ray owns accumulator slot (wave, k, lane group) from the tile where it enters the strip
    // to the tile where it leaves, and emits ONE partial sum per strip it crosses (~6 per ray instead of ~27 per ray for
    // 32 x 16 tiles).  Slots of one angle are dealt round-robin by ray number (ray j -> slot j mod M, M = the widest window of
    // rays of that angle alive in one tile, rounded up to 4), four neighbouring rays forming the four lane groups of one
    // (wave, k): they have the same number of entry batches per tile but for the ends, which is what the per-(tile, wave, k)
    // batch counts fs_cnt (wave-uniform loop bounds) are padded to.
    //   fs_item[i]      one strip of one pass (sorted by work, heaviest first)
    //   fs_cnt          16 bytes per (tile, wave): HALF batches of slot k = byte k (h >> 1 whole batches, then, h odd, one half batch:
    //                   a stream unit whose 4 entries are stored twice)
    //   fs_gstart/gseg0 per (item, lane group): first batch of its entry stream / first partial-sum id
    //   fs_off / fs_w   entry batches of TILE_BATCH entries {LDS byte offset | flush flag << 31, weight}; a flagged batch ends a
    //                   ray's stay in the strip: the accumulator is stored as the group's next partial sum and cleared
    //   fs_rseg_*       per ray the partial-sum ids of its strips, ascending strip (the reduce kernel's fixed order)
    static constexpr int FS_W = 16, FS_H = 16, FS_WAVES = 8, FS_GROUPS = 32, FS_KMAX = 16;
    struct FsItem { int32_t pass, v0; uint32_t tile0, ntiles, cnt0, g0, work, pad; };
    bool fs_ok = false;
    int fs_npass = 0, fs_kused = 0;
    uint32_t fs_nseg = 0;
    uint64_t fs_real_entries = 0;                // statistics: matrix entries / padded entry slots / volume pixels staged
    uint64_t fs_slots = 0, fs_staged_pixels = 0;
    std::vector<int32_t> fs_orient;              // [npass] 0: march along y (strips of columns), 1: march along z
    std::vector<int32_t> fs_shift;               // [npass * N]
    std::vector<FsItem> fs_item;
    std::vector<uint8_t> fs_cnt;
    std::vector<uint32_t> fs_gstart, fs_gseg0;
    // entries as the kernel loads them: low word = LDS byte offset | flush flag << 31, high word = the weight's bits.  Allocated
    // WITHOUT initialisation and filled by the emission threads (1.6 GB at 1024^2 x 120: a zero fill alone cost seconds)
    std::unique_ptr<uint64_t[]> fs_ent;
    size_t fs_ent_n = 0;
    // The same strips as entry LISTS of the wave-per-angle form (k_fp_list, round 4): a workgroup of FL_WAVES waves marches an item (pass,
    // strip of FL_W pixels, march segment) in tiles of FL_TH march steps; wave w owns the rays of the pass's w-th angle, ray j in
    // accumulator j mod FL_ACC (two registers: 128 slices); per (item, tile, wave) a list of entries {byte offset of the pixel in the
    // double-buffered LDS tile | accumulator register, weight} padded to whole batches of FL_BATCH, and a list of flush records
    // {accumulator register, partial-sum id} of the rays whose stay in the strip ends in that tile.
    static constexpr int FL_W = 16, FL_H = 16, FL_TH = 8, FL_WAVES = 16, FL_ACC = 32, FL_BATCH = 16, FL_PIXB = 512, FL_REGS = 2;
    struct FlItem { int32_t pass, v0; uint32_t tile0, ntiles, lp0, work, pad0, pad1; };
    bool fl_ok = false;
    int fl_npass = 0;
    uint32_t fl_nseg = 0;
    uint64_t fl_real_entries = 0, fl_slots = 0, fl_staged_pixels = 0;
    double fl_balance = 0.0;                     // mean / max batches per wave of a tile, over all tiles (1 = every wave equally loaded)
    std::vector<FlItem> fl_item;                 // heaviest first
    std::vector<int32_t> fl_orient, fl_shift;    // [npass], [npass * N]
    std::vector<uint32_t> fl_ptr, fl_fptr;       // [sum of ntiles * FL_WAVES + 1] first batch / first flush record of list lp0 + tile * FL_WAVES + wave
    std::unique_ptr<uint64_t[]> fl_ent;          // [(batches + 1) * FL_BATCH]
    size_t fl_ent_n = 0;
    std::vector<uint64_t> fl_flush;              // register | id << 32
    std::vector<uint32_t> fl_rseg_ptr, fl_rseg_idx;   // per ray the partial-sum ids of its stays, ascending strip (the reduce kernel's fixed order)
    static uint64_t fs_pack(uint32_t off, float w) { uint32_t b; std::memcpy(&b, &w, 4); return (uint64_t)off | ((uint64_t)b << 32); }
    static uint32_t fs_off_of(uint64_t e) { return (uint32_t)e; }
    static float fs_w_of(uint64_t e) { uint32_t b = (uint32_t)(e >> 32); float w; std::memcpy(&w, &b, 4); return w; }
    std::vector<uint32_t> fs_rseg_ptr, fs_rseg_idx;
};

void build_parallel_ray(int N, int P, const double *angles_rad, Coo &out);
bool coo_from_triplets(int64_t nrow, int64_t ncol, int64_t nnz, const float *rows, const float *cols,
                       const float *vals, Coo &out, std::string &err);
void sort_rows(Coo &m);
bool build_tables(const Coo &m, int N, int P, Tables &t, std::string &err);
void build_walk(const Coo &m, int N, int P, Tables &t);
void build_segments(int N, int P, int seg_len, Tables &t);
void build_tiles(const Coo &m, int N, int P, int TY, int TZ, int pixel_bytes, Tables &t);
void build_sart_tiles(const Coo &m, int N, int P, int TY, int TZ, int max_rows, int pixel_bytes, Tables &t);
void build_bp_tiles(int N, int P, int TY, int TZ, int stage_angles, int max_rows, int row_bytes, int pad_angles, Tables &t);
// needs t.cell (build_tables); regs_per_pixel: accumulator registers of one pixel (index step)
void build_bp_lists(int N, int P, int TY, int TZ, int stage_angles, int max_rows, int row_bytes, int waves, int batch, int regs_per_pixel, Tables &t);
bool build_fp_lists(const Coo &m, int N, int P, Tables &t, std::string &why);
bool build_fp_strips(const Coo &m, int N, int P, int pixel_bytes, int nchunk, Tables &t, std::string &why);   // nchunk: 64-slice chunks of the slab

}  // namespace tomo
