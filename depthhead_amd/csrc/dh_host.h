// dh_host.h -- the host-only logic of the runtime behind include/depthhead_hip.h: error slots, forest validation, the
// diagnostic switches, patch-grid and tile geometry, the general path's integer node bounds, the mean-shift kernel table,
// upload chunking, and the validation / packing of BIWI run-length coded payloads.
//
// No HIP type or call appears here: dh_host.cpp and dh_biwi.cpp build with plain g++ as well as with hipcc, which is
// what puts them under AddressSanitizer / UBSan / ThreadSanitizer on the CPU (tests/host/, tests/test_host_sanitize.py;
// GPU sanitizers are unavailable on the pool).  dh_api.hip is the only caller in the product library.  Not part of the ABI.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <new>
#include <vector>

#include "../../include/depthhead_hip.h"

// ------------------------------------------------------------------ errors
// One message slot per host thread (dh_last_error); returns `code` so that `return dh_fail_(...)` reads naturally.
int dh_fail_(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *dh_err_get_(void);
void dh_err_set_(const char *msg);

// Every extern "C" entry point runs its body through this: the header promises that nothing throws or aborts across the
// boundary, and the bodies allocate (std::vector, std::string, std::thread).  bad_alloc -> DH_ENOMEM, anything else -> DH_EINVAL.
template <typename F>
static inline int dh_guard_(const char *what, F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return dh_fail_(DH_ENOMEM, "%s: out of host memory", what);
    } catch (const std::exception &e) {
        return dh_fail_(DH_EINVAL, "%s: %s", what, e.what());
    } catch (...) {
        return dh_fail_(DH_EINVAL, "%s: unexpected exception", what);
    }
}

// ------------------------------------------------------------------ forest (host copy, validated)
struct dh_forest {
    std::vector<int32_t> roots;
    std::vector<dh_node> nodes;
    std::vector<double> leaf_prob;
    std::vector<uint32_t> off_begin, rot_begin;
    std::vector<float> offsets;
    std::vector<double> rotations;
    uint32_t max_depth = 0;
    uint16_t max_x = 0, max_y = 0;   // largest rectangle corner used by any node
    bool uniform = false;            // every split rectangle has the same non-empty size
    uint16_t rw = 0, rh = 0;
};
// Body of dh_forest_create: copies and validates (header comment of dh_forest_create lists what is refused).
int dh_forest_build_(const dh_forest_desc *d, dh_forest **out);

// The offset votes once more as (x, y, z, 0) records, every leaf's run padded to a multiple of 4 records (64 bytes):
// b4[l] = index of leaf l's first record, o4 = 4 floats per record (+ 4 records of slack).
void dh_pack_off4_(const dh_forest &f, std::vector<uint32_t> &b4, std::vector<float> &o4);

// General-path node with integer split bounds (k_traverse<false, true>): with C_i = max(c_i, 1),
// delta = (s1 C2 - s2 C1) / (C1 C2) and |d - delta| < 2^-35, so D = s1 C2 - s2 C1 <= ilo decides Zero and
// D >= ilo + 1 + amb decides One (pad covers the rounding of the f64 products; see dh_build_nodes_g_).
struct __attribute__((aligned(16))) NodeG {
    uint8_t  r1[4], r2[4];      // x0, y0, x1, y1 of the two rectangles
    int64_t  ilo;
    int32_t  child_zero, child_one;
    uint32_t amb;               // integers strictly between ilo and ihi (saturating)
    uint32_t cc;                // C1 | C2 << 16
};
void dh_build_nodes_g_(const dh_forest &f, std::vector<NodeG> &out);   // patches of at most 255 x 255

// ------------------------------------------------------------------ diagnostic switches
// The environment is read ONCE, by dh_predictor_create; none of these changes results (the GPU suite runs under each).
// The switches that truncate kernels for per-phase profiling ("results invalid") exist only in builds with
// -DDH_PROFILING_KNOBS (tools/), never in the product library.
struct Knobs {
    bool force_general = false;       // DH_FORCE_GENERAL: mixed-rectangle path for any forest
    bool no_leaf_hist = false;        // DH_NO_LEAF_HIST
    bool box_no_ring = false;         // DH_BOX_NO_RING
    uint32_t leaf_hist_max = 16384;   // DH_LEAF_HIST_MAX: per-frame leaf histogram for the rotation gather (64 KB per frame at most)
    int lds_budget_kb = 0;            // DH_LDS_BUDGET_KB
    int tile_x = 0, tile_y = 0;       // DH_TILE=px,py
    int box_band = 64;                // DH_BOX_BAND
    int box_bands = 0;                // DH_BOX_BANDS: bands per frame of the LDS-ring k_boxsum (0 = chosen by dh_box_bands_)
    int max_resident = 512;           // DH_MAX_RESIDENT_FRAMES
    int chunks = 0;                   // DH_CHUNKS: forked sub-batches per call; 0 = automatic (two once a call brings >= 512 frames)
    bool box_dense = false;           // DH_BOX_DENSE: k_boxsum stores every cell (no skipping of zero over zero)
    bool no_region = false;           // DH_NO_REGION: k_cluster always gathers its first region itself
    int region_min_hits = 0;          // DH_REGION_MIN_HITS: hit records in a frame from which k_region pre-gathers (0 = automatic)
    int top_levels = -1;              // DH_TOP_LEVELS: tree levels walked from the LDS copy of the tree tops (-1 = auto, 0 = none)
    bool no_zero_fold = false;        // DH_NO_ZERO_FOLD: the per-batch counters get a fill dispatch of their own instead of being cleared by k_boxsum
    bool no_tile_list = false;        // DH_NO_TILE_LIST: k_traverse launches a workgroup per tile position, empty ones included
    bool vote_exact = false;          // DH_VOTE_EXACT: k_vote takes the two IEEE divisions for every vote
    bool no_absorb = false;           // DH_NO_ABSORB: uniform path walks the guarded node table
    bool no_general_int = false;      // DH_NO_GENERAL_INT: general path with the f64 divisions on every visit
    int stage_chunk = 64;             // DH_STAGE_CHUNK: frames per upload chunk of the host entry points
    int host_threads = 8;             // DH_HOST_THREADS: host threads that validate / pack run-length coded payloads
    bool cl_stamps = false;
    int trav_stop = 0, emit_stop = 0, vote_stop = 0, cl_stop = 0;   // (-DDH_PROFILING_KNOBS builds only)
    bool trav_stamps = false;
};
Knobs dh_read_knobs_();

// ------------------------------------------------------------------ geometry
struct Geom {
    int w = 0, h = 0, nx = 0, ny = 0, npatch = 0;
    int px = 0, py = 0, tiles_x = 0, tiles_y = 0, ss_max = 0, ss_row = 0, swz_log2 = 0, swz_q = 0;
    bool uniform = false;       // uniform-rectangle path: k_boxsum feeds k_traverse<true>
    int top_levels = 0;         // uniform path with the walk table: tree levels walked from the LDS copy of the tree tops
    int flag_words = 0;         // per frame: u32 words holding one flag byte per tile
    int win_cap = 0;            // slots of a frame's window list: tiles * px * py
    int box_plane = 0, box_rows = 0, box_ow = 0, box_oh = 0, box_parts = 0, box_bands = 0;
    size_t lds = 0;
};
// Sliding-window grid (prediction.rs:535-548, 684-686); refuses what the reference panics on.
int dh_patch_grid_(const dh_params &p, int w, int h, int *nx, int *ny);

// LDS image one tile works on (k_traverse.hip): rw, rh > 0 selects the uniform (box-sum region) layout, 0 the general (SAT) one
void dh_traverse_swizzle(int px, int step, int sw, int rw, int *swz_log2, int *swz_q, int *ss_row);
int dh_traverse_row_stride(int px, int step, int sw, int rw);
size_t dh_traverse_lds_bytes(int px, int py, int step, int sw, int sh, int top_words, int rw, int rh);

// What choose_tile needs to know of a predictor.
struct TileQuery {
    dh_params params{};
    int f_rw = 0, f_rh = 0;          // the forest's one rectangle size (uniform path), else 0
    uint32_t n_trees = 0;
    bool absorb_ok = false;          // the uniform path walks the walk table (tree tops in LDS)
    int top_levels = -1;             // forced number of LDS tree levels, or -1
    bool one_pass = false;           // single-frame workspace: a tile holds at most 3200 / n_trees windows (its walks are ONE lock-step pass), wider than high
    int lds_budget_kb = 0, tile_x = 0, tile_y = 0, box_band = 64;
};
// Tile of PX x PY window positions per workgroup for g.{w, h, nx, ny, uniform}: fills the rest of g.  Returns DH_OK,
// a negative DH_E* (message set), or 1 = "no tile fits the uniform layout: retry on the general path".
int dh_choose_tile_(const TileQuery &q, Geom &g);
static const int kBoxSpan = 256, kBoxMaxRect = 96;   // image columns one k_boxsum wave spans; largest rectangle edge it serves
// Bands a frame is cut into by the LDS-ring k_boxsum for a batch of n frames: band height `*oh` (whole mask blocks of `blk` rows)
// and the number of bands.  A workgroup is 4 waves = 4 (part, band) units and the chip holds `wg_slots` of them (3 per CU: the
// ring's LDS); a wave marches its band + rh - 1 rows.  Picks the band count with the smallest (march) x (rounds of workgroups).
int dh_box_bands_(int n, int parts, int rows, int blk, int rh, int wg_slots, int *oh);

// ------------------------------------------------------------------ small numeric tables (f32, no FMA: -ffp-contract=off)
void dh_mat3_inv_f32_(const float m[9], float o[9]);                 // Mat3<f32>::inv (meancov_estimation.rs:339-352)
void dh_build_kernel_r2_(float sigma, std::vector<float> &k, size_t padded);   // the same weights indexed by dx^2 + dy^2 + dz^2 (0 .. 300)
void dh_build_kernel_table_(float sigma, std::vector<float> &k);    // FullArray3D::build_kernel(20, sigma) (meanshift.rs:228-252), summation order
int dh_blur_taps_(float sigma, std::vector<float> &k);               // imageproc gaussian_kernel_f32 (restated; parity unpinned)

// ------------------------------------------------------------------ upload chunking of the host entry points
#define DH_STAGE_EVENTS 16       // upload chunks in flight per slice
// Chunk starts of a slice of m frames: cstart[0 .. nchunks], cstart[nchunks] = m.  `single` (parity taps on: they describe
// ONE device batch) gives one chunk.  Sizes taper towards the end of the slice (each at most half of what is left, at least
// 16 frames): the kernels of the last chunk are the only ones no upload hides.
int dh_chunk_plan_(int m, int stage_chunk, bool single, int cstart[DH_STAGE_EVENTS + 1]);

// ------------------------------------------------------------------ BIWI run-length coded payloads (biwi.rs:81-103)
struct DhRun { uint32_t dst, src; };   // first destination pixel of a non-empty run, index of its first value in the blob (u16 units); = uint2 on the device
void dh_parallel_for_(int n, int threads, const std::function<void(int)> &fn);
// One pass over a payload's run headers with the checks of dh_biwi_decode_depth.  runs == nullptr: count only.
// Returns the number of non-empty runs or -1 (message in err).
long dh_rle_scan_(const uint8_t *buf, size_t len, uint32_t W, uint32_t H, DhRun *runs, uint32_t dst0, uint32_t src0, char *err, size_t errn);
struct RlePlan {
    uint32_t W = 0, H = 0;
    std::vector<size_t> blob_off;     // [n + 1] byte offset of every payload in the blob (16-byte aligned)
    std::vector<uint32_t> run_begin;  // [n + 1]
    size_t nruns = 0;
};
// Pass A: validates every payload and counts its runs.  Nothing has been touched when this fails.
int dh_rle_plan_(const uint8_t *const *bufs, const size_t *lens, int n, int threads, RlePlan &plan);
// Pass B: payload bytes as they are into blob[plan.blob_off[n]], run table into runs[plan.nruns].
void dh_rle_pack_(const uint8_t *const *bufs, const size_t *lens, int n, int threads, const RlePlan &plan, uint8_t *blob, DhRun *runs);
