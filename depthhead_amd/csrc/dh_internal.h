// dh_internal.h -- structures shared by the host runtime (dh_api.hip) and the gfx950 kernels
// (k_*.hip): kernel argument blocks, device views of the forest, record layouts, launchers.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dh_host.h"

#define DH_GRID 20             // GUESS_GRID_PARTS and the mean-shift kernel edge
#define DH_GRID3 8000          // 20^3
#define DH_POSGRID 400         // 20^2
#define DH_ROTPARTS 120
#define DH_REGION_CELLS (26 * 26 * 26)   // k_cluster's LDS region of an accumulator (RG^3 in k_cluster.hip)
#define DH_KERN_R2 304                    // Gaussian weights by squared distance 0 .. 300 (the 20^3 kernel holds 301 distinct values), padded
#define DH_SUPER_CELLS (64 * 64 * 64)    // k_region's block of an accumulator in global memory (SRG^3 in k_cluster.hip)

// leaf_flags bits (written by k_leaf_prepare)
#define LF_PROB 1u   // prob > 0.0                      (prediction.rs:590)
#define LF_ROT 2u    // trace(cov(rotations)) <= 400.0  (prediction.rs:600)
#define LF_OFF 4u    // trace(cov(offsets))  <= 5200.0  (prediction.rs:643)
#define LF_FIN 8u    // every offset vote of the leaf is finite and below 1e30 in magnitude (k_vote's pinhole fast path relies on it)

// Everything k_traverse needs to turn (patch, leaf) into its three hit records: 4 x 16-byte loads.
struct __attribute__((aligned(16))) LeafTpl {
    float    omin[3], omax[3];   // offset bounding box (off_min / off_max)
    uint32_t v;                  // valtoadd
    uint32_t fc;                 // LF_* | n_offsets << 8
    uint32_t ob;                 // first offset vote
    uint32_t rlo, rhi;           // rotation-bin bounding box (rlo = 0xFFFFFFFF: no rotation votes)
    uint32_t rb, n_rot;          // rotation cells: first index, n_distinct_fine | n_distinct_rough << 16
    uint32_t flags;              // LF_* again, next to prob: k_emit gates a window with ONE 16-byte gather per tree
    double   prob;               // leaf probability (houghforest.rs:75), byte offset 56
};

// Device view of a forest: the flat arrays of dh_forest_desc plus the per-leaf tables that depend
// only on the leaf (last note) and are computed once by k_leaf_prepare.
struct DevForest {
    const int32_t  *roots;
    const dh_node  *nodes;
    const double   *leaf_prob;
    const uint32_t *off_begin;
    const uint32_t *rot_begin;
    const float    *offsets;
    const float4   *off4;        // the same votes as 16-byte records (x, y, z, 0), every leaf's run starting on a 64-byte line:
    const uint32_t *off4_begin;  // per leaf, index of its first record in off4 (a multiple of 4); hit records carry this index
    const double   *rotations;
    uint32_t n_trees, n_nodes, n_leaves, n_off, n_rot;
    // prepared
    uint32_t *leaf_v;      // valtoadd                                   (prediction.rs:594-595)
    uint8_t  *leaf_flags;  // LF_*
    // rotation votes depend only on the leaf, and a leaf's votes fall into few distinct cells: each
    // leaf's slice [rot_begin, rot_begin + n) of these arrays holds its DISTINCT cells + multiplicities
    uint32_t *rot_bin;     // distinct fine bins r1 | r2<<8 | r3<<16 | votes<<24 (:605-627; a bin with more than 255 votes: several entries)
    uint16_t *rot_mult;    // (k_leaf_prepare's scratch: votes per distinct fine bin before they move into rot_bin's top byte)
    uint16_t *rot_rough;   // (k_leaf_prepare's scratch) distinct indices into the 20^3 guess grid   (:630-636)
    uint16_t *rough_mult;  // (scratch) votes per distinct guess-grid cell
    uint32_t *rough_cell;  // the two as one word per distinct guess-grid cell: index | votes << 16 (one load per cell in k_vote)
    float    *off_min;     // per leaf, 3 floats: component-wise min of its offsets (-inf if non-finite)
    float    *off_max;     // per leaf, 3 floats                                     (+inf if non-finite)
    uint32_t *rbin_box;    // per leaf: component-wise minimum of its rotation bins, r1 | r2<<8 | r3<<16
    uint32_t *rbin_box_hi; // per leaf: component-wise maximum, same packing
    struct LeafTpl *tpl;   // per leaf: everything a hit record needs, one 64-byte line
    uint4 *rot_dir;        // per leaf: {rbin_box (0xFFFFFFFF: no rotation votes), rbin_box_hi, rot_begin, distinct fine bins}: what the mean shift's
                           // gather by leaves needs of a leaf, in one load
};

// One (gated patch, voting leaf) pair = three self-contained 16/32-byte records, so that neither the
// vote kernels nor the mean shift chase leaf tables: after one coalesced record read the only
// dependent access left is the vote array itself.
struct __attribute__((aligned(16))) HitRec {
    float    p3[3];    // patch centre in camera space (prediction.rs:554)
    uint32_t ob;       // index of the leaf's first offset vote
};
// Cell bounding box of the hit's position votes: cell_k = trunc(p3_k - o_k) is monotone in o_k, so
// every vote of the leaf lands in [lo_k, hi_k] = [trunc(p3_k - omax_k), trunc(p3_k - omin_k)].
struct __attribute__((aligned(16))) HitBox {
    int32_t  lo[3];
    int32_t  hi0;
    int32_t  hi1, hi2;
    uint32_t v;        // valtoadd of the leaf (prediction.rs:594-595)
    uint32_t fc;       // LF_* flags | n_offsets << 8
};
// Rotation votes do not depend on the patch: bounding box of the leaf's rotation bins.
struct __attribute__((aligned(16))) HitRot {
    uint32_t lo;       // r1 | r2<<8 | r3<<16 minima; 0xFFFFFFFF when the leaf casts no rotation vote
    uint32_t hi;
    uint32_t rb;       // index of the leaf's first rotation cell
    uint32_t n_rot;    // distinct fine bins | distinct guess-grid cells << 16
};

struct TraverseArgs {
    const uint16_t *frames;
    int n_frames, w, h;
    int step, sw, sh;
    int nx, ny;             // patch grid
    int px, py;             // tile size in patches
    int tiles_x, tiles_y;
    int ss_max;             // LDS capacity of the SAT (general path) / box-sum region (uniform path) in words
    int ss_row;             // its row stride in words (general: odd; uniform: a multiple of 4), the same for every tile
    int uniform;            // 1: every split rectangle is rw x rh -> box-sum fast path fed by k_boxsum
    int rw, rh;
    uint32_t area;          // rw * rh
    int swz_log2, swz_q;    // uniform path: LDS slot of region cell (y, x) = y * ss_row + (x & (m - 1)) * swz_q + (x >> swz_log2), m = 1 << swz_log2
    const uint32_t *box;    // uniform path: [n_frames][box_rows][m planes][box_plane] box-sum images written by k_boxsum
    int box_plane, box_rows;
    const uint32_t *tile_list;       // nullable: [8][tile_list_stride] the flagged (frame group << 16 | tile) pairs per x = frame mod 8 (k_tile_list)
    const uint32_t *tile_list_count; // [8]
    uint32_t tile_list_stride;
    const uint32_t *gen;       // the batch's tile-flag tag (BoxArgs::gen)
    const uint8_t *tile_flags; // [n_frames][tiles] == *gen: the tile's region holds a non-zero box sum (k_boxsum) / its footprint a non-zero pixel (k_pixflags)
    const void *nodes_u;    // NodeU[n_nodes], built by k_nodes_compact for this region layout
    const void *nodes_a;    // uniform path: NodeU[n_nodes + n_amb + 1] walk table (children as byte offsets; walk_absorb, k_nodes_compact), or NULL
    uint32_t walk_lb;        // with nodes_a: byte offset of the entry behind its nodes ((n_nodes + n_amb) << 4); leaf l is walk_lb + 16 l
    const uint32_t *amb_list; // with nodes_a: the ambiguous nodes (k_nodes_compact)
    const uint32_t *top_tab; // with nodes_a: the first top_levels levels of every tree as an implicit heap + the entries below them (k_top_build)
    int top_levels;
    const void *nodes_g;    // general path: NodeG[n_nodes] with integer split bounds (patches up to 255 x 255), else NULL
    unsigned long long *dbg_stamps; // profiling: [8] summed cycles per phase (region build, gate, walks) over all workgroups (env DH_TRAV_STAMPS)
    int stop_phase;         // profiling knob (env DH_TRAV_STOP): 0 = run everything, 9 / 1 / 3 = return at entry / after the region build / after the gate
    DevForest f;
    // window list (read by k_emit): tile t owns slots [t * px * py, t * px * py + win_count[frame][t])
    uint32_t *win_count;    // [n_frames][tiles] active windows per tile (zeroed per batch by the host)
    uint32_t *win_patch;    // [n_frames][win_cap] position of the window in the frame's window grid
    void     *win_leaf;     // [n_frames][T][win_cap] leaf reached in every tree: u16 entries when leaf_ls == 1 (forests of <= 65 535 leaves), else i32
    int       leaf_ls;      // log2 of the entry size of win_leaf (1 or 2)
    int       win_cap;      // tiles * px * py
    int32_t  *dbg_leaf;     // nullable [n][npatch][T]
    uint8_t  *dbg_flags;    // nullable [n][npatch]
};

// k_emit: probability gate + hit records for every slot of the window list.
struct EmitArgs {
    const uint16_t *frames;
    int n_frames, w, h;
    int step, lw, lh;
    int nx, npatch;
    int px, py, tiles;
    float kinv[9];
    DevForest f;
    const uint32_t *win_count;
    const uint32_t *win_patch;
    const void     *win_leaf;   // see TraverseArgs
    int       leaf_ls;
    int       win_cap;
    HitRec   *hits;
    HitBox   *hit_box;
    HitRot   *hit_rot;
    uint32_t *hit_count;    // [n_frames]
    uint32_t  hits_cap;     // records per frame
    uint32_t *leaf_hits;    // nullable [n_frames][n_leaves]: how often each leaf cast rotation votes (zeroed per batch)
    uint32_t *gen;          // nullable: the tile-flag tag of this kernel sequence, moved on here (every reader of the batch's flags has run)
    uint8_t  *dbg_flags;    // nullable [n][npatch]: bit 1 set for windows that pass the gate
    int stop;               // profiling knob (env DH_EMIT_STOP): 1 / 2 = return after the window lookup / after the gate
};

// k_boxsum: per frame the image of all rw x rh rectangle sums, out[y][x] = sum of the rectangle whose
// top-left pixel is (x, y), for x <= w - rw, y <= h - rh, columns de-interleaved like k_traverse's LDS region.
struct BoxArgs {
    const uint16_t *frames;
    const uint16_t *zeros;  // >= 8 zero bytes, 8-byte aligned (stands in for the columns right of the image)
    int n_frames, w, h;
    int rw, rh;
    uint32_t *out;
    int plane, rows, lg;    // a row is (1 << lg) planes of `plane` words (multiple of 4): column x sits in plane x mod m at x / m; rows = h - rh + 1
    int ow, oh;             // rectangle origins one wave produces: ow columns (multiple of 4, <= 256 - rw) x oh rows
    uint8_t *tile_flags;    // [n_frames][tiles_x * tiles_y] k_traverse tiles with a non-zero sum in their region: set to the batch's TAG (*gen)
    const uint32_t *gen;    // the batch's tile-flag tag, 1 .. 255 (k_emit moves it on): a flag holding any other value is clear -- or a stale,
                            // harmless false positive (the flags are conservative) -- so the flags never need a fill
    uint32_t *zero_ptr;     // nullable: the per-batch counters [0, zero_lo) and [zero_hi, zero_end) (words), zeroed by this kernel's waves
    uint32_t zero_lo, zero_hi, zero_end;   // instead of by a fill of their own (the tile flags sit in [zero_lo, zero_hi))
    int tiles_x, tiles_y;   // k_traverse's tiling: tile (tx, ty) reads columns [tx * tpx, tx * tpx + tbw), rows [ty * tpy, ty * tpy + tbh)
    int tpx, tpy, tbw, tbh;
    int parts, bands;       // waves across / down a frame
    int blocks_per_frame;   // ceil(parts * bands / 4)
    int ring;               // 1: keep the rh - 1 window rows in a wave-private LDS ring (rh - 1 <= 28) instead of re-reading them
    // Sparse stores: [n_frames][ceil(rows / 32)][parts] 64-bit masks, bit l = lane l of that column part wrote a non-zero sum into
    // that 32-row block the last time this frame slot was filled.  A lane whose four sums of a row are zero and whose bit is clear
    // skips the store: the cells hold zero already (buffer and masks are zeroed together when the workspace is allocated, and every
    // fill keeps "cell non-zero => bit set").  Band heights are multiples of 32 rows so that one wave owns a block.  NULL: store everything.
    unsigned long long *blk_mask;
    int mask_blocks;        // ceil(rows / block height)
    int blk_shift;          // log2 of the mask blocks' height: 5 (32 rows); 3 on single-frame workspaces, whose bands are 8 rows
};

// k_pixflags: tile flags of the general path (non-zero pixel under the tile's footprint).
struct PixFlagArgs {
    const uint16_t *frames;
    int n_frames, w, h;
    uint8_t *tile_flags;    // [n_frames][tiles_x * tiles_y], set to the batch's tag
    const uint32_t *gen;    // see BoxArgs
    int tiles_x, tiles_y;   // tile (tx, ty) covers pixel columns [tx * tpx, tx * tpx + tfw), rows [ty * tpy, ty * tpy + tfh)
    int tpx, tpy, tfw, tfh;
};

struct VoteArgs {
    int n_frames, w, h;
    float k[9];
    DevForest f;
    const HitRec   *hits;
    const HitBox   *hit_box;
    const HitRot   *hit_rot;
    const uint32_t *hit_count;
    uint32_t  hits_cap;
    uint32_t *pos_grid;     // [n][400]
    uint32_t *rot_grid;     // [n][8000]
    const uint32_t *leaf_hits; // nullable [n][n_leaves]: rotation votes per leaf (k_emit); then the 20^3 grid is built per leaf, not per hit
    int stop;               // profiling knob (env DH_VOTE_STOP): 1 / 2 / 3 = return after the LDS set-up / the hit records / the leaf histogram
    int cell_fast;          // w and h are multiples of 20: a vote's guess-grid cell may be taken from an approximate quotient (vote_positions)
    float sx, sy;           // 20 / w, 20 / h
    float kxs, cxs, kys, cys; // pinhole intrinsics: fx * sx, cx * sx, fy * sy, cy * sy (k_vote's approximate cell quotient)
};

struct ClusterArgs {
    const uint16_t *frames;
    int n_frames, w, h;
    float kinv[9];
    DevForest f;
    const HitRec   *hits;
    const HitBox   *hit_box;
    const HitRot   *hit_rot;
    const uint32_t *hit_count;
    uint32_t  hits_cap;
    const uint32_t *leaf_hits; // nullable [n_frames][n_leaves], see TraverseArgs
    const uint32_t *pos_grid;
    const uint32_t *rot_grid;
    const float    *kern_r2;   // DH_KERN_R2 floats: the 20^3 Gaussian kernel by squared distance dx^2 + dy^2 + dz^2 (kernel_function takes nothing
                               // else, meanshift.rs:228-232; FullArray3D::build_kernel, :244-252)
    uint32_t  iterations;
    const float   *midp_guess; // nullable, n*3
    const double  *rot_guess;  // nullable, n*3
    const uint8_t *guess_mask; // nullable, n
    dh_pose  *out;
    uint32_t *pre_region;      // nullable [n][2][64^3]: the accumulators' cells around the initial guesses, gathered by k_region (zeroed per batch)
    int       pre_slices;      // workgroups per (frame, accumulator) of k_region
    uint32_t  pre_min_hits;    // frames with fewer hit records are left to k_cluster alone (both kernels read the same count)
    unsigned long long *dbg_stamps; // profiling twin (env DH_CL_STAMPS): [16] summed cycles per phase of the rotation workgroups
    int32_t  *dbg_guess;       // nullable [n][6]
    int32_t  *dbg_trace;       // nullable [2][n][iterations+1][3]
    uint32_t *dbg_steps;       // nullable [2][n]
    int stop;               // profiling knob (env DH_CL_STOP): low 4 bits 1 / 2 / 3 = return after the initial guess / the first region build / the first weighted sum; + 16 / 32: the position / rotation workgroups return at once
};

struct VotesDumpArgs {
    int frame, which;
    DevForest f;
    const HitRec   *hits;
    const HitBox   *hit_box;
    const HitRot   *hit_rot;
    const uint32_t *hit_count;
    uint32_t  hits_cap;
    int32_t  *out;          // cap*4
    uint32_t  cap;
    uint32_t *count;
};

// Sibling consumers of the walk (prediction.rs:760-905): they read the per-(patch, tree) leaf ids
// and the background flags that k_traverse writes when asked to.
struct AuxArgs {
    const uint16_t *frames;
    int n_frames, w, h;
    int step, sw, sh, lw, lh, nx, ny;
    float k[9], kinv[9];
    DevForest f;
    const int32_t *leaf;     // [n][npatch][T]
    const uint8_t *flags;    // [n][npatch] bit0 = non-background
    uint8_t  *mask;          // predict_mask output [n][h][w]
    uint32_t *hough32;       // 2-D Hough votes accumulated in 32 bits [n][h][w]
};

// k_rle_decode: BIWI run-length coded depth payloads -> frames (biwi.rs:81-103), see dh_kernels.hip
struct RleArgs {
    const uint16_t *blob;       // the uploaded payload bytes, viewed as u16 (every run's data sits at an even byte offset)
    const uint2    *runs;       // per non-empty run: {first destination pixel in `frames`, index of its first value in `blob`}
    const uint32_t *run_begin;  // [n_frames + 1] runs of frame i = [run_begin[i], run_begin[i + 1])
    uint16_t       *frames;     // [n_frames][h][w], zero-filled before the launch
    int n_frames, blocks_per_frame;
};

struct Mat3Arg { float m[9]; };

// launchers (dh_kernels.hip)
hipError_t dh_launch_blur_u16(const uint16_t *in, uint16_t *tmp, uint16_t *out, int n, int w, int h, const float *kern, int klen, hipStream_t s);
hipError_t dh_launch_argmax2d(const uint16_t *hough, const uint16_t *frames, int n, int w, int h, const float kinv[9], dh_pose *out, hipStream_t s);
hipError_t dh_launch_rle_decode(const RleArgs &a, hipStream_t s);
hipError_t dh_region_init();             // k_region's dynamic-LDS attribute (called by dh_kernels_init, device current)
hipError_t dh_launch_zero(void *ptr, size_t bytes, hipStream_t s);   // ptr, bytes multiples of 16
hipError_t dh_launch_mask(const AuxArgs &a, hipStream_t s);
hipError_t dh_launch_hough2d(const AuxArgs &a, uint16_t *out, hipStream_t s);
hipError_t dh_kernels_init(int device);   // once per device: dynamic-LDS limit of the walk kernel
hipError_t dh_launch_leaf_prepare(const DevForest &f, hipStream_t s);
hipError_t dh_launch_nodes_compact(const DevForest &f, int ss, int swz_log2, int swz_q, uint32_t area, void *out, void *out_a, uint32_t *any_amb,
                                   uint32_t *amb_list, uint32_t n_amb, hipStream_t s);
#define DH_AMB_CAP 4096      // ambiguous nodes the walk table can hold (more: the guarded node table is walked)
hipError_t dh_launch_traverse(const TraverseArgs &a, size_t lds_bytes, hipStream_t s);
hipError_t dh_launch_tile_list(const uint8_t *flags, const uint32_t *gen, int n_frames, int tiles, uint32_t *list, uint32_t *count, uint32_t stride, hipStream_t s);
hipError_t dh_launch_emit(const EmitArgs &a, hipStream_t s);
hipError_t dh_launch_vote(const VoteArgs &a, hipStream_t s);
hipError_t dh_launch_cluster(const ClusterArgs &a, hipStream_t s);
hipError_t dh_launch_region(const ClusterArgs &a, hipStream_t s);
hipError_t dh_launch_votes_dump(const VotesDumpArgs &a, hipStream_t s);
hipError_t dh_launch_boxsum(const BoxArgs &a, hipStream_t s);
hipError_t dh_launch_pixflags(const PixFlagArgs &a, hipStream_t s);
hipError_t dh_launch_top_build(const DevForest &f, const void *nodes_a, uint32_t n_amb, int top_levels, uint32_t *out, hipStream_t s);
