// dh_api.hip -- host runtime behind the C ABI of include/depthhead_hip.h: forest validation,
// HBM residency of the forest and its per-leaf tables, the per-batch workspace, kernel sequencing
// on a caller-supplied HIP stream, profiling events and the parity taps.
//
// Sits where HoughPrediction::predict_parameter_generic sits in the reference
// (src/hough/prediction.rs:421-493); the unit of work is a batch of frames.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "dh_internal.h"

// ------------------------------------------------------------------ errors
#define fail dh_fail_          // (dh_host.cpp: one message slot per host thread)
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? DH_ENOMEM : DH_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------ roctx ranges (tracing)
// With profiling on (dh_set_profiling) every kernel launch of a batch sits inside a named roctx range on the host thread --
// "dh:boxsum", "dh:traverse", "dh:emit", "dh:vote", "dh:cluster", and "dh:batch n=..." around them -- which rocprofv3
// --marker-trace shows beside the kernel trace.  The marker library is looked up at run time (librocprofiler-sdk-roctx.so,
// else libroctx64.so): the product library has no link-time dependency on a profiler, and without one the ranges are no-ops.
#include <dlfcn.h>
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(name, RTLD_LAZY | RTLD_GLOBAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() { static const Roctx r; return r; }      // (thread-safe initialisation; only ever touched with profiling on)
struct Range {
    bool on;
    Range(bool enabled, const char *name) : on(enabled && roctx().push) { if (on) roctx().push(name); }
    ~Range() { if (on) roctx().pop(); }
};
}   // namespace

extern "C" const char *dh_last_error(void) { return dh_err_get_(); }
extern "C" int dh_version(void) { return DH_VERSION; }

// ------------------------------------------------------------------ forest (host side: dh_host.cpp validates and copies)
static int forest_create_(const dh_forest_desc *d, dh_forest **out) { return dh_forest_build_(d, out); }

static int forest_destroy_(dh_forest *f) {
    delete f;
    return DH_OK;
}

static int forest_info_(const dh_forest *f, uint32_t *n_trees, uint32_t *n_nodes, uint32_t *n_leaves, uint32_t *max_depth) {
    if (!f) return fail(DH_EINVAL, "dh_forest_info: NULL forest");
    if (n_trees) *n_trees = (uint32_t)f->roots.size();
    if (n_nodes) *n_nodes = (uint32_t)f->nodes.size();
    if (n_leaves) *n_leaves = (uint32_t)f->leaf_prob.size();
    if (max_depth) *max_depth = f->max_depth;
    return DH_OK;
}

// ------------------------------------------------------------------ geometry (dh_host.cpp)
static int patch_grid_(const dh_params *p, int w, int h, int *nx, int *ny) {
    if (!p || !nx || !ny) return fail(DH_EINVAL, "dh_patch_grid: NULL argument");
    return dh_patch_grid_(*p, w, h, nx, ny);
}

// Entry points run on the predictor's device and leave the caller's current device as they found it.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ------------------------------------------------------------------ predictor
#define DH_MAX_CHUNKS 8
#define DH_MIN_CHUNK_FRAMES 16

struct dh_predictor {
    int device = 0;
    Knobs knobs;
    dh_params params{};
    uint32_t n_trees = 0, n_nodes = 0, n_leaves = 0, n_off = 0, n_rot = 0, max_depth = 0;
    DevForest dev{};
    std::vector<void *> forest_allocs;
    float *kern_r2 = nullptr;    // device, DH_KERN_R2 floats: the mean-shift kernel by squared distance
    uint16_t *zeros = nullptr;   // device, 64 zero bytes (k_boxsum reads them for columns right of the image)
    bool f_uniform = false;      // forest has one split-rectangle size
    int f_rw = 0, f_rh = 0;
    void *nodes_g = nullptr;     // NodeG[n_nodes]: general-path nodes with integer split bounds (patches up to 255 x 255), else NULL
    void *nodes_u = nullptr;     // 16-byte compact nodes for the current region layout (uniform path)
    void *nodes_a = nullptr;     // NodeU[n_nodes + n_amb + 1]: the same nodes as the walk table of walk_absorb (children as byte offsets), or NULL
    uint32_t *amb_flag = nullptr; // device word: number of nodes with an ambiguity band (k_nodes_compact's probe pass)
    uint32_t *amb_list = nullptr; // device: their indices (at most DH_AMB_CAP)
    uint32_t n_amb = 0;
    bool absorb_ok = false;      // the uniform path walks nodes_a (at most DH_AMB_CAP ambiguous nodes, table offsets fit 32 bits)
    uint32_t *top_tab = nullptr; // [T][2^top_levels] {offsets, ilo} heap + [T][2^top_levels] entry offsets (k_top_build), copied to LDS by every tile
    int top_levels = -1;         // DH_TOP_LEVELS, or -1: choose_tile decides per geometry
    long long nodes_u_key = 0;   // (ss_row, swizzle) the compact nodes were built for
    hipStream_t own_stream = nullptr;
    hipStream_t copy_stream = nullptr;    // host entry points: uploads run here, ahead of the kernels on own_stream
    hipEvent_t ev_stage[DH_STAGE_EVENTS] = {};   // chunk k uploaded (recorded on an upload stream, waited for on own_stream)
    hipEvent_t ev_slice = nullptr;        // the kernels that read the staging buffers are done (recorded on own_stream)
    // run-length coded input (dh_predict_batch_rle): pinned staging + device copies of payload blob and run table
    uint8_t *pin_small = nullptr; size_t pin_small_cap = 0;   // page-locked staging of a slice's small host arrays: poses out, guesses in (small_stage)
    uint8_t *pin_blob = nullptr;  size_t pin_blob_cap = 0;
    uint2 *pin_runs = nullptr;    size_t pin_runs_cap = 0;
    uint32_t *pin_begin = nullptr; size_t pin_begin_cap = 0;
    uint8_t *dev_blob = nullptr;  size_t dev_blob_cap = 0;
    uint2 *dev_runs = nullptr;    size_t dev_runs_cap = 0;
    uint32_t *dev_begin = nullptr; size_t dev_begin_cap = 0;
    int chunks = 1;                       // sub-batches per call (env DH_CHUNKS)
    hipStream_t aux_stream[DH_MAX_CHUNKS - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[DH_MAX_CHUNKS - 1] = {};
    // workspace
    Geom geom;
    int cap_frames = 0;
    uint16_t *ws_frames = nullptr;   // host-API staging only
    size_t ws_frames_bytes = 0;
    HitRec *hits = nullptr;
    HitBox *hit_box = nullptr;
    HitRot *hit_rot = nullptr;
    uint32_t *box = nullptr;         // [cap][box_rows][m][box_plane] rectangle-sum images (uniform path)
    uint32_t *tile_list = nullptr;  // [DH_MAX_CHUNKS][8][ceil(cap / 8) * tiles] + [DH_MAX_CHUNKS][8] counts behind it (k_tile_list)
    size_t tile_list_stride = 0;    // entries per x
    unsigned long long *box_mask = nullptr;   // [cap][ceil(box_rows / 32)][box_parts] which lanes wrote non-zero sums last time (BoxArgs::blk_mask)
    uint32_t *win_patch = nullptr;   // [cap][win_cap] window list: position in the window grid
    uint8_t *win_leaf = nullptr;     // [cap][T][win_cap] window list: leaf per tree (u16 entries for forests of <= 65 535 leaves, else i32)
    int leaf_ls = 2;                 // log2 of its entry size
    uint32_t *leaf_hits = nullptr;   // [cap][n_leaves] rotation-vote histogram (inside `counters`), only for forests of <= DH_LEAF_HIST_MAX leaves
    size_t zero_words = 0;           // words of `counters` zeroed before every batch
    uint32_t *gen = nullptr;         // [DH_MAX_CHUNKS] tile-flag tags, one per kernel sequence in flight (BoxArgs::gen)
    size_t zero_lo = 0, zero_hi = 0; // the tile flags' words inside `counters`: [zero_lo, zero_hi)
    int blk_shift = 5;               // log2 height of k_boxsum's mask blocks in this workspace (BoxArgs::blk_shift)
    uint32_t hits_cap = 0;
    uint32_t *pre_region = nullptr;  // [pre_cap][2][64^3] the cells of both accumulators around the initial guesses, gathered by k_region (small batches with many hit records)
    int pre_cap = 0;
    uint32_t pre_min_hits = 0;       // frames with fewer hit records are gathered by k_cluster alone
    uint32_t *counters = nullptr;    // [n] hit_count | [n][400] pos_grid | [n][8000] rot_grid (one memset)
    dh_pose *ws_poses = nullptr;
    float *ws_midp = nullptr;
    double *ws_rot = nullptr;
    uint8_t *ws_mask = nullptr;
    float *blur_kern = nullptr;      // device: gaussian_kernel_f32(gaussian_sigma) of the 2-D Hough variant, built on first use
    int blur_klen = 0;
    float blur_sigma = 0.0f;
    // leaf-id outputs for predict_mask / the 2-D Hough image, allocated on first use
    int32_t *aux_leaf = nullptr;
    uint8_t *aux_flags = nullptr;
    uint32_t *aux_u32 = nullptr;
    void *aux_out = nullptr;
    size_t aux_out_bytes = 0;
    int aux_cap = 0;
    // captured batch (hipGraph)
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool capturing = false;          // inside dh_graph_capture: one ordered pass on the capture stream
    bool graph_stale = false;        // the workspace a captured batch points into was reallocated: dh_graph_launch refuses
    // taps
    bool debug = false;
    int32_t *dbg_leaf = nullptr;
    uint8_t *dbg_flags = nullptr;
    int32_t *dbg_guess = nullptr, *dbg_trace = nullptr;
    uint32_t *dbg_steps = nullptr;
    int32_t *dbg_votes = nullptr;
    size_t dbg_votes_cap = 0;
    uint32_t *dbg_vcount = nullptr;
    bool dbg_valid = false;
    // last batch
    int last_n = 0;
    const uint16_t *last_frames = nullptr;
    // profiling
    bool profiling = false;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // start, emit end, vote end, cluster end, boxsum end, traverse end
    bool ev_valid = false;
};

template <typename T>
static int dev_alloc(dh_predictor *p, T **out, size_t count, bool track_forest = false) {
    void *ptr = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&ptr, bytes);
    if (e != hipSuccess) return fail(DH_ENOMEM, "hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    if (track_forest) p->forest_allocs.push_back(ptr);
    *out = (T *)ptr;
    return DH_OK;
}
template <typename T>
static int upload(dh_predictor *p, const T **out, const std::vector<T> &v) {
    T *d = nullptr;
    int rc = dev_alloc(p, &d, v.size(), true);
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = d;
    return DH_OK;
}

static int build_kernel_table(dh_predictor *p) {
    std::vector<float> r2;
    dh_build_kernel_r2_(p->params.gaussian_sigma, r2, DH_KERN_R2);    // get_or_build_kernel caches it per sigma (prediction.rs:310-317)
    HIP_TRY(hipMemcpy(p->kern_r2, r2.data(), r2.size() * sizeof(float), hipMemcpyHostToDevice));
    return DH_OK;
}

static void drop_graph(dh_predictor *p) {
    if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    p->graph_exec = nullptr; p->graph = nullptr;
}

static void free_workspace(dh_predictor *p) {
    // a captured batch has the old workspace pointers baked in: replaying it would touch freed memory
    if (p->graph_exec) { drop_graph(p); p->graph_stale = true; }
    void *ptrs[] = {p->tile_list, p->box_mask, p->pre_region, p->box, p->win_patch, p->win_leaf, p->aux_leaf, p->aux_flags, p->aux_u32, p->aux_out, p->ws_frames, p->hits, p->hit_box, p->hit_rot, p->counters, p->ws_poses, p->ws_midp, p->ws_rot, p->ws_mask, p->dbg_leaf,
                    p->dbg_flags, p->dbg_guess, p->dbg_trace, p->dbg_steps, p->dbg_votes, p->dbg_vcount};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    p->pre_region = nullptr; p->pre_cap = 0; p->box_mask = nullptr; p->tile_list = nullptr; p->tile_list_stride = 0;
    p->box = nullptr; p->win_patch = nullptr; p->win_leaf = nullptr; p->leaf_hits = nullptr; p->zero_words = 0;
    p->aux_leaf = nullptr; p->aux_flags = nullptr; p->aux_u32 = nullptr; p->aux_out = nullptr; p->aux_out_bytes = 0; p->aux_cap = 0;
    p->ws_frames = nullptr; p->hits = nullptr; p->hit_box = nullptr; p->hit_rot = nullptr; p->counters = nullptr; p->ws_poses = nullptr; p->ws_midp = nullptr;
    p->ws_rot = nullptr; p->ws_mask = nullptr; p->dbg_leaf = nullptr; p->dbg_flags = nullptr; p->dbg_guess = nullptr;
    p->dbg_trace = nullptr; p->dbg_steps = nullptr; p->dbg_votes = nullptr; p->dbg_vcount = nullptr;
    p->ws_frames_bytes = 0; p->dbg_votes_cap = 0; p->cap_frames = 0; p->hits_cap = 0; p->dbg_valid = false;
    p->geom = Geom();
}

static int predictor_destroy_(dh_predictor *p) {
    if (!p) return DH_OK;
    (void)hipSetDevice(p->device);
    if (p->own_stream) (void)hipStreamSynchronize(p->own_stream);
    drop_graph(p);
    free_workspace(p);
    for (void *q : p->forest_allocs) (void)hipFree(q);
    if (p->kern_r2) (void)hipFree(p->kern_r2);
    if (p->blur_kern) (void)hipFree(p->blur_kern);
    if (p->zeros) (void)hipFree(p->zeros);
    if (p->gen) (void)hipFree(p->gen);
    for (auto &e : p->ev) if (e) (void)hipEventDestroy(e);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    for (auto &e : p->ev_join) if (e) (void)hipEventDestroy(e);
    for (auto &st : p->aux_stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (p->copy_stream) { (void)hipStreamSynchronize(p->copy_stream); (void)hipStreamDestroy(p->copy_stream); }
    for (auto &e : p->ev_stage) if (e) (void)hipEventDestroy(e);
    if (p->ev_slice) (void)hipEventDestroy(p->ev_slice);
    if (p->pin_small) (void)hipHostFree(p->pin_small);
    if (p->pin_blob) (void)hipHostFree(p->pin_blob);
    if (p->pin_runs) (void)hipHostFree(p->pin_runs);
    if (p->pin_begin) (void)hipHostFree(p->pin_begin);
    if (p->dev_blob) (void)hipFree(p->dev_blob);
    if (p->dev_runs) (void)hipFree(p->dev_runs);
    if (p->dev_begin) (void)hipFree(p->dev_begin);
    if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
    delete p;
    return DH_OK;
}

static int predictor_build(dh_predictor *p, const dh_forest *f, const dh_params *prm, int device);

static int predictor_create_(const dh_forest *f, const dh_params *prm, int device, dh_predictor **out) {
    if (!f || !prm || !out) return fail(DH_EINVAL, "dh_predictor_create: NULL argument");
    *out = nullptr;
    if (prm->stepwidth == 0 || prm->subimage_width == 0 || prm->subimage_height == 0) return fail(DH_EINVAL, "zero stepwidth / patch size");
    if (prm->subimage_width > 4096 || prm->subimage_height > 4096) return fail(DH_ESIZE, "patch larger than 4096");
    // rect sums are taken modulo 2^32: exact while sw*sh*65535 < 2^32
    if ((uint64_t)prm->subimage_width * prm->subimage_height * 65535ull >= (1ull << 32)) return fail(DH_ESIZE, "patch area %ux%u too large for u32 rectangle sums", prm->subimage_width, prm->subimage_height);
    if (!(prm->gaussian_sigma == prm->gaussian_sigma)) return fail(DH_EINVAL, "sigma is NaN");
    if (f->max_x > prm->subimage_width || f->max_y > prm->subimage_height)
        return fail(DH_EFOREST, "a split rectangle (max corner %u,%u) leaves the %ux%u patch", f->max_x, f->max_y, prm->subimage_width, prm->subimage_height);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(DH_EINVAL, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    dh_predictor *p = new (std::nothrow) dh_predictor;
    if (!p) return fail(DH_ENOMEM, "out of host memory");
    // (a host allocation failing half-way -- the packed vote arrays, the general-path nodes -- must not leak the device side)
    int rc = dh_guard_("dh_predictor_create", [&]() -> int { return predictor_build(p, f, prm, device); });
    if (rc != DH_OK) {
        const std::string keep = dh_err_get_();
        predictor_destroy_(p);
        dh_err_set_(keep.c_str());
        return rc;
    }
    *out = p;
    return DH_OK;
}

static int predictor_build(dh_predictor *p, const dh_forest *f, const dh_params *prm, int device) {
    p->device = device;
    p->params = *prm;
    p->knobs = dh_read_knobs_();   // the only place the environment is read
    p->n_trees = (uint32_t)f->roots.size(); p->n_nodes = (uint32_t)f->nodes.size(); p->n_leaves = (uint32_t)f->leaf_prob.size();
    p->n_off = f->off_begin.back(); p->n_rot = f->rot_begin.back(); p->max_depth = f->max_depth;
    p->f_uniform = f->uniform; p->f_rw = f->rw; p->f_rh = f->rh;
    int rc = DH_OK;
    DevForest &d = p->dev;
    d.n_trees = p->n_trees; d.n_nodes = p->n_nodes; d.n_leaves = p->n_leaves; d.n_off = p->n_off; d.n_rot = p->n_rot;
#define STEP(x) if (rc == DH_OK) rc = (x)
    STEP(upload(p, &d.roots, f->roots));
    STEP(upload(p, &d.nodes, f->nodes));
    STEP(upload(p, &d.leaf_prob, f->leaf_prob));
    STEP(upload(p, &d.off_begin, f->off_begin));
    STEP(upload(p, &d.rot_begin, f->rot_begin));
    STEP(upload(p, &d.offsets, f->offsets));
    if (rc == DH_OK) {
        // the offset votes once more as (x, y, z, 0) records, every leaf's run on a 64-byte boundary: the kernels
        // that walk a leaf's votes (k_vote, k_cluster) then touch whole cache lines with one 16-byte load per lane
        std::vector<uint32_t> b4;
        std::vector<float> o4f;
        dh_pack_off4_(*f, b4, o4f);
        std::vector<float4> o4(o4f.size() / 4);
        memcpy(o4.data(), o4f.data(), o4f.size() * sizeof(float));
        STEP(upload(p, &d.off4, o4));
        STEP(upload(p, &d.off4_begin, b4));
    }
    STEP(upload(p, &d.rotations, f->rotations));
    STEP(dev_alloc(p, &d.leaf_v, p->n_leaves, true));
    STEP(dev_alloc(p, &d.leaf_flags, p->n_leaves, true));
    STEP(dev_alloc(p, &d.rot_bin, p->n_rot, true));
    STEP(dev_alloc(p, &d.rot_rough, p->n_rot, true));
    STEP(dev_alloc(p, &d.rot_mult, p->n_rot, true));
    STEP(dev_alloc(p, &d.rough_mult, p->n_rot, true));
    STEP(dev_alloc(p, &d.rough_cell, p->n_rot, true));
    STEP(dev_alloc(p, &d.off_min, (size_t)p->n_leaves * 3, true));
    STEP(dev_alloc(p, &d.off_max, (size_t)p->n_leaves * 3, true));
    STEP(dev_alloc(p, &d.rbin_box, p->n_leaves, true));
    STEP(dev_alloc(p, &d.rbin_box_hi, p->n_leaves, true));
    STEP(dev_alloc(p, &d.tpl, p->n_leaves, true));
    STEP(dev_alloc(p, &d.rot_dir, p->n_leaves, true));
    STEP(dev_alloc(p, &p->kern_r2, DH_KERN_R2));
    STEP(dev_alloc(p, &p->zeros, 32));
    STEP(dev_alloc(p, &p->gen, DH_MAX_CHUNKS));
    { uint4 *nu = nullptr; STEP(dev_alloc(p, &nu, p->n_nodes, true)); p->nodes_u = nu; }
    if (p->n_nodes > 0 && (size_t)p->n_nodes + p->n_leaves + 2 * DH_AMB_CAP < ((size_t)1 << 27) && !p->knobs.no_absorb) {   // (byte offsets into the table stay below 2^31)
        STEP(dev_alloc(p, &p->amb_flag, 1, true));
        STEP(dev_alloc(p, &p->amb_list, DH_AMB_CAP, true));
    }
    if (rc == DH_OK && prm->subimage_width <= 255 && prm->subimage_height <= 255 && p->n_nodes > 0) {
        // Integer split bounds of the general path (NodeG: dh_host.h, k_traverse.hip)
        std::vector<NodeG> ng;
        dh_build_nodes_g_(*f, ng);
        NodeG *dg = nullptr;
        STEP(dev_alloc(p, &dg, p->n_nodes, true));
        if (rc == DH_OK && hipMemcpy(dg, ng.data(), ng.size() * sizeof(NodeG), hipMemcpyHostToDevice) != hipSuccess) rc = fail(DH_EHIP, "hipMemcpy(NodeG)");
        p->nodes_g = dg;
    }
#undef STEP
    auto hipstep = [&](hipError_t e, const char *what) {
        if (rc == DH_OK && e != hipSuccess) rc = fail(DH_EHIP, "%s: %s", what, hipGetErrorString(e));
    };
    if (rc == DH_OK) hipstep(hipMemset(p->zeros, 0, 64), "hipMemset");
    if (rc == DH_OK) { const uint32_t ones[DH_MAX_CHUNKS] = {1, 1, 1, 1, 1, 1, 1, 1}; hipstep(hipMemcpy(p->gen, ones, sizeof ones, hipMemcpyHostToDevice), "hipMemcpy(gen)"); }
    if (rc == DH_OK) hipstep(dh_kernels_init(device), "hipFuncSetAttribute");
    if (rc == DH_OK) hipstep(hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking), "hipStreamCreate");
    if (rc == DH_OK) hipstep(dh_launch_leaf_prepare(d, p->own_stream), "k_leaf_prepare launch");
    if (rc == DH_OK) hipstep(hipStreamSynchronize(p->own_stream), "k_leaf_prepare");
    if (rc == DH_OK && p->amb_flag && p->f_uniform) {
        // which nodes carry an ambiguity band?  (independent of the region layout: probed once with a dummy one)
        uint32_t n_amb = DH_AMB_CAP + 1;
        hipstep(hipMemsetAsync(p->amb_flag, 0, sizeof(uint32_t), p->own_stream), "hipMemset");
        if (rc == DH_OK) hipstep(dh_launch_nodes_compact(d, 1, 0, 4, (uint32_t)(p->f_rw * p->f_rh), nullptr, nullptr, p->amb_flag, p->amb_list, 0, p->own_stream), "k_nodes_compact launch");
        if (rc == DH_OK) hipstep(hipMemcpyAsync(&n_amb, p->amb_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, p->own_stream), "hipMemcpy");
        if (rc == DH_OK) hipstep(hipStreamSynchronize(p->own_stream), "k_nodes_compact");
        // (the walk table keeps 12 bytes of LDS per tree even with no level in LDS: not for forests of thousands of trees)
        p->absorb_ok = rc == DH_OK && n_amb <= DH_AMB_CAP && (size_t)p->n_trees * 12 <= 8 * 1024;
        if (p->absorb_ok) {
            p->n_amb = n_amb;
            uint4 *na = nullptr;
            int r2 = dev_alloc(p, &na, (size_t)p->n_nodes + n_amb + 1, true);
            if (r2) rc = r2;
            p->nodes_a = na;
            p->top_levels = p->knobs.top_levels;          // -1: choose_tile decides per geometry
            if (rc == DH_OK) {
                uint32_t *tt = nullptr;
                r2 = dev_alloc(p, &tt, (size_t)p->n_trees * (1u << 8) * 3, true);      // room for the 8 levels choose_tile may go to
                if (r2) rc = r2;
                p->top_tab = tt;
            }
        }
    }
    if (rc == DH_OK) rc = build_kernel_table(p);
    for (auto &e : p->ev)
        if (rc == DH_OK) hipstep(hipEventCreate(&e), "hipEventCreate");
    p->chunks = p->knobs.chunks;
    if (rc == DH_OK) hipstep(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming), "hipEventCreate");
    if (rc == DH_OK) hipstep(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking), "hipStreamCreate");
    for (auto &e : p->ev_stage)
        if (rc == DH_OK) hipstep(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    if (rc == DH_OK) hipstep(hipEventCreateWithFlags(&p->ev_slice, hipEventDisableTiming), "hipEventCreate");
    for (int i = 0; i < DH_MAX_CHUNKS - 1; ++i) {
        if (rc == DH_OK) hipstep(hipStreamCreateWithFlags(&p->aux_stream[i], hipStreamNonBlocking), "hipStreamCreate");
        if (rc == DH_OK) hipstep(hipEventCreateWithFlags(&p->ev_join[i], hipEventDisableTiming), "hipEventCreate");
    }
    return rc;
}

static int predictor_update_sigma_(dh_predictor *p, float val) {
    if (!p) return fail(DH_EINVAL, "NULL predictor");
    if (val == p->params.gaussian_sigma || val <= 0.0f || val != val) return DH_OK;   // prediction.rs:321-323
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    p->params.gaussian_sigma = val;
    return build_kernel_table(p);
}
static int predictor_sigma_(const dh_predictor *p, float *out) {
    if (!p || !out) return fail(DH_EINVAL, "NULL argument");
    *out = p->params.gaussian_sigma;
    return DH_OK;
}

static int choose_tile(const dh_predictor *p, Geom &g, int cap) {
    TileQuery q;
    q.one_pass = cap == 1 && p->knobs.tile_x == 0;     // (a workspace for ONE frame: see dh_choose_tile_)
    q.params = p->params; q.f_rw = p->f_rw; q.f_rh = p->f_rh; q.n_trees = p->n_trees; q.absorb_ok = p->absorb_ok; q.top_levels = p->top_levels;
    q.lds_budget_kb = p->knobs.lds_budget_kb; q.tile_x = p->knobs.tile_x; q.tile_y = p->knobs.tile_y; q.box_band = p->knobs.box_band;
    return dh_choose_tile_(q, g);
}

static int reserve(dh_predictor *p, int n, int w, int h) {
    if (n <= 0) return fail(DH_EINVAL, "batch size must be positive");
    Geom g;
    g.w = w; g.h = h;
    int rc = dh_patch_grid_(p->params, w, h, &g.nx, &g.ny);
    if (rc) return rc;
    g.npatch = g.nx * g.ny;
    bool same_geom = p->geom.w == w && p->geom.h == h;
    bool dbg_ok = !p->debug || p->dbg_leaf;
    if (same_geom && n <= p->cap_frames && dbg_ok) return DH_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    int cap = std::max(n, same_geom ? p->cap_frames : 0);
    free_workspace(p);
    if (g.npatch > 0) {
        // uniform-rectangle fast path: one rectangle size (<= 96 x 96, so a k_boxsum wave yields
        // >= 160 columns), rectangle sums fit i32
        g.uniform = p->f_uniform && (long)p->f_rw * p->f_rh <= 32768 && p->f_rw <= kBoxMaxRect && p->f_rh <= kBoxMaxRect && !p->knobs.force_general;
        rc = choose_tile(p, g, cap);
        if (rc > 0) { g.uniform = false; rc = choose_tile(p, g, cap); }   // no tile fits the uniform layout
        if (rc) return rc;
    }
    // k_boxsum's band / mask-block height: the shortest of 8, 16, 32 rows whose waves (12 per CU) still fit the chip at once for a
    // workspace of this many frames -- with few frames a wave's march of band + rh - 1 rows IS that kernel's duration
    p->blk_shift = 5;
    if (g.uniform && g.box_rows > 0)
        for (int sh = 3; sh < 5; ++sh)
            if ((long)cap * g.box_parts * ((g.box_rows + (1 << sh) - 1) >> sh) <= 12L * 256) { p->blk_shift = sh; break; }
    size_t hits_cap = std::max<size_t>((size_t)g.npatch * p->n_trees, 1);
    if (hits_cap > 0xffffffffull) return fail(DH_ESIZE, "too many (patch, tree) pairs per frame");
    p->hits_cap = (uint32_t)hits_cap;
#define STEP(x) if (rc == DH_OK) rc = (x)
    STEP(dev_alloc(p, &p->hits, (size_t)cap * hits_cap));
    STEP(dev_alloc(p, &p->hit_box, (size_t)cap * hits_cap));
    STEP(dev_alloc(p, &p->hit_rot, (size_t)cap * hits_cap));
    const bool leaf_hist = p->n_leaves <= p->knobs.leaf_hist_max && !p->knobs.no_leaf_hist;
    // k_region pays a fixed ~20 us (second initial guess, flush, launch) for spreading the first region gather over
    // several workgroups: worth it from ~8 k hit records per frame on the rotation-record path of large forests, from
    // ~65 k with the leaf histogram (measured: config 3 cluster 0.77 -> 0.34 ms; config 5, 38 k records per frame, would lose)
    p->pre_min_hits = p->knobs.region_min_hits > 0 ? (uint32_t)p->knobs.region_min_hits : (leaf_hist ? 65536u : 8192u);
    // (only where a frame of this geometry can plausibly hold that many records: a few per cent of its (window, tree) pairs vote)
    if (!p->knobs.no_region && hits_cap >= (leaf_hist ? 8 : 4) * (size_t)p->pre_min_hits) {
        // k_region serves batches of up to 128 frames (beyond that the (frame, accumulator) workgroups of k_cluster fill the chip themselves)
        p->pre_cap = std::min(cap, 128);
        STEP(dev_alloc(p, &p->pre_region, (size_t)p->pre_cap * 2 * DH_SUPER_CELLS));   // (2 MB per frame; zeroed per batch, before k_region)
    }
    STEP(dev_alloc(p, &p->win_patch, (size_t)cap * std::max(g.win_cap, 1)));
    p->leaf_ls = p->n_leaves <= 65535u ? 1 : 2;
    STEP(dev_alloc(p, &p->win_leaf, ((size_t)cap * std::max(g.win_cap, 1) * p->n_trees) << p->leaf_ls));
    if (!p->knobs.no_tile_list && g.npatch > 0) {
        p->tile_list_stride = (size_t)((cap + 7) / 8) * g.tiles_x * g.tiles_y;
        if (p->tile_list_stride < ((size_t)1 << 31)) STEP(dev_alloc(p, &p->tile_list, (size_t)DH_MAX_CHUNKS * 8 * (p->tile_list_stride + 1)));
    }
    if (g.uniform) {
        const size_t words = (size_t)cap * g.box_rows * ((size_t)g.box_plane << g.swz_log2);
        STEP(dev_alloc(p, &p->box, words));
        // (the slack columns stay 0.  Zero-fills of a new workspace are ordered explicitly: issued on the predictor's stream and
        // waited for below -- the streams here are non-blocking ones, which the legacy stream of a plain hipMemset does not order)
        if (rc == DH_OK && hipMemsetAsync(p->box, 0, words * sizeof(uint32_t), p->own_stream) != hipSuccess) rc = fail(DH_EHIP, "hipMemset(box)");
        if (!p->knobs.box_dense) {
            // zeroed together with the images: "cell non-zero => mask bit set" holds from the start
            const size_t mw = (size_t)cap * ((g.box_rows + 7) / 8) * g.box_parts;        // (sized for 8-row blocks: single-frame workspaces)
            STEP(dev_alloc(p, &p->box_mask, mw));
            if (rc == DH_OK && hipMemsetAsync(p->box_mask, 0, mw * sizeof(unsigned long long), p->own_stream) != hipSuccess) rc = fail(DH_EHIP, "hipMemset(box_mask)");
        }
    }
    if (rc == DH_OK && hipStreamSynchronize(p->own_stream) != hipSuccess) rc = fail(DH_EHIP, "zero-fill of the rectangle-sum images");
    // one block, one memset per batch: hit counters | guess grids | tile flags | window counts | leaf histogram
    const size_t counter_words = (size_t)cap * (1 + DH_POSGRID + DH_GRID3 + (size_t)g.flag_words + (size_t)g.tiles_x * g.tiles_y);
    STEP(dev_alloc(p, &p->counters, counter_words + (leaf_hist ? (size_t)cap * p->n_leaves : 0) + 4));   // (+4: the zero-fill kernel rounds up to 16 bytes)
    if (rc == DH_OK) p->leaf_hits = leaf_hist ? p->counters + counter_words : nullptr;
    p->zero_words = counter_words + (leaf_hist ? (size_t)cap * p->n_leaves : 0);
    // (zeroed once here: the tile flags carry tags and get no fill of their own when k_boxsum clears the other counters)
    if (rc == DH_OK && (hipMemsetAsync(p->counters, 0, (p->zero_words + 4) * sizeof(uint32_t), p->own_stream) != hipSuccess ||
                        hipStreamSynchronize(p->own_stream) != hipSuccess)) rc = fail(DH_EHIP, "zero-fill of the counters");
    p->zero_lo = (size_t)cap * (1 + DH_POSGRID + DH_GRID3); p->zero_hi = p->zero_lo + (size_t)cap * g.flag_words;
    STEP(dev_alloc(p, &p->ws_poses, cap));
    STEP(dev_alloc(p, &p->ws_midp, (size_t)cap * 3));
    STEP(dev_alloc(p, &p->ws_rot, (size_t)cap * 3));
    STEP(dev_alloc(p, &p->ws_mask, cap));
    if (p->debug) {
        STEP(dev_alloc(p, &p->dbg_leaf, (size_t)cap * std::max(g.npatch, 1) * p->n_trees));
        STEP(dev_alloc(p, &p->dbg_flags, (size_t)cap * std::max(g.npatch, 1)));
        STEP(dev_alloc(p, &p->dbg_guess, (size_t)cap * 6));
        STEP(dev_alloc(p, &p->dbg_trace, (size_t)2 * cap * (p->params.meanshift_iterations + 1) * 3));
        STEP(dev_alloc(p, &p->dbg_steps, (size_t)2 * cap));
        STEP(dev_alloc(p, &p->dbg_vcount, 1));
    }
#undef STEP
    if (rc != DH_OK) { free_workspace(p); return rc; }
    const long long nkey = ((long long)g.ss_row << 32) | ((long long)g.top_levels << 24) | ((long long)g.swz_q << 4) | g.swz_log2;
    if (g.npatch > 0 && g.uniform && p->nodes_u_key != nkey) {     // compact nodes carry LDS offsets for this row stride
        hipError_t e = dh_launch_nodes_compact(p->dev, g.ss_row, g.swz_log2, g.swz_q, (uint32_t)(p->f_rw * p->f_rh), p->nodes_u,
                                               p->absorb_ok ? p->nodes_a : nullptr, nullptr, p->amb_list, p->n_amb, p->own_stream);
        if (e == hipSuccess && p->absorb_ok) e = dh_launch_top_build(p->dev, p->nodes_a, p->n_amb, g.top_levels, p->top_tab, p->own_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(p->own_stream);
        if (e != hipSuccess) { free_workspace(p); return fail(DH_EHIP, "k_nodes_compact: %s", hipGetErrorString(e)); }
        p->nodes_u_key = nkey;
    }
    p->geom = g;
    p->cap_frames = cap;
    return DH_OK;
}

static int predictor_set_forking_(dh_predictor *p, int chunks) {
    if (!p) return fail(DH_EINVAL, "NULL predictor");
    if (chunks < 0 || chunks > DH_MAX_CHUNKS) return fail(DH_EINVAL, "dh_predictor_set_forking: chunks = %d, expected 0 .. %d", chunks, DH_MAX_CHUNKS);
    p->chunks = chunks;
    return DH_OK;
}
static int predictor_reserve_(dh_predictor *p, int n, int w, int h) {
    if (!p) return fail(DH_EINVAL, "NULL predictor");
    return reserve(p, n, w, h);
}

// Enqueue the kernels (k_boxsum / k_pixflags, k_traverse, k_emit, k_vote, [k_region,] k_cluster) for frames [f0, f0 + n) of the batch on stream s.
static int enqueue_range(dh_predictor *p, const uint16_t *frames, int f0, int n, int w, int h, const float K[9],
                         const float kinv[9], const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask,
                         dh_pose *out, hipStream_t s, bool profile, int32_t *leaf_out = nullptr, uint8_t *flags_out = nullptr,
                         bool traverse_only = false, int chunk = 0, bool zero_fold = false) {
    const Geom &g = p->geom;
    uint32_t *gen = p->gen + chunk;      // this kernel sequence's tile-flag tag
    uint32_t *hit_count = p->counters + f0;
    uint32_t *pos_grid = p->counters + p->cap_frames + (size_t)f0 * DH_POSGRID;
    uint32_t *rot_grid = p->counters + p->cap_frames + (size_t)p->cap_frames * DH_POSGRID + (size_t)f0 * DH_GRID3;
    const size_t hoff = (size_t)f0 * p->hits_cap;
    const uint16_t *fr = frames + (size_t)f0 * w * h;
    char batch_name[48];
    snprintf(batch_name, sizeof batch_name, "dh:batch n=%d %dx%d", n, w, h);
    Range batch_range(profile, batch_name);
    if (profile) HIP_TRY(hipEventRecord(p->ev[0], s));
    uint32_t *box = g.uniform ? p->box + (size_t)f0 * g.box_rows * ((size_t)g.box_plane << g.swz_log2) : nullptr;
    // tile flags: one byte per tile, frames packed back to back (the slice always starts at the frame the memset covered)
    uint8_t *tile_flags = (uint8_t *)(p->counters + (size_t)p->cap_frames * (1 + DH_POSGRID + DH_GRID3)) + (size_t)f0 * g.tiles_x * g.tiles_y;
    if (g.npatch > 0 && g.uniform) {
        BoxArgs ba{};
        ba.frames = fr; ba.zeros = p->zeros; ba.n_frames = n; ba.w = w; ba.h = h; ba.rw = p->f_rw; ba.rh = p->f_rh;
        ba.tile_flags = tile_flags; ba.tiles_x = g.tiles_x; ba.tiles_y = g.tiles_y; ba.gen = gen;
        if (zero_fold) { ba.zero_ptr = p->counters; ba.zero_lo = (uint32_t)p->zero_lo; ba.zero_hi = (uint32_t)p->zero_hi; ba.zero_end = (uint32_t)p->zero_words; }
        ba.tpx = g.px * (int)p->params.stepwidth; ba.tpy = g.py * (int)p->params.stepwidth;
        ba.tbw = (g.px - 1) * (int)p->params.stepwidth + (int)p->params.subimage_width - p->f_rw + 1;
        ba.tbh = (g.py - 1) * (int)p->params.stepwidth + (int)p->params.subimage_height - p->f_rh + 1;
        ba.out = box; ba.plane = g.box_plane; ba.rows = g.box_rows; ba.lg = g.swz_log2;
        ba.ow = g.box_ow; ba.oh = g.box_oh; ba.parts = g.box_parts; ba.bands = g.box_bands;
        // LDS-ring instance (each pixel read once): 4 waves x (rh - 1) packed rows per workgroup, so about
        // 12 waves fit a CU; the bands are made as tall as keeps the whole launch resident at once
        const bool ring_on = !p->knobs.box_no_ring;
        if (ring_on && p->f_rh >= 2 && p->f_rh - 1 <= 28) {      // 4 x 28 x 512 B + the prefix rows < 64 KB
            // (a single-frame workspace: 8-row bands and mask blocks -- a wave's march of band + rh - 1 rows IS the kernel's
            // duration there: 31 instead of 55 rows at rh = 24; one 320 x 240 frame 86 -> 79 us with 16-row bands, 76 with 8)
            const int blk = 1 << p->blk_shift;
            int oh = 0;
            // workgroups the chip holds at once: per CU what the ring's LDS leaves room for (53 KB at rh = 24: three), at most the six
            // that 76 VGPRs allow; 256 CUs
            const size_t lds_wg = (size_t)4 * (p->f_rh - 1) * 64 * 8 + (size_t)4 * (256 + 96 + 8) * 4;
            const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(6, ((size_t)160 * 1024) / lds_wg));
            int bands = dh_box_bands_(n, g.box_parts, g.box_rows, blk, p->f_rh, wg_per_cu * 256, &oh);
            if (p->knobs.box_bands > 0) {                     // DH_BOX_BANDS: experiments
                bands = std::min(p->knobs.box_bands, std::max(1, (g.box_rows + blk - 1) / blk));
                oh = ((g.box_rows + bands - 1) / bands + blk - 1) & ~(blk - 1);
            }
            ba.oh = oh;                                       // whole mask blocks per band (BoxArgs::blk_mask)
            ba.bands = (g.box_rows + ba.oh - 1) / ba.oh;
            ba.ring = 1;
        }
        ba.blk_shift = p->blk_shift;
        ba.mask_blocks = (g.box_rows + (1 << p->blk_shift) - 1) >> p->blk_shift;
        ba.blk_mask = p->box_mask ? p->box_mask + (size_t)f0 * ba.mask_blocks * g.box_parts : nullptr;
        ba.blocks_per_frame = (ba.parts * ba.bands + 3) / 4;
        { Range r(profile, "dh:boxsum"); HIP_TRY(dh_launch_boxsum(ba, s)); }
    }
    if (g.npatch > 0 && !g.uniform) {
        PixFlagArgs fa{};
        fa.frames = fr; fa.n_frames = n; fa.w = w; fa.h = h; fa.tile_flags = tile_flags; fa.gen = gen;
        fa.tiles_x = g.tiles_x; fa.tiles_y = g.tiles_y;
        fa.tpx = g.px * (int)p->params.stepwidth; fa.tpy = g.py * (int)p->params.stepwidth;
        fa.tfw = (g.px - 1) * (int)p->params.stepwidth + (int)p->params.subimage_width;
        fa.tfh = (g.py - 1) * (int)p->params.stepwidth + (int)p->params.subimage_height;
        { Range r(profile, "dh:pixflags"); HIP_TRY(dh_launch_pixflags(fa, s)); }
    }
    // product mode: the flagged tiles as compact lists, so that the workgroups of empty tiles sit at the end of k_traverse's grid
    // (with the taps on, every tile position keeps its workgroup: those of empty tiles write the taps' "background")
    // (the list moves the empty tiles' workgroups behind the flagged ones; a batch whose tiles all fit the chip's 512 workgroup
    // slots at once gains nothing from that and saves the dispatch: 3 us of a single frame's 96)
    const bool use_list = p->tile_list && g.npatch > 0 && !leaf_out && !flags_out && !p->debug && (long)n * g.tiles_x * g.tiles_y > 512;
    uint32_t *tl_list = use_list ? p->tile_list + (size_t)chunk * 8 * (p->tile_list_stride + 1) : nullptr;
    uint32_t *tl_count = use_list ? tl_list + 8 * p->tile_list_stride : nullptr;
    if (use_list) { Range r(profile, "dh:tile_list"); HIP_TRY(dh_launch_tile_list(tile_flags, gen, n, g.tiles_x * g.tiles_y, tl_list, tl_count, (uint32_t)p->tile_list_stride, s)); }
    if (profile) HIP_TRY(hipEventRecord(p->ev[4], s));
    if (g.npatch > 0) {
        TraverseArgs ta{};
        ta.frames = fr; ta.n_frames = n; ta.w = w; ta.h = h;
        ta.step = (int)p->params.stepwidth; ta.sw = (int)p->params.subimage_width; ta.sh = (int)p->params.subimage_height;
        ta.nx = g.nx; ta.ny = g.ny; ta.px = g.px; ta.py = g.py; ta.tiles_x = g.tiles_x; ta.tiles_y = g.tiles_y;
        ta.ss_max = g.ss_max; ta.ss_row = g.ss_row; ta.swz_log2 = g.swz_log2; ta.swz_q = g.swz_q;
        ta.uniform = g.uniform ? 1 : 0; ta.rw = p->f_rw; ta.rh = p->f_rh; ta.area = (uint32_t)(p->f_rw * p->f_rh);
        ta.nodes_u = p->nodes_u; ta.nodes_a = p->absorb_ok ? p->nodes_a : nullptr; ta.walk_lb = (p->n_nodes + p->n_amb) << 4; ta.amb_list = p->amb_list; ta.top_tab = p->top_tab; ta.top_levels = g.top_levels; ta.nodes_g = p->knobs.no_general_int ? nullptr : p->nodes_g;
        ta.box = box; ta.box_plane = g.box_plane; ta.box_rows = g.box_rows;
        ta.tile_flags = tile_flags; ta.gen = gen;
#ifdef DH_PROFILING_KNOBS
        ta.stop_phase = p->knobs.trav_stop;
        static unsigned long long *stamps = nullptr;
        if (p->knobs.trav_stamps) {
            if (!stamps) { HIP_TRY(hipMalloc((void **)&stamps, 64)); HIP_TRY(hipMemset(stamps, 0, 64)); }
            else {
                unsigned long long hst[8];
                HIP_TRY(hipMemcpy(hst, stamps, 64, hipMemcpyDeviceToHost));
                fprintf(stderr, "[k_traverse cycles/phase summed over the workgroups that reach it] region=%llu gate=%llu walks=%llu\n", hst[0], hst[1], hst[2]);
                HIP_TRY(hipMemset(stamps, 0, 64));
            }
            ta.dbg_stamps = stamps;
        }
#endif
        ta.f = p->dev;
        const int tiles = g.tiles_x * g.tiles_y;
        uint32_t *win_count = p->counters + (size_t)p->cap_frames * (1 + DH_POSGRID + DH_GRID3 + (size_t)g.flag_words) + (size_t)f0 * tiles;
        ta.win_count = win_count; ta.win_cap = g.win_cap;
        ta.win_patch = p->win_patch + (size_t)f0 * g.win_cap; ta.win_leaf = p->win_leaf + (((size_t)f0 * g.win_cap * p->n_trees) << p->leaf_ls); ta.leaf_ls = p->leaf_ls;
        ta.dbg_leaf = leaf_out ? leaf_out : p->debug ? p->dbg_leaf + (size_t)f0 * g.npatch * p->n_trees : nullptr;
        ta.dbg_flags = flags_out ? flags_out : p->debug ? p->dbg_flags + (size_t)f0 * g.npatch : nullptr;
        if (use_list) { ta.tile_list = tl_list; ta.tile_list_count = tl_count; ta.tile_list_stride = (uint32_t)p->tile_list_stride; }
        { Range r(profile, "dh:traverse"); HIP_TRY(dh_launch_traverse(ta, g.lds, s)); }
        if (profile) HIP_TRY(hipEventRecord(p->ev[5], s));
        if (!traverse_only) {
            EmitArgs ea{};
            ea.frames = fr; ea.n_frames = n; ea.w = w; ea.h = h;
            ea.step = ta.step; ea.lw = ta.sw / 2; ea.lh = ta.sh / 2; ea.nx = g.nx; ea.npatch = g.npatch;
            ea.px = g.px; ea.py = g.py; ea.tiles = tiles;
            memcpy(ea.kinv, kinv, 9 * sizeof(float));
            ea.f = p->dev;
            ea.win_count = win_count; ea.win_patch = ta.win_patch; ea.win_leaf = ta.win_leaf; ea.leaf_ls = p->leaf_ls; ea.win_cap = g.win_cap;
            ea.hits = p->hits + hoff; ea.hit_box = p->hit_box + hoff; ea.hit_rot = p->hit_rot + hoff;
            ea.hit_count = hit_count; ea.hits_cap = p->hits_cap;
            ea.leaf_hits = p->leaf_hits ? p->leaf_hits + (size_t)f0 * p->n_leaves : nullptr;
            ea.gen = gen;
            ea.dbg_flags = ta.dbg_flags;
#ifdef DH_PROFILING_KNOBS
            ea.stop = p->knobs.emit_stop;
#endif
            { Range r(profile, "dh:emit"); HIP_TRY(dh_launch_emit(ea, s)); }
        }
    } else if (profile) HIP_TRY(hipEventRecord(p->ev[5], s));
    if (profile) HIP_TRY(hipEventRecord(p->ev[1], s));
    if (traverse_only) return DH_OK;
    {
        VoteArgs va{};
        va.n_frames = n; va.w = w; va.h = h; va.cell_fast = p->knobs.vote_exact ? 0 : 1;
        memcpy(va.k, K, sizeof va.k);
        va.f = p->dev; va.hits = p->hits + hoff; va.hit_box = p->hit_box + hoff; va.hit_rot = p->hit_rot + hoff;
        va.hit_count = hit_count; va.hits_cap = p->hits_cap;
        va.pos_grid = pos_grid; va.rot_grid = rot_grid;
        va.leaf_hits = p->leaf_hits ? p->leaf_hits + (size_t)f0 * p->n_leaves : nullptr;
#ifdef DH_PROFILING_KNOBS
        va.stop = p->knobs.vote_stop;
#endif
        { Range r(profile, "dh:vote"); HIP_TRY(dh_launch_vote(va, s)); }
    }
    if (profile) HIP_TRY(hipEventRecord(p->ev[2], s));
    {
        ClusterArgs ca{};
        ca.frames = fr; ca.n_frames = n; ca.w = w; ca.h = h;
        memcpy(ca.kinv, kinv, 9 * sizeof(float));
        ca.f = p->dev; ca.hits = p->hits + hoff; ca.hit_box = p->hit_box + hoff; ca.hit_rot = p->hit_rot + hoff;
        ca.hit_count = hit_count; ca.hits_cap = p->hits_cap;
        ca.leaf_hits = p->leaf_hits ? p->leaf_hits + (size_t)f0 * p->n_leaves : nullptr;
        ca.pos_grid = pos_grid; ca.rot_grid = rot_grid; ca.kern_r2 = p->kern_r2;
        ca.iterations = p->params.meanshift_iterations;
#ifdef DH_PROFILING_KNOBS
        ca.stop = p->knobs.cl_stop;
        static unsigned long long *cl_stamps = nullptr;
        if (p->knobs.cl_stamps) {
            if (!cl_stamps) { HIP_TRY(hipMalloc((void **)&cl_stamps, 128)); HIP_TRY(hipMemset(cl_stamps, 0, 128)); }
            else {
                unsigned long long hst[16];
                HIP_TRY(hipMemcpy(hst, cl_stamps, 128, hipMemcpyDeviceToHost));
                fprintf(stderr, "[k_cluster rotation workgroups, cycles summed] guess=%llu zero=%llu gather=%llu | sums: count sweep=%llu (4)=%llu scan+park=%llu (6)=%llu chain=%llu update=%llu | sums=%llu workgroups=%llu\n",
                        hst[0], hst[1], hst[2], hst[3], hst[4], hst[5], hst[6], hst[7], hst[8], hst[14], hst[15]);
                HIP_TRY(hipMemset(cl_stamps, 0, 128));
            }
            ca.dbg_stamps = cl_stamps;
        }
#endif
        ca.midp_guess = midp_guess ? midp_guess + (size_t)f0 * 3 : nullptr;
        ca.rot_guess = rot_guess ? rot_guess + (size_t)f0 * 3 : nullptr;
        ca.guess_mask = guess_mask ? guess_mask + f0 : nullptr;
        ca.out = out + f0;
        if (p->debug) { ca.dbg_guess = p->dbg_guess; ca.dbg_trace = p->dbg_trace; ca.dbg_steps = p->dbg_steps; }
        // few frames with many hit records each: the first region of every accumulator is gathered by several workgroups
        const int slices = std::min(16, 256 / std::max(n, 1));
        if (p->pre_region && slices >= 2 && f0 + n <= p->pre_cap && ca.iterations > 0) {
            ca.pre_region = p->pre_region + (size_t)f0 * 2 * DH_SUPER_CELLS;
            ca.pre_slices = slices;
            ca.pre_min_hits = p->pre_min_hits;
            // (a kernel node inside a captured graph, like the counters' fill)
            const size_t pre_bytes = (size_t)n * 2 * DH_SUPER_CELLS * sizeof(uint32_t);
            if (p->capturing) HIP_TRY(dh_launch_zero(ca.pre_region, pre_bytes, s));
            else HIP_TRY(hipMemsetAsync(ca.pre_region, 0, pre_bytes, s));
            { Range r(profile, "dh:region"); HIP_TRY(dh_launch_region(ca, s)); }
        }
        { Range r(profile, "dh:cluster"); HIP_TRY(dh_launch_cluster(ca, s)); }
    }
    if (profile) { HIP_TRY(hipEventRecord(p->ev[3], s)); p->ev_valid = true; }
    return DH_OK;
}

// Largest number of frames whose hit records are resident at once (64 B x window positions x trees
// per frame: 9 MB per frame at BASELINE config 2).  Larger batches are walked in slices on the same
// stream, reusing the workspace.
static int max_resident_frames(const dh_predictor *p) { return p->knobs.max_resident; }

static int predict_batch_device_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                                       const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask,
                                       dh_pose *out, void *stream_) {
    if (!p || !frames || !K || !out) return fail(DH_EINVAL, "dh_predict_batch_device: NULL argument");
    if (n == 0) return DH_OK;
    if (n < 0) return fail(DH_EINVAL, "negative batch size");
    DeviceGuard guard(p->device);
    if (!guard.ok) return fail(DH_EHIP, "cannot select device %d", p->device);
    const int slice = p->debug ? n : std::min(n, max_resident_frames(p));   // the taps index the whole batch
    int rc = reserve(p, slice, w, h);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream_;
    float kinv[9];
    dh_mat3_inv_f32_(K, kinv);   // cached in a RefCell by the reference (types.rs:436-441)
    for (int f0 = 0; f0 < n; f0 += slice) {
        const int m = std::min(slice, n - f0);
        const uint16_t *fr = frames + (size_t)f0 * w * h;
        const float *mg = midp_guess ? midp_guess + (size_t)f0 * 3 : nullptr;
        const double *rg = rot_guess ? rot_guess + (size_t)f0 * 3 : nullptr;
        const uint8_t *gm = guess_mask ? guess_mask + f0 : nullptr;
        // Forked sub-batches: two halves of the slice on two streams let the latency-bound tail kernels of one half run
        // beside the head kernels of the other.  Measured on MI355X: it pays once each half still has >= 256 frames
        // (512 frames per call: 479 k -> 533 k frames/s with 2 chunks, 483 k with 4; 256 frames per call: no gain),
        // so that is the automatic choice; DH_CHUNKS forces a count.
        int chunks = p->chunks > 0 ? p->chunks : (m >= 512 ? 2 : 1);
        if (p->profiling || p->debug || p->capturing || m < 2 * DH_MIN_CHUNK_FRAMES) chunks = 1;
        chunks = std::min(chunks, m / DH_MIN_CHUNK_FRAMES);
        // The per-batch counters: one kernel sequence on the uniform path has them cleared by its first kernel (k_boxsum: every
        // wave of its grid takes a share; the tile flags carry tags and need no fill) -- one dispatch less per batch, 4.7 us of a
        // single frame's 95.  Forked sub-batches, the general path and frames smaller than the patch keep the fill.
        const bool fold = chunks <= 1 && p->geom.uniform && p->geom.npatch > 0 && p->zero_words < 0xffffffffull && !p->knobs.no_zero_fold;
        if (!fold) {
            // (inside a captured graph the zero-fill is a kernel node: see k_zero)
            if (p->capturing) HIP_TRY(dh_launch_zero(p->counters, (p->zero_words * sizeof(uint32_t) + 15) & ~(size_t)15, s));
            else HIP_TRY(hipMemsetAsync(p->counters, 0, p->zero_words * sizeof(uint32_t), s));
        }
        if (chunks <= 1) {
            rc = enqueue_range(p, fr, 0, m, w, h, K, kinv, mg, rg, gm, out + f0, s, p->profiling, nullptr, nullptr, false, 0, fold);
            if (rc) return rc;
        } else {
            HIP_TRY(hipEventRecord(p->ev_fork, s));
            for (int c = 0; c < chunks; ++c) {
                const int c0 = (int)((long long)m * c / chunks), c1 = (int)((long long)m * (c + 1) / chunks);
                hipStream_t cs = c == 0 ? s : p->aux_stream[c - 1];
                if (c > 0) HIP_TRY(hipStreamWaitEvent(cs, p->ev_fork, 0));
                rc = enqueue_range(p, fr, c0, c1 - c0, w, h, K, kinv, mg, rg, gm, out + f0, cs, false, nullptr, nullptr, false, c);
                if (rc) return rc;
                if (c > 0) {
                    HIP_TRY(hipEventRecord(p->ev_join[c - 1], cs));
                    HIP_TRY(hipStreamWaitEvent(s, p->ev_join[c - 1], 0));
                }
            }
        }
    }
    p->last_n = std::min(n, slice);   // the taps describe the last resident slice
    p->last_frames = frames;
    p->dbg_valid = p->debug;
    return DH_OK;
}

static int stage_frames(dh_predictor *p, const uint16_t *frames, int n, int w, int h);
static int ensure_frame_staging(dh_predictor *p, size_t fbytes);

template <typename T>
static int grow_pinned(T **buf, size_t *cap, size_t need) {
    if (need <= *cap) return DH_OK;
    if (*buf) (void)hipHostFree(*buf);
    *buf = nullptr; *cap = 0;
    const size_t want = need + need / 4 + 4096;
    void *q = nullptr;
    hipError_t e = hipHostMalloc(&q, want * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) return fail(DH_ENOMEM, "hipHostMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e));
    *buf = (T *)q; *cap = want;
    return DH_OK;
}
template <typename T>
static int grow_device(T **buf, size_t *cap, size_t need) {
    if (need <= *cap) return DH_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *cap = 0;
    const size_t want = need + need / 4 + 4096;
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, want * sizeof(T));
    if (e != hipSuccess) return fail(DH_ENOMEM, "hipMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e));
    *buf = (T *)q; *cap = want;
    return DH_OK;
}

// The small host arrays of a slice (poses out; guesses and their mask in) cross PCIe through the predictor's OWN page-locked
// staging block: the caller's arrays are ordinary pageable memory a few hundred bytes long, typically sharing their pages
// with other allocations of other host threads, and handing such ranges to the runtime's pageable-copy path (pin the pages,
// DMA, unpin -- per copy, beside other threads doing the same to neighbouring bytes) is both slower than a pinned copy and
// the one place where concurrent predictors met inside the runtime.  Layout for m frames: poses | rot guesses | mid guesses | mask.
struct SmallStage {
    dh_pose *poses; double *rot; float *midp; uint8_t *mask;
};
static int small_stage(dh_predictor *p, int m, SmallStage *st) {
    const size_t need = (size_t)m * (sizeof(dh_pose) + 3 * sizeof(double) + 3 * sizeof(float) + 1) + 64;
    int rc = grow_pinned(&p->pin_small, &p->pin_small_cap, need);      // (only between slices: both streams are idle)
    if (rc) return rc;
    st->poses = (dh_pose *)p->pin_small;
    st->rot = (double *)(st->poses + m);
    st->midp = (float *)(st->rot + (size_t)m * 3);
    st->mask = (uint8_t *)(st->midp + (size_t)m * 3);
    return DH_OK;
}
// guesses of frames [f0, f0 + m) -> staging -> the workspace's device arrays, on stream s
static int upload_guesses(dh_predictor *p, const SmallStage &st, int f0, int m, const float *midp_guess, const double *rot_guess,
                          const uint8_t *guess_mask, hipStream_t s) {
    if (midp_guess) {
        memcpy(st.midp, midp_guess + (size_t)f0 * 3, (size_t)m * 3 * sizeof(float));
        HIP_TRY(hipMemcpyAsync(p->ws_midp, st.midp, (size_t)m * 3 * sizeof(float), hipMemcpyHostToDevice, s));
    }
    if (rot_guess) {
        memcpy(st.rot, rot_guess + (size_t)f0 * 3, (size_t)m * 3 * sizeof(double));
        HIP_TRY(hipMemcpyAsync(p->ws_rot, st.rot, (size_t)m * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    if (guess_mask) {
        memcpy(st.mask, guess_mask + f0, (size_t)m);
        HIP_TRY(hipMemcpyAsync(p->ws_mask, st.mask, (size_t)m, hipMemcpyHostToDevice, s));
    }
    return DH_OK;
}
// poses of the slice: device -> staging on s, wait, -> the caller's array
static int download_poses(dh_predictor *p, const SmallStage &st, int m, dh_pose *out, hipStream_t s) {
    HIP_TRY(hipMemcpyAsync(st.poses, p->ws_poses, (size_t)m * sizeof(dh_pose), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    memcpy(out, st.poses, (size_t)m * sizeof(dh_pose));
    return DH_OK;
}

// Host entry point.  The frames cross PCIe in chunks on copy_stream while the kernels of the previous chunk run on
// own_stream (the path is PCIe-bound: 614 KB per frame in, 40 bytes out), so a batch takes about its upload time plus
// the kernels of the last chunk.  The poses of a slice come back in one copy.
static int predict_batch_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                                const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask, dh_pose *out) {
    if (!p || !frames || !K || !out) return fail(DH_EINVAL, "dh_predict_batch: NULL argument");
    if (n == 0) return DH_OK;
    if (n < 0) return fail(DH_EINVAL, "negative batch size");
    HIP_TRY(hipSetDevice(p->device));
    const int slice = p->debug ? n : std::min(n, max_resident_frames(p));
    hipStream_t s = p->own_stream, cs = p->copy_stream;
    const size_t fpx = (size_t)w * h;
    for (int f0 = 0; f0 < n; f0 += slice) {      // host batches are staged slice by slice
        const int m = std::min(slice, n - f0);
        int rc = reserve(p, m, w, h);
        if (rc) return rc;
        rc = ensure_frame_staging(p, (size_t)m * fpx * sizeof(uint16_t));
        if (rc) return rc;
        SmallStage st;
        rc = small_stage(p, m, &st);
        if (rc == DH_OK) rc = upload_guesses(p, st, f0, m, midp_guess, rot_guess, guess_mask, s);
        if (rc) return rc;
        // the parity taps describe ONE device batch: with them on, the slice is a single chunk
        int cstart[DH_STAGE_EVENTS + 1];
        const int nchunks = dh_chunk_plan_(m, p->knobs.stage_chunk, p->debug, cstart);
        auto predict_chunk = [&](int c0, int cm) {
            return dh_predict_batch_device(p, p->ws_frames + (size_t)c0 * fpx, cm, w, h, K, midp_guess ? p->ws_midp + (size_t)c0 * 3 : nullptr,
                                           rot_guess ? p->ws_rot + (size_t)c0 * 3 : nullptr, guess_mask ? p->ws_mask + c0 : nullptr, p->ws_poses + c0, s);
        };
        if (nchunks == 1) {
            // latency path (single frames, small batches): one copy on the compute stream itself
            HIP_TRY(hipMemcpyAsync(p->ws_frames, frames + (size_t)f0 * fpx, (size_t)m * fpx * sizeof(uint16_t), hipMemcpyHostToDevice, s));
            rc = predict_chunk(0, m);
            if (rc) { (void)hipStreamSynchronize(s); return rc; }
        } else {
            // Chunk k + 1 is uploaded on copy_stream while the kernels of chunk k run on own_stream.  From page-locked host
            // memory (dh_host_alloc, or any buffer the caller registered with HIP) the copies are true asynchronous DMA and
            // the whole batch takes its PCIe time; from pageable memory each copy is issued synchronously (the runtime pins,
            // copies, unpins), which still overlaps the kernels but not the next chunk's preparation.
            // (Measured and rejected: three host threads issuing the pageable chunk copies on three streams -- 60-82 k frames/s,
            // erratic, against 80 k from this one thread: the pinning of the pages serialises in the driver.)
            HIP_TRY(hipStreamWaitEvent(cs, p->ev_slice, 0));          // the previous slice's kernels have read the staging buffer
            for (int k = 0; k < nchunks; ++k) {
                const int c0 = cstart[k], cm = cstart[k + 1] - c0;
                HIP_TRY(hipMemcpyAsync(p->ws_frames + (size_t)c0 * fpx, frames + (size_t)(f0 + c0) * fpx, (size_t)cm * fpx * sizeof(uint16_t), hipMemcpyHostToDevice, cs));
                HIP_TRY(hipEventRecord(p->ev_stage[k], cs));
                HIP_TRY(hipStreamWaitEvent(s, p->ev_stage[k], 0));
                rc = predict_chunk(c0, cm);
                if (rc) { (void)hipStreamSynchronize(cs); (void)hipStreamSynchronize(s); return rc; }
            }
        }
        HIP_TRY(hipEventRecord(p->ev_slice, s));
        rc = download_poses(p, st, m, out + f0, s);
        if (rc) return rc;
    }
    return DH_OK;
}

// Page-locked host memory for frame buffers: uploads from it are asynchronous DMA at PCIe speed.
static int host_alloc_(size_t bytes, void **out) {
    if (!out) return fail(DH_EINVAL, "dh_host_alloc: NULL argument");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) { *out = nullptr; return fail(DH_ENOMEM, "hipHostMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); }
    return DH_OK;
}
static int host_free_(void *ptr) {
    if (ptr && hipHostFree(ptr) != hipSuccess) return fail(DH_EHIP, "hipHostFree");
    return DH_OK;
}

// ------------------------------------------------------------------ run-length coded input (BIWI `.bin`, biwi.rs:81-103)
// Validates every payload (dh_rle_plan_), packs blob + run table into pinned memory (dh_rle_pack_) and sizes the device
// buffers.  Nothing has been launched when this fails.
static_assert(sizeof(DhRun) == sizeof(uint2), "run table entries are uint2 on the device");
static int rle_prepare(dh_predictor *p, const uint8_t *const *bufs, const size_t *lens, int n, RlePlan &plan) {
    int rc = dh_rle_plan_(bufs, lens, n, p->knobs.host_threads, plan);
    if (rc) return rc;
    const size_t blob_bytes = plan.blob_off[n];
    rc = grow_pinned(&p->pin_begin, &p->pin_begin_cap, (size_t)n + 1);
    if (rc == DH_OK) rc = grow_pinned(&p->pin_blob, &p->pin_blob_cap, blob_bytes);
    if (rc == DH_OK) rc = grow_pinned(&p->pin_runs, &p->pin_runs_cap, std::max<size_t>(plan.nruns, 1));
    if (rc == DH_OK) rc = grow_device(&p->dev_blob, &p->dev_blob_cap, blob_bytes);
    if (rc == DH_OK) rc = grow_device(&p->dev_runs, &p->dev_runs_cap, std::max<size_t>(plan.nruns, 1));
    if (rc == DH_OK) rc = grow_device(&p->dev_begin, &p->dev_begin_cap, (size_t)n + 1);
    if (rc) return rc;
    memcpy(p->pin_begin, plan.run_begin.data(), ((size_t)n + 1) * sizeof(uint32_t));
    dh_rle_pack_(bufs, lens, n, p->knobs.host_threads, plan, p->pin_blob, (DhRun *)p->pin_runs);
    return DH_OK;
}

// Uploads chunk [c0, c0 + cm) of the prepared batch on the copy stream and enqueues its decode into `frames_dev`
// (frame c0 first) on stream s.
static int rle_upload_decode(dh_predictor *p, const std::vector<size_t> &blob_off, int c0, int cm, int ci, uint32_t W, uint32_t H,
                             uint16_t *frames_dev, hipStream_t s) {
    hipStream_t cs = p->copy_stream;
    const size_t b0 = blob_off[c0], b1 = blob_off[c0 + cm];
    const uint32_t r0 = p->pin_begin[c0], r1 = p->pin_begin[c0 + cm];
    HIP_TRY(hipMemcpyAsync(p->dev_blob + b0, p->pin_blob + b0, b1 - b0, hipMemcpyHostToDevice, cs));
    if (r1 > r0) HIP_TRY(hipMemcpyAsync(p->dev_runs + r0, p->pin_runs + r0, (size_t)(r1 - r0) * sizeof(uint2), hipMemcpyHostToDevice, cs));
    hipEvent_t ev = p->ev_stage[ci % DH_STAGE_EVENTS];
    HIP_TRY(hipEventRecord(ev, cs));
    HIP_TRY(hipStreamWaitEvent(s, ev, 0));
    HIP_TRY(hipMemsetAsync(frames_dev, 0, (size_t)cm * W * H * sizeof(uint16_t), s));          // the empty runs (biwi.rs:90-93)
    RleArgs ra{};
    ra.blob = (const uint16_t *)p->dev_blob; ra.runs = p->dev_runs; ra.run_begin = p->dev_begin + c0;
    // the run table addresses pixels of the whole batch: frame c0 of the chunk is pixel c0 * W * H there
    ra.frames = frames_dev - (size_t)c0 * W * H; ra.n_frames = cm;
    const uint32_t per_frame = (r1 - r0 + (uint32_t)cm - 1) / (uint32_t)cm;
    ra.blocks_per_frame = (int)std::max(1u, std::min(64u, (per_frame + 15) / 16));
    HIP_TRY(dh_launch_rle_decode(ra, s));
    return DH_OK;
}

static int biwi_decode_depth_device_(dh_predictor *p, const uint8_t *const *bufs, const size_t *lens, int n, uint16_t *frames_dev,
                                           size_t cap_px, uint32_t *w, uint32_t *h) {
    if (!p || !bufs || !lens || !w || !h) return fail(DH_EINVAL, "dh_biwi_decode_depth_device: NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->own_stream));             // the pinned staging buffers are free
    HIP_TRY(hipStreamSynchronize(p->copy_stream));
    RlePlan plan;
    int rc = rle_prepare(p, bufs, lens, n, plan);
    if (rc) return rc;
    const std::vector<size_t> &blob_off = plan.blob_off;
    const uint32_t W = plan.W, H = plan.H;
    *w = W; *h = H;
    if (!frames_dev) return DH_OK;                            // size query (validates as well)
    if (cap_px < (size_t)n * W * H) return fail(DH_EINVAL, "output holds %zu pixels, the batch has %zu", cap_px, (size_t)n * W * H);
    HIP_TRY(hipMemcpyAsync(p->dev_begin, p->pin_begin, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, p->own_stream));
    const int chunk = std::min(n, p->knobs.stage_chunk * 2);
    int ci = 0;
    for (int c0 = 0; c0 < n; c0 += chunk, ++ci) {
        const int cm = std::min(chunk, n - c0);
        rc = rle_upload_decode(p, blob_off, c0, cm, ci, W, H, frames_dev + (size_t)c0 * W * H, p->own_stream);
        if (rc) { (void)hipStreamSynchronize(p->copy_stream); (void)hipStreamSynchronize(p->own_stream); return rc; }
    }
    HIP_TRY(hipStreamSynchronize(p->own_stream));
    return DH_OK;
}

static int predict_batch_rle_(dh_predictor *p, const uint8_t *const *bufs, const size_t *lens, int n, const float K[9],
                                    const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask, dh_pose *out) {
    if (!p || !bufs || !lens || !K || !out) return fail(DH_EINVAL, "dh_predict_batch_rle: NULL argument");
    if (n == 0) return DH_OK;
    if (n < 0) return fail(DH_EINVAL, "negative batch size");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = p->own_stream;
    const int slice = p->debug ? n : std::min(n, max_resident_frames(p));
    for (int f0 = 0; f0 < n; f0 += slice) {
        const int m = std::min(slice, n - f0);
        HIP_TRY(hipStreamSynchronize(s));                         // pinned staging and device blob of the previous slice are free
        HIP_TRY(hipStreamSynchronize(p->copy_stream));
        RlePlan plan;
        int rc = rle_prepare(p, bufs + f0, lens + f0, m, plan);   // validates: nothing launched on failure
        if (rc) return rc;
        const std::vector<size_t> &blob_off = plan.blob_off;
        const uint32_t W = plan.W, H = plan.H;
        const int w = (int)W, h = (int)H;
        rc = reserve(p, m, w, h);
        if (rc == DH_OK) rc = ensure_frame_staging(p, (size_t)m * W * H * sizeof(uint16_t));
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(p->dev_begin, p->pin_begin, ((size_t)m + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        SmallStage st;
        rc = small_stage(p, m, &st);
        if (rc == DH_OK) rc = upload_guesses(p, st, f0, m, midp_guess, rot_guess, guess_mask, s);
        if (rc) return rc;
        const int chunk = p->debug ? m : std::min(m, p->knobs.stage_chunk * 2);   // compressed chunks are small: twice the raw chunk
        int ci = 0;
        for (int c0 = 0; c0 < m; c0 += chunk, ++ci) {
            const int cm = std::min(chunk, m - c0);
            uint16_t *fr = p->ws_frames + (size_t)c0 * W * H;
            rc = rle_upload_decode(p, blob_off, c0, cm, ci, W, H, fr, s);
            if (rc == DH_OK)
                rc = dh_predict_batch_device(p, fr, cm, w, h, K, midp_guess ? p->ws_midp + (size_t)c0 * 3 : nullptr,
                                             rot_guess ? p->ws_rot + (size_t)c0 * 3 : nullptr, guess_mask ? p->ws_mask + c0 : nullptr, p->ws_poses + c0, s);
            if (rc) { (void)hipStreamSynchronize(p->copy_stream); (void)hipStreamSynchronize(s); return rc; }
        }
        rc = download_poses(p, st, m, out + f0, s);
        if (rc) return rc;
    }
    return DH_OK;
}

// ------------------------------------------------------------------ hipGraph capture of one batch
// For launch-bound use (small frames / single frames, BASELINE config 5): the zero-fill and the kernel
// launches (k_boxsum ... k_cluster) of one dh_predict_batch_device call are captured once and replayed with one host call.
static int graph_destroy_(dh_predictor *p) {
    if (!p) return fail(DH_EINVAL, "NULL predictor");
    drop_graph(p);
    p->graph_stale = false;
    return DH_OK;
}

static int graph_capture_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                                const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask, dh_pose *out) {
    if (!p || !frames || !K || !out) return fail(DH_EINVAL, "dh_graph_capture: NULL argument");
    if (n <= 0) return fail(DH_EINVAL, "batch size must be positive");
    if (p->debug || p->profiling) return fail(DH_ESTATE, "taps / profiling cannot be captured");
    HIP_TRY(hipSetDevice(p->device));
    int rc = reserve(p, std::min(n, max_resident_frames(p)), w, h);   // every allocation happens before the capture starts
    if (rc) return rc;
    dh_graph_destroy(p);
    HIP_TRY(hipStreamSynchronize(p->own_stream));
    HIP_TRY(hipStreamBeginCapture(p->own_stream, hipStreamCaptureModeThreadLocal));
    p->capturing = true;
    rc = dh_predict_batch_device(p, frames, n, w, h, K, midp_guess, rot_guess, guess_mask, out, p->own_stream);
    p->capturing = false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(p->own_stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail(DH_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    p->graph = g;
    HIP_TRY(hipGraphInstantiate(&p->graph_exec, p->graph, nullptr, nullptr, 0));
    return DH_OK;
}

static int graph_launch_(dh_predictor *p, void *stream) {
    if (!p) return fail(DH_EINVAL, "NULL predictor");
    if (p->graph_stale) return fail(DH_ESTATE, "the captured batch is stale: the workspace was reallocated after dh_graph_capture (larger batch, other frame size or debug taps); capture again");
    if (!p->graph_exec) return fail(DH_ESTATE, "no captured batch (dh_graph_capture)");
    HIP_TRY(hipGraphLaunch(p->graph_exec, (hipStream_t)stream));
    return DH_OK;
}

// ------------------------------------------------------------------ predict_mask / 2-D Hough votes
static int aux_reserve(dh_predictor *p, int n, int w, int h, size_t out_bytes) {
    int rc = reserve(p, n, w, h);
    if (rc) return rc;
    const Geom &g = p->geom;
    if (p->aux_cap < p->cap_frames || !p->aux_leaf) {
        HIP_TRY(hipDeviceSynchronize());
        for (void *q : {(void *)p->aux_leaf, (void *)p->aux_flags, (void *)p->aux_u32}) if (q) (void)hipFree(q);
        p->aux_leaf = nullptr; p->aux_flags = nullptr; p->aux_u32 = nullptr; p->aux_cap = 0;
        rc = dev_alloc(p, &p->aux_leaf, (size_t)p->cap_frames * std::max(g.npatch, 1) * p->n_trees);
        if (rc == DH_OK) rc = dev_alloc(p, &p->aux_flags, (size_t)p->cap_frames * std::max(g.npatch, 1));
        if (rc == DH_OK) rc = dev_alloc(p, &p->aux_u32, (size_t)p->cap_frames * w * h);
        if (rc) return rc;
        p->aux_cap = p->cap_frames;
    }
    if (out_bytes > p->aux_out_bytes) {
        HIP_TRY(hipDeviceSynchronize());
        if (p->aux_out) (void)hipFree(p->aux_out);
        p->aux_out = nullptr; p->aux_out_bytes = 0;
        uint8_t *q = nullptr;
        rc = dev_alloc(p, &q, out_bytes);
        if (rc) return rc;
        p->aux_out = q; p->aux_out_bytes = out_bytes;
    }
    return DH_OK;
}

// Taps of imageproc's gaussian_kernel_f32 (dh_blur_taps_, dh_host.cpp; parity unpinned), uploaded once per sigma.
static int blur_kernel(dh_predictor *p) {
    const float sigma = p->params.gaussian_sigma;
    if (p->blur_kern && p->blur_sigma == sigma) return DH_OK;
    std::vector<float> k;
    int rc0 = dh_blur_taps_(sigma, k);
    if (rc0) return rc0;
    HIP_TRY(hipDeviceSynchronize());
    if (p->blur_kern) (void)hipFree(p->blur_kern);
    p->blur_kern = nullptr;
    int rc = dev_alloc(p, &p->blur_kern, k.size());
    if (rc) return rc;
    HIP_TRY(hipMemcpy(p->blur_kern, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice));
    p->blur_klen = (int)k.size(); p->blur_sigma = sigma;
    return DH_OK;
}

static int aux_run(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], uint8_t *mask,
                   uint16_t *hough, hipStream_t s, bool blur = false, dh_pose *poses2d = nullptr) {
    const Geom &g = p->geom;
    float kinv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, kid[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (K) dh_mat3_inv_f32_(K, kinv);
    HIP_TRY(hipMemsetAsync(p->counters, 0, (size_t)p->cap_frames * sizeof(uint32_t), s));   // hit counters only
    HIP_TRY(hipMemsetAsync(p->counters + (size_t)p->cap_frames * (1 + DH_POSGRID + DH_GRID3), 0,
                           (size_t)p->cap_frames * ((size_t)p->geom.flag_words + (size_t)p->geom.tiles_x * p->geom.tiles_y) * sizeof(uint32_t), s));   // tile flags, window counts
    HIP_TRY(hipMemsetAsync(p->aux_flags, 0, (size_t)n * std::max(g.npatch, 1), s));
    int rc = enqueue_range(p, frames, 0, n, w, h, K ? K : kid, kinv, nullptr, nullptr, nullptr, p->ws_poses, s, false,
                           p->aux_leaf, p->aux_flags, true);
    if (rc) return rc;
    AuxArgs a{};
    a.frames = frames; a.n_frames = n; a.w = w; a.h = h;
    a.step = (int)p->params.stepwidth; a.sw = (int)p->params.subimage_width; a.sh = (int)p->params.subimage_height;
    a.lw = a.sw / 2; a.lh = a.sh / 2; a.nx = g.nx; a.ny = g.ny;
    memcpy(a.k, K ? K : kid, sizeof a.k); memcpy(a.kinv, kinv, sizeof a.kinv);
    a.f = p->dev; a.leaf = p->aux_leaf; a.flags = p->aux_flags;
    if (mask) {
        HIP_TRY(hipMemsetAsync(mask, 0, (size_t)n * w * h, s));          // ImageBuffer::new zero-fills (prediction.rs:852)
        a.mask = mask;
        HIP_TRY(dh_launch_mask(a, s));
    }
    if (hough) {
        HIP_TRY(hipMemsetAsync(p->aux_u32, 0, (size_t)n * w * h * sizeof(uint32_t), s));
        a.hough32 = p->aux_u32;
        HIP_TRY(dh_launch_hough2d(a, hough, s));
        // gaussian_blur_f32 (prediction.rs:844): horizontal pass into the (now free) 32-bit scratch, vertical pass back
        if (blur) HIP_TRY(dh_launch_blur_u16(hough, (uint16_t *)p->aux_u32, hough, n, w, h, p->blur_kern, p->blur_klen, s));
        if (poses2d) HIP_TRY(dh_launch_argmax2d(hough, frames, n, w, h, kinv, poses2d, s));
    }
    p->last_n = 0;   // the pose taps do not refer to this run
    return DH_OK;
}

static int predict_mask_device_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, uint8_t *mask, void *stream) {
    if (!p || !frames || !mask) return fail(DH_EINVAL, "dh_predict_mask_device: NULL argument");
    if (n <= 0) return n == 0 ? DH_OK : fail(DH_EINVAL, "negative batch size");
    DeviceGuard guard(p->device);
    if (!guard.ok) return fail(DH_EHIP, "cannot select device %d", p->device);
    const int slice = std::min(n, max_resident_frames(p));
    for (int f0 = 0; f0 < n; f0 += slice) {
        const int m = std::min(slice, n - f0);
        int rc = aux_reserve(p, m, w, h, 0);
        if (rc == DH_OK) rc = aux_run(p, frames + (size_t)f0 * w * h, m, w, h, nullptr, mask + (size_t)f0 * w * h, nullptr, (hipStream_t)stream);
        if (rc) return rc;
    }
    return DH_OK;
}

static int hough_image_device_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                                     uint16_t *out, void *stream) {
    if (!p || !frames || !K || !out) return fail(DH_EINVAL, "dh_hough_image_device: NULL argument");
    if (n <= 0) return n == 0 ? DH_OK : fail(DH_EINVAL, "negative batch size");
    DeviceGuard guard(p->device);
    if (!guard.ok) return fail(DH_EHIP, "cannot select device %d", p->device);
    const int slice = std::min(n, max_resident_frames(p));
    for (int f0 = 0; f0 < n; f0 += slice) {
        const int m = std::min(slice, n - f0);
        int rc = aux_reserve(p, m, w, h, 0);
        if (rc == DH_OK) rc = aux_run(p, frames + (size_t)f0 * w * h, m, w, h, K, nullptr, out + (size_t)f0 * w * h, (hipStream_t)stream);
        if (rc) return rc;
    }
    return DH_OK;
}

static int ensure_frame_staging(dh_predictor *p, size_t fbytes) {
    if (fbytes > p->ws_frames_bytes) {
        HIP_TRY(hipStreamSynchronize(p->own_stream));
        HIP_TRY(hipStreamSynchronize(p->copy_stream));
        if (p->ws_frames) (void)hipFree(p->ws_frames);
        p->ws_frames = nullptr; p->ws_frames_bytes = 0;
        size_t want = fbytes;
        int rc = dev_alloc(p, &p->ws_frames, want / sizeof(uint16_t));
        if (rc) return rc;
        p->ws_frames_bytes = want;
    }
    return DH_OK;
}
static int stage_frames(dh_predictor *p, const uint16_t *frames, int n, int w, int h) {
    size_t fbytes = (size_t)n * w * h * sizeof(uint16_t);
    int rc = ensure_frame_staging(p, fbytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(p->ws_frames, frames, fbytes, hipMemcpyHostToDevice, p->own_stream));
    return DH_OK;
}

static int predict_mask_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, uint8_t *mask) {
    if (!p || !frames || !mask) return fail(DH_EINVAL, "dh_predict_mask: NULL argument");
    if (n <= 0) return n == 0 ? DH_OK : fail(DH_EINVAL, "negative batch size");
    HIP_TRY(hipSetDevice(p->device));
    const int slice = std::min(n, max_resident_frames(p));
    for (int f0 = 0; f0 < n; f0 += slice) {
        const int m = std::min(slice, n - f0);
        const size_t ob = (size_t)m * w * h;
        int rc = aux_reserve(p, m, w, h, ob);
        if (rc == DH_OK) rc = stage_frames(p, frames + (size_t)f0 * w * h, m, w, h);
        if (rc == DH_OK) rc = aux_run(p, p->ws_frames, m, w, h, nullptr, (uint8_t *)p->aux_out, nullptr, p->own_stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(mask + (size_t)f0 * w * h, p->aux_out, ob, hipMemcpyDeviceToHost, p->own_stream));
        HIP_TRY(hipStreamSynchronize(p->own_stream));
    }
    return DH_OK;
}

static int hough_image_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], uint16_t *out) {
    if (!p || !frames || !K || !out) return fail(DH_EINVAL, "dh_hough_image: NULL argument");
    if (n <= 0) return n == 0 ? DH_OK : fail(DH_EINVAL, "negative batch size");
    HIP_TRY(hipSetDevice(p->device));
    const int slice = std::min(n, max_resident_frames(p));
    for (int f0 = 0; f0 < n; f0 += slice) {
        const int m = std::min(slice, n - f0);
        const size_t ob = (size_t)m * w * h * sizeof(uint16_t);
        int rc = aux_reserve(p, m, w, h, ob);
        if (rc == DH_OK) rc = stage_frames(p, frames + (size_t)f0 * w * h, m, w, h);
        if (rc == DH_OK) rc = aux_run(p, p->ws_frames, m, w, h, K, nullptr, (uint16_t *)p->aux_out, p->own_stream);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(out + (size_t)f0 * w * h, p->aux_out, ob, hipMemcpyDeviceToHost, p->own_stream));
        HIP_TRY(hipStreamSynchronize(p->own_stream));
    }
    return DH_OK;
}

// HoughPrediction::build_hough_image in full (prediction.rs:760-845) and predict_parameter_from2dhough (:343-367).
static int hough2d_device(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], uint16_t *img_out,
                          dh_pose *pose_out, hipStream_t s) {
    DeviceGuard guard(p->device);
    if (!guard.ok) return fail(DH_EHIP, "cannot select device %d", p->device);
    int rc = blur_kernel(p);
    if (rc) return rc;
    const int slice = std::min(n, max_resident_frames(p));
    for (int f0 = 0; f0 < n; f0 += slice) {
        const int m = std::min(slice, n - f0);
        const size_t ob = (size_t)m * w * h * sizeof(uint16_t);
        rc = aux_reserve(p, m, w, h, img_out ? 0 : ob);       // no caller image: the blurred image lives in the scratch output
        if (rc) return rc;
        uint16_t *img = img_out ? img_out + (size_t)f0 * w * h : (uint16_t *)p->aux_out;
        rc = aux_run(p, frames + (size_t)f0 * w * h, m, w, h, K, nullptr, img, s, true, pose_out ? pose_out + f0 : nullptr);
        if (rc) return rc;
    }
    return DH_OK;
}

static int build_hough_image_device_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                                           uint16_t *out, void *stream) {
    if (!p || !frames || !K || !out) return fail(DH_EINVAL, "dh_build_hough_image_device: NULL argument");
    if (n <= 0) return n == 0 ? DH_OK : fail(DH_EINVAL, "negative batch size");
    return hough2d_device(p, frames, n, w, h, K, out, nullptr, (hipStream_t)stream);
}

static int predict_from2dhough_device_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                                             dh_pose *out, void *stream) {
    if (!p || !frames || !K || !out) return fail(DH_EINVAL, "dh_predict_from2dhough_device: NULL argument");
    if (n <= 0) return n == 0 ? DH_OK : fail(DH_EINVAL, "negative batch size");
    return hough2d_device(p, frames, n, w, h, K, nullptr, out, (hipStream_t)stream);
}

static int build_hough_image_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], uint16_t *out) {
    if (!p || !frames || !K || !out) return fail(DH_EINVAL, "dh_build_hough_image: NULL argument");
    if (n <= 0) return n == 0 ? DH_OK : fail(DH_EINVAL, "negative batch size");
    HIP_TRY(hipSetDevice(p->device));
    int rc = blur_kernel(p);
    if (rc) return rc;
    const int slice = std::min(n, max_resident_frames(p));
    for (int f0 = 0; f0 < n; f0 += slice) {
        const int m = std::min(slice, n - f0);
        const size_t ob = (size_t)m * w * h * sizeof(uint16_t);
        rc = aux_reserve(p, m, w, h, ob);
        if (rc == DH_OK) rc = stage_frames(p, frames + (size_t)f0 * w * h, m, w, h);
        if (rc == DH_OK) rc = aux_run(p, p->ws_frames, m, w, h, K, nullptr, (uint16_t *)p->aux_out, p->own_stream, true, nullptr);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(out + (size_t)f0 * w * h, p->aux_out, ob, hipMemcpyDeviceToHost, p->own_stream));
        HIP_TRY(hipStreamSynchronize(p->own_stream));
    }
    return DH_OK;
}

static int predict_from2dhough_(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], dh_pose *out) {
    if (!p || !frames || !K || !out) return fail(DH_EINVAL, "dh_predict_from2dhough: NULL argument");
    if (n <= 0) return n == 0 ? DH_OK : fail(DH_EINVAL, "negative batch size");
    HIP_TRY(hipSetDevice(p->device));
    int rc = blur_kernel(p);
    if (rc) return rc;
    const int slice = std::min(n, max_resident_frames(p));
    for (int f0 = 0; f0 < n; f0 += slice) {
        const int m = std::min(slice, n - f0);
        rc = aux_reserve(p, m, w, h, (size_t)m * w * h * sizeof(uint16_t));
        if (rc == DH_OK) rc = stage_frames(p, frames + (size_t)f0 * w * h, m, w, h);
        SmallStage st;
        if (rc == DH_OK) rc = small_stage(p, m, &st);
        if (rc == DH_OK) rc = aux_run(p, p->ws_frames, m, w, h, K, nullptr, (uint16_t *)p->aux_out, p->own_stream, true, p->ws_poses);
        if (rc == DH_OK) rc = download_poses(p, st, m, out + f0, p->own_stream);
        if (rc) return rc;
    }
    return DH_OK;
}

// ------------------------------------------------------------------ profiling
static int set_profiling_(dh_predictor *p, int on) {
    if (!p) return fail(DH_EINVAL, "NULL predictor");
    p->profiling = on != 0;
    p->ev_valid = false;
    return DH_OK;
}
static int get_timing_(dh_predictor *p, dh_timing *out) {
    if (!p || !out) return fail(DH_EINVAL, "NULL argument");
    if (!p->ev_valid) return fail(DH_ESTATE, "no profiled batch yet (dh_set_profiling + a batch)");
    HIP_TRY(hipEventSynchronize(p->ev[3]));
    HIP_TRY(hipEventElapsedTime(&out->boxsum_ms, p->ev[0], p->ev[4]));
    HIP_TRY(hipEventElapsedTime(&out->traverse_ms, p->ev[4], p->ev[5]));
    HIP_TRY(hipEventElapsedTime(&out->emit_ms, p->ev[5], p->ev[1]));
    HIP_TRY(hipEventElapsedTime(&out->vote_ms, p->ev[1], p->ev[2]));
    HIP_TRY(hipEventElapsedTime(&out->cluster_ms, p->ev[2], p->ev[3]));
    HIP_TRY(hipEventElapsedTime(&out->total_ms, p->ev[0], p->ev[3]));
    out->n_frames = (uint32_t)p->last_n;

    return DH_OK;
}

// ------------------------------------------------------------------ parity taps
static int debug_enable_(dh_predictor *p, int on) {
    if (!p) return fail(DH_EINVAL, "NULL predictor");
    p->debug = on != 0;
    if (!p->debug) p->dbg_valid = false;
    return DH_OK;
}
static int tap_ready(dh_predictor *p) {
    if (!p) return fail(DH_EINVAL, "NULL predictor");
    if (p->last_n == 0) return fail(DH_ESTATE, "no batch has run on this predictor");
    hipError_t e = hipSetDevice(p->device);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(DH_EHIP, "sync: %s", hipGetErrorString(e));
    return DH_OK;
}
static int tap_ready_dbg(dh_predictor *p) {
    int rc = tap_ready(p);
    if (rc) return rc;
    if (!p->dbg_valid) return fail(DH_ESTATE, "debug taps were not enabled for the last batch");
    return DH_OK;
}
static int debug_leaf_indices_(dh_predictor *p, int32_t *out, size_t cap) {
    int rc = tap_ready_dbg(p);
    if (rc) return rc;
    size_t n = (size_t)p->last_n * p->geom.npatch * p->n_trees;
    if (!out || cap < n) return fail(DH_EINVAL, "buffer too small: need %zu elements", n);
    if (n) HIP_TRY(hipMemcpy(out, p->dbg_leaf, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return DH_OK;
}
static int debug_patch_flags_(dh_predictor *p, uint8_t *out, size_t cap) {
    int rc = tap_ready_dbg(p);
    if (rc) return rc;
    size_t n = (size_t)p->last_n * p->geom.npatch;
    if (!out || cap < n) return fail(DH_EINVAL, "buffer too small: need %zu elements", n);
    if (n) HIP_TRY(hipMemcpy(out, p->dbg_flags, n, hipMemcpyDeviceToHost));
    return DH_OK;
}
static int debug_grids_(dh_predictor *p, uint32_t *pos_grid, uint32_t *rot_grid) {
    int rc = tap_ready(p);
    if (rc) return rc;
    const uint32_t *pg = p->counters + p->cap_frames, *rg = pg + (size_t)p->cap_frames * DH_POSGRID;
    if (pos_grid) HIP_TRY(hipMemcpy(pos_grid, pg, (size_t)p->last_n * DH_POSGRID * 4, hipMemcpyDeviceToHost));
    if (rot_grid) HIP_TRY(hipMemcpy(rot_grid, rg, (size_t)p->last_n * DH_GRID3 * 4, hipMemcpyDeviceToHost));
    return DH_OK;
}
static int debug_hit_counts_(dh_predictor *p, uint32_t *out) {
    int rc = tap_ready(p);
    if (rc) return rc;
    if (!out) return fail(DH_EINVAL, "NULL output");
    HIP_TRY(hipMemcpy(out, p->counters, (size_t)p->last_n * 4, hipMemcpyDeviceToHost));
    return DH_OK;
}
static int debug_geometry_(dh_predictor *p, int32_t out[10]) {
    if (!p || !out) return fail(DH_EINVAL, "NULL argument");
    if (p->cap_frames == 0) return fail(DH_ESTATE, "no workspace yet (dh_predictor_reserve or a batch)");
    const Geom &g = p->geom;
    const bool walk_tab = g.uniform && p->absorb_ok;
    const int32_t v[10] = {(g.uniform ? 1 : 0) | (walk_tab ? 1 << 8 : 0) | (walk_tab ? g.top_levels << 16 : 0), g.px, g.py, g.tiles_x, g.tiles_y, g.swz_log2, g.swz_q, g.ss_row, p->f_rw, p->f_rh};
    memcpy(out, v, sizeof v);
    return DH_OK;
}
static int debug_guesses_(dh_predictor *p, int32_t *out) {
    int rc = tap_ready_dbg(p);
    if (rc) return rc;
    if (!out) return fail(DH_EINVAL, "NULL output");
    HIP_TRY(hipMemcpy(out, p->dbg_guess, (size_t)p->last_n * 6 * 4, hipMemcpyDeviceToHost));
    return DH_OK;
}
static int debug_meanshift_(dh_predictor *p, int which, int32_t *trace, uint32_t *steps) {
    int rc = tap_ready_dbg(p);
    if (rc) return rc;
    if (which < 0 || which > 1) return fail(DH_EINVAL, "which must be 0 or 1");
    size_t per = (size_t)(p->params.meanshift_iterations + 1) * 3;
    // device layout is [2][last batch n][...]: the kernel indexed with n_frames = last_n
    if (trace) HIP_TRY(hipMemcpy(trace, p->dbg_trace + (size_t)which * p->last_n * per, (size_t)p->last_n * per * 4, hipMemcpyDeviceToHost));
    if (steps) HIP_TRY(hipMemcpy(steps, p->dbg_steps + (size_t)which * p->last_n, (size_t)p->last_n * 4, hipMemcpyDeviceToHost));
    return DH_OK;
}
static int debug_votes_(dh_predictor *p, int frame, int which, int32_t *out, size_t cap, size_t *count) {
    int rc = tap_ready(p);
    if (rc) return rc;
    if (frame < 0 || frame >= p->last_n || which < 0 || which > 1 || !count) return fail(DH_EINVAL, "bad frame / which / count");
    // k_emit writes the rotation records only without the leaf histogram or with the taps on
    if (which == 1 && p->leaf_hits && !p->dbg_valid) return fail(DH_ESTATE, "rotation votes need dh_debug_enable(1) before the batch");
    if (cap > 0xffffffffull) cap = 0xffffffffull;
    if (!p->dbg_vcount) { rc = dev_alloc(p, &p->dbg_vcount, 1); if (rc) return rc; }
    if (cap > p->dbg_votes_cap) {
        if (p->dbg_votes) (void)hipFree(p->dbg_votes);
        p->dbg_votes = nullptr; p->dbg_votes_cap = 0;
        rc = dev_alloc(p, &p->dbg_votes, cap * 4);
        if (rc) return rc;
        p->dbg_votes_cap = cap;
    }
    HIP_TRY(hipMemset(p->dbg_vcount, 0, 4));
    VotesDumpArgs a{};
    a.frame = frame; a.which = which; a.f = p->dev; a.hits = p->hits; a.hit_box = p->hit_box; a.hit_rot = p->hit_rot; a.hit_count = p->counters; a.hits_cap = p->hits_cap;
    a.out = p->dbg_votes; a.cap = (uint32_t)cap; a.count = p->dbg_vcount;
    HIP_TRY(dh_launch_votes_dump(a, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    uint32_t c = 0;
    HIP_TRY(hipMemcpy(&c, p->dbg_vcount, 4, hipMemcpyDeviceToHost));
    *count = c;
    size_t ncopy = std::min<size_t>(c, cap);
    if (ncopy && out) HIP_TRY(hipMemcpy(out, p->dbg_votes, ncopy * 16, hipMemcpyDeviceToHost));
    return DH_OK;
}

// ------------------------------------------------------------------ the C ABI
// Every entry point of include/depthhead_hip.h runs its body (the *_ functions above) inside dh_guard_: the header promises
// that nothing throws or aborts across the boundary, and the bodies allocate (std::vector, std::string, std::thread).
#define DH_API(name, params, args) \
    extern "C" int dh_##name params { return dh_guard_("dh_" #name, [&]() -> int { return name##_ args; }); }
DH_API(forest_create, (const dh_forest_desc *d, dh_forest **out), (d, out))
DH_API(forest_destroy, (dh_forest *f), (f))
DH_API(forest_info, (const dh_forest *f, uint32_t *n_trees, uint32_t *n_nodes, uint32_t *n_leaves, uint32_t *max_depth), (f, n_trees, n_nodes, n_leaves, max_depth))
DH_API(patch_grid, (const dh_params *p, int w, int h, int *nx, int *ny), (p, w, h, nx, ny))
DH_API(predictor_destroy, (dh_predictor *p), (p))
DH_API(predictor_create, (const dh_forest *f, const dh_params *prm, int device, dh_predictor **out), (f, prm, device, out))
DH_API(predictor_update_sigma, (dh_predictor *p, float val), (p, val))
DH_API(predictor_sigma, (const dh_predictor *p, float *out), (p, out))
DH_API(predictor_reserve, (dh_predictor *p, int n, int w, int h), (p, n, w, h))
DH_API(predictor_set_forking, (dh_predictor *p, int chunks), (p, chunks))
DH_API(predict_batch_device, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask, dh_pose *out, void *stream_), (p, frames, n, w, h, K, midp_guess, rot_guess, guess_mask, out, stream_))
DH_API(predict_batch, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask, dh_pose *out), (p, frames, n, w, h, K, midp_guess, rot_guess, guess_mask, out))
DH_API(host_alloc, (size_t bytes, void **out), (bytes, out))
DH_API(host_free, (void *ptr), (ptr))
DH_API(biwi_decode_depth_device, (dh_predictor *p, const uint8_t *const *bufs, const size_t *lens, int n, uint16_t *frames_dev, size_t cap_px, uint32_t *w, uint32_t *h), (p, bufs, lens, n, frames_dev, cap_px, w, h))
DH_API(predict_batch_rle, (dh_predictor *p, const uint8_t *const *bufs, const size_t *lens, int n, const float K[9], const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask, dh_pose *out), (p, bufs, lens, n, K, midp_guess, rot_guess, guess_mask, out))
DH_API(graph_destroy, (dh_predictor *p), (p))
DH_API(graph_capture, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask, dh_pose *out), (p, frames, n, w, h, K, midp_guess, rot_guess, guess_mask, out))
DH_API(graph_launch, (dh_predictor *p, void *stream), (p, stream))
DH_API(predict_mask_device, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, uint8_t *mask, void *stream), (p, frames, n, w, h, mask, stream))
DH_API(hough_image_device, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], uint16_t *out, void *stream), (p, frames, n, w, h, K, out, stream))
DH_API(predict_mask, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, uint8_t *mask), (p, frames, n, w, h, mask))
DH_API(hough_image, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], uint16_t *out), (p, frames, n, w, h, K, out))
DH_API(build_hough_image_device, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], uint16_t *out, void *stream), (p, frames, n, w, h, K, out, stream))
DH_API(predict_from2dhough_device, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], dh_pose *out, void *stream), (p, frames, n, w, h, K, out, stream))
DH_API(build_hough_image, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], uint16_t *out), (p, frames, n, w, h, K, out))
DH_API(predict_from2dhough, (dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], dh_pose *out), (p, frames, n, w, h, K, out))
DH_API(set_profiling, (dh_predictor *p, int on), (p, on))
DH_API(get_timing, (dh_predictor *p, dh_timing *out), (p, out))
DH_API(debug_enable, (dh_predictor *p, int on), (p, on))
DH_API(debug_leaf_indices, (dh_predictor *p, int32_t *out, size_t cap), (p, out, cap))
DH_API(debug_patch_flags, (dh_predictor *p, uint8_t *out, size_t cap), (p, out, cap))
DH_API(debug_grids, (dh_predictor *p, uint32_t *pos_grid, uint32_t *rot_grid), (p, pos_grid, rot_grid))
DH_API(debug_hit_counts, (dh_predictor *p, uint32_t *out), (p, out))
DH_API(debug_geometry, (dh_predictor *p, int32_t out[10]), (p, out))
DH_API(debug_guesses, (dh_predictor *p, int32_t *out), (p, out))
DH_API(debug_meanshift, (dh_predictor *p, int which, int32_t *trace, uint32_t *steps), (p, which, trace, steps))
DH_API(debug_votes, (dh_predictor *p, int frame, int which, int32_t *out, size_t cap, size_t *count), (p, frame, which, out, cap, count))
#undef DH_API
