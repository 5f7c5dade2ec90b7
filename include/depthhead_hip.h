/*
 * depthhead_hip.h -- C ABI of libdepthhead_hip.so: the MI355X (gfx950) implementation of
 * depthhead's Hough-forest head-pose inference path.
 *
 * The reference (Entscheider/depthhead, pure Rust) has no FFI or plugin interface; its public
 * seam for this path is
 *     HoughPrediction::predict_parameter_parallel(&self, img: Arc<DepthImage>,
 *         intrinsic: &IntrinsicMatrix, midp_guess: Option<[f32;3]>, rot_guess: Option<[f64;3]>)
 *         -> PredictionResult                                   (src/hough/prediction.rs:397-409)
 * and its serial twin predict_parameter (:376-388).  This library sits where
 * predict_parameter_generic (:421-493) sits, with a FRAME BATCH as the unit of work, so a Rust
 * `extern "C"` shim can keep the method signature unchanged.
 *
 * Plain C: pointers and sizes only.  Every entry point returns an int status (0 = DH_OK,
 * negative = error) and never throws or aborts across the boundary; dh_last_error() gives the
 * message of the calling thread's last failure.
 *
 * Threading (mirrors `HoughPrediction: !Sync`, prediction.rs:253 / types.rs:405): a dh_forest is
 * immutable and may be shared; a dh_predictor is NOT thread-safe -- one per host thread / GPU
 * stream.  Distinct predictors may run concurrently.
 */
#ifndef DEPTHHEAD_HIP_H
#define DEPTHHEAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DH_VERSION 100 /* 0.1.0 */

/* ---- status codes ---- */
#define DH_OK 0
#define DH_EINVAL (-1)  /* NULL / out-of-range argument                                    */
#define DH_EFOREST (-2) /* forest violates an invariant the reference would panic on        */
#define DH_EHIP (-3)    /* HIP runtime error (message has the hipError string)              */
#define DH_ENOMEM (-4)  /* host or device allocation failed                                 */
#define DH_ESIZE (-5)   /* geometry unsupported: frame smaller than the patch, etc.         */
#define DH_ESTATE (-6)  /* call not valid in this state (e.g. debug tap without a batch)    */

/* ---- compile-time constants of the reference (src/hough/prediction.rs:270-286, :314, :584) ---- */
#define DH_ZSCALEFACTOR 1
#define DH_GUESS_GRID_PARTS 20
#define DH_ROT_GRID_PARTS 120
#define DH_MAX_VARIANCE_ROT 400.0
#define DH_MAX_VARIANCE_OFFSET 5200.0f
#define DH_MEANSHIFT_KERNEL_SIZE 20
#define DH_PROB_GATE 0.7

/* One split node.  Replaces houghforest.rs:63-68 `NodeParam{r1: Rect, r2: Rect, threshold: f64}`
 * plus the child links stamm keeps.  r = {x0, y0, x1, y1}: Rect.topleft / Rect.bottomright
 * (src/types.rs:33-37), relative to the patch.  child >= 0 is a node index, child < 0 is the leaf
 * ~child.  child_one is taken when avg(r1) - avg(r2) > threshold (Binar::One,
 * houghforest.rs:185-193), child_zero otherwise. */
typedef struct dh_node {
    uint16_t r1[4];
    uint16_t r2[4];
    double   threshold;
    int32_t  child_zero;
    int32_t  child_one;
} dh_node; /* 32 bytes */

/* Host-side description of a forest; dh_forest_create copies everything.  Replaces the serde
 * model `RandomForest<LeafParam, HoughTreeFunctions>` (prediction.rs:33-34) with
 * `LeafParam{prob: f64, offsets: Vec<Vec3<f32>>, rotations: Vec<Vec3<f64>>}` (houghforest.rs:73-78)
 * in CSR form. */
typedef struct dh_forest_desc {
    uint32_t        n_trees;
    const int32_t  *roots;      /* [n_trees] node index, or ~leaf for a single-leaf tree */
    uint32_t        n_nodes;
    const dh_node  *nodes;      /* [n_nodes] */
    uint32_t        n_leaves;
    const double   *leaf_prob;  /* [n_leaves] */
    const uint32_t *off_begin;  /* [n_leaves + 1] into offsets   */
    const uint32_t *rot_begin;  /* [n_leaves + 1] into rotations */
    const float    *offsets;    /* [off_begin[n_leaves] * 3] mm      */
    const double   *rotations;  /* [rot_begin[n_leaves] * 3] degrees */
} dh_forest_desc;

/* The serialised scalars of `HoughPrediction` (prediction.rs:239-256). */
typedef struct dh_params {
    uint32_t stepwidth;
    uint32_t subimage_width;
    uint32_t subimage_height;
    float    gaussian_sigma;        /* used as the VARIANCE of the kernel, prediction.rs:314 */
    uint32_t meanshift_iterations;
} dh_params;

/* `PredictionResult` (prediction.rs:259-267).  bounding_box is always Rect(0,0,0,0) there
 * (:491) and is not carried.  36 payload bytes; `reserved` fills the natural padding and is 0. */
typedef struct dh_pose {
    float    mid_point[3]; /* mm, camera space, integer-valued          */
    uint32_t reserved;
    double   rotation[3];  /* radians, multiples of 3.14159/60          */
} dh_pose; /* 40 bytes */

typedef struct dh_forest dh_forest;       /* opaque, immutable */
typedef struct dh_predictor dh_predictor; /* opaque, one per thread/stream */

/* Average duration of each kernel over the last batch, when profiling is on. */
typedef struct dh_timing {
    float traverse_ms; /* tile build + background gate + tree walks (excludes boxsum_ms, emit_ms) */
    float vote_ms;     /* coarse 20x20 / 20^3 guess grids                    */
    float cluster_ms;  /* initial guesses + both mean shifts                 */
    float total_ms;    /* first kernel start -> last kernel end              */
    uint32_t n_frames;
    float boxsum_ms;   /* rectangle-sum images or pixel flags, and the list of flagged tiles */
    float emit_ms;     /* probability gate + hit records                     */
    uint32_t reserved;
} dh_timing;

const char *dh_last_error(void);
int dh_version(void);

/* Validate and copy a forest.  Rejected with DH_EFOREST where the reference would panic or read
 * out of bounds: child/root index out of range, a node reachable twice (cycle / DAG), a rectangle
 * with x1 < x0 or y1 < y0 (u32 underflow, types.rs:47-52), non-monotone CSR arrays, a leaf with
 * prob > 0 and no offset (division by zero, prediction.rs:594) or no rotation (unwrap of None,
 * :600), a rotation whose bin leaves [0,120) after the single wrap (index out of bounds, :636). */
int dh_forest_create(const dh_forest_desc *desc, dh_forest **out);
int dh_forest_destroy(dh_forest *f);
int dh_forest_info(const dh_forest *f, uint32_t *n_trees, uint32_t *n_nodes, uint32_t *n_leaves,
                   uint32_t *max_depth);

/* Upload the forest to `device`, precompute the per-leaf vote tables on the GPU and build the
 * mean-shift kernel table (get_or_build_kernel, prediction.rs:310-317).  Rejects (DH_EFOREST) a
 * split rectangle that leaves the patch and (DH_ESIZE) a patch whose pixel sum can exceed 2^32.
 * One window's summed-area table must also fit the 160 KB LDS of a CU: the first batch / reserve call
 * refuses (DH_ESIZE) patches beyond about 195 x 195 (the reference's only trainer uses 80 x 80). */
int dh_predictor_create(const dh_forest *f, const dh_params *p, int device, dh_predictor **out);
int dh_predictor_destroy(dh_predictor *p);
/* HoughPrediction::update_sigma / sigma (prediction.rs:320-331): no-op for val <= 0 or unchanged. */
int dh_predictor_update_sigma(dh_predictor *p, float val);
int dh_predictor_sigma(const dh_predictor *p, float *out);

/* predict_parameter_parallel over n frames held in HOST memory (row-major u16, index y*w+x,
 * types.rs:10).  K: row-major 3x3 intrinsic (types.rs:405).  midp_guess (n*3 f32) / rot_guess
 * (n*3 f64, radians) are the Option<> arguments: NULL = None for every frame; guess_mask (n bytes,
 * may be NULL = all present) selects per frame: bit0 = midp_guess is Some, bit1 = rot_guess is
 * Some.  Synchronous: copies in, runs, copies out.  The upload is pipelined: frames cross PCIe in chunks while the
 * kernels of the previous chunk run (without dh_debug_enable the parity taps then describe the LAST chunk only). */
int dh_predict_batch(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                     const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask,
                     dh_pose *out);

/* Page-locked host memory for frame buffers (a camera driver or file reader writes frames there): dh_predict_batch
 * then uploads by asynchronous DMA at PCIe speed, chunk k + 1 while the kernels of chunk k run.  Pageable buffers work
 * too, more slowly (the runtime pins and unpins the pages of every copy). */
int dh_host_alloc(size_t bytes, void **out);
int dh_host_free(void *ptr);

/* The same batch handed over as BIWI run-length coded depth payloads -- the bytes of the `.bin` files that
 * db_reader::biwi::read_depth (src/db_reader/biwi.rs:81-103) parses: bufs[i] / lens[i] = payload of frame i; every
 * frame must decode to the same w x h.  The host only walks the run headers (with the checks of dh_biwi_decode_depth:
 * a truncated payload or a run that overruns the image gives DH_EINVAL BEFORE anything is launched); payloads and
 * the run table cross PCIe instead of the 2-byte pixels (a BIWI frame is ~80 % background), the depth images are
 * rebuilt on the device, byte for byte what dh_biwi_decode_depth yields, and predicted.  Synchronous. */
int dh_predict_batch_rle(dh_predictor *p, const uint8_t *const *bufs, const size_t *lens, int n, const float K[9],
                         const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask, dh_pose *out);
/* The decode step alone: n payloads -> DEVICE frames [n][h][w] u16 (frames_dev == NULL: validate and return
 * *w, *h only).  Synchronous (returns when the frames are in place). */
int dh_biwi_decode_depth_device(dh_predictor *p, const uint8_t *const *bufs, const size_t *lens, int n,
                                uint16_t *frames_dev, size_t cap_px, uint32_t *w, uint32_t *h);

/* Same with DEVICE pointers (frames, guesses, mask, out all device-resident) on `stream`
 * (a hipStream_t; NULL = default stream).  Asynchronous: returns after enqueueing; buffers must
 * stay valid until the stream reaches this point.  Performs no allocation and no host
 * synchronisation once the workspace for (n, w, h) exists, so it can be captured in a hipGraph
 * (call dh_predictor_reserve first). */
int dh_predict_batch_device(dh_predictor *p, const uint16_t *frames, int n, int w, int h,
                            const float K[9], const float *midp_guess, const double *rot_guess,
                            const uint8_t *guess_mask, dh_pose *out, void *stream);

/* ---- sibling consumers of the tree walk (same kernel, different epilogue) ----
 * HoughPrediction::predict_mask (prediction.rs:850-905): per window the mean leaf probability,
 * as u8 = (prob * 255) painted into a stepwidth x stepwidth block; mask is n*h*w bytes. */
int dh_predict_mask(dh_predictor *p, const uint16_t *frames, int n, int w, int h, uint8_t *mask);
int dh_predict_mask_device(dh_predictor *p, const uint16_t *frames, int n, int w, int h, uint8_t *mask,
                           void *stream);
/* The VOTING stage of HoughPrediction::build_hough_image (prediction.rs:760-840): every leaf with
 * prob >= 0.95 casts (255 * prob) / n_offsets at the projected vote pixel, u16 wrapping.  out is
 * n*h*w u16, the image the reference then passes to imageproc::filter::gaussian_blur_f32 (:844):
 * dh_build_hough_image below adds that blur, dh_predict_from2dhough the argmax of :343-367. */
int dh_hough_image(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                   uint16_t *out);
int dh_hough_image_device(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                          uint16_t *out, void *stream);

/* HoughPrediction::build_hough_image IN FULL (prediction.rs:760-845): the votes above passed through
 * imageproc::filter::gaussian_blur_f32(_, gaussian_sigma) (:844) -- a separable Gaussian with taps at 0..ceil(2 sigma),
 * f32 accumulation in tap order, each pass clamped and truncated to u16, image borders replicated.  imageproc 0.12.0
 * (Cargo.lock:555) is an external crate whose source is not vendored: PARITY UNPINNED, the blur restates the crate's
 * published algorithm (see oracle/dh_oracle.c orc_gaussian_blur_u16).  Needs gaussian_sigma > 0 (the crate asserts). */
int dh_build_hough_image(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], uint16_t *out);
int dh_build_hough_image_device(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                                uint16_t *out, void *stream);
/* HoughPrediction::predict_parameter_from2dhough (prediction.rs:343-367): argmax of that image -- `max_by_key`
 * keeps the LAST of equal maxima -- lifted to 3-D with the frame's depth at that pixel (img_to_space_coord);
 * rotation is always (0, 0, 0) there (:363). */
int dh_predict_from2dhough(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9], dh_pose *out);
int dh_predict_from2dhough_device(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                                  dh_pose *out, void *stream);

/* hipGraph path for launch-bound use (single frames, small frames): capture ONE
 * dh_predict_batch_device call -- device pointers, sizes and K are baked in -- then replay it with one
 * host call per batch; new inputs are written into the same buffers between replays. */
int dh_graph_capture(dh_predictor *p, const uint16_t *frames, int n, int w, int h, const float K[9],
                     const float *midp_guess, const double *rot_guess, const uint8_t *guess_mask, dh_pose *out);
int dh_graph_launch(dh_predictor *p, void *stream);
int dh_graph_destroy(dh_predictor *p);

/* Allocate the workspace for batches of up to n frames of w x h. */
int dh_predictor_reserve(dh_predictor *p, int n, int w, int h);

/* Forked sub-batches inside one device call.  A call of >= 512 frames on an otherwise idle GPU runs faster as two halves on two
 * streams (the latency-bound tail kernels of one half beside the head kernels of the other); a caller that keeps several
 * predictors in flight already has that overlap and loses to the fork's event plumbing (MI355X, 512 frames per call, four
 * predictors: 669 k frames/s forked, 713 k whole).  chunks: 0 = automatic (two halves from 512 frames on; the default, or
 * DH_CHUNKS), 1 = never fork, 2 .. 8 = that many.  No counterpart in the reference (rayon decides there). */
int dh_predictor_set_forking(dh_predictor *p, int chunks);

/* Number of sliding-window positions for a frame size (prediction.rs:535-548, 684-686). */
int dh_patch_grid(const dh_params *p, int w, int h, int *nx, int *ny);

/* ---- BIWI Kinect Head Pose Database formats (frame ingest, src/db_reader/biwi.rs) ----
 * read_depth (biwi.rs:81-103): run-length coded depth `.bin` -> row-major u16.  Call with out == NULL
 * to obtain *w, *h.  Where the reference returns an io::Error (truncated file) or panics (a run
 * overruns the image) this returns DH_EINVAL. */
int dh_biwi_decode_depth(const uint8_t *buf, size_t len, uint16_t *out, size_t cap_px, uint32_t *w, uint32_t *h);
/* read_cal (biwi.rs:27-60): `depth.cal` text -> row-major 3x3 intrinsic (first three lines). */
int dh_biwi_parse_cal(const char *text, size_t len, float K[9]);
/* read_gt (biwi.rs:63-77): 24-byte pose file -> head position (mm), its projection, rotation (deg). */
int dh_biwi_parse_pose(const uint8_t *buf, size_t len, const float K[9], float pos3d[3], float pos2d[2], float rot[3]);

/* ---- profiling ---- */
int dh_set_profiling(dh_predictor *p, int on); /* HIP events around each kernel, on the launch stream; and roctx ranges
                                                * ("dh:batch ...", "dh:boxsum", "dh:traverse", "dh:emit", "dh:vote", "dh:cluster") around the
                                                * launches, visible to rocprofv3 --marker-trace when a roctx library is present */
int dh_get_timing(dh_predictor *p, dh_timing *out); /* synchronises the recorded events */

/* ---- parity taps (tests only; each refers to the LAST batch run on this predictor) ----
 * dh_debug_enable(1) makes the next batch record leaf indices, patch flags and mean-shift traces. */
int dh_debug_enable(dh_predictor *p, int on);
/* [n][n_patches][n_trees] leaf index per (patch, tree), -1 for background patches. */
int dh_debug_leaf_indices(dh_predictor *p, int32_t *out, size_t cap_elems);
/* [n][n_patches]: bit0 = non-background (prediction.rs:567-571), bit1 = prob gate passed (:584). */
int dh_debug_patch_flags(dh_predictor *p, uint8_t *out, size_t cap_elems);
/* Coarse guess grids (prediction.rs:529-533): pos_grid [n][400], rot_grid [n][8000]. */
int dh_debug_grids(dh_predictor *p, uint32_t *pos_grid, uint32_t *rot_grid);
/* [n][6]: mean-shift start cells, mid x,y,z then rot x,y,z (prediction.rs:437-460, :750). */
int dh_debug_guesses(dh_predictor *p, int32_t *out);
/* Every vote of one frame as (x, y, z, value) records, unaggregated and in no particular order:
 * which = 0 the position accumulator `mid` (prediction.rs:667), 1 the rotation accumulator `rot`
 * (:635).  *count receives the number of votes (may exceed cap; then only cap are written). */
int dh_debug_votes(dh_predictor *p, int frame, int which, int32_t *out, size_t cap_records, size_t *count);
/* Mean-shift positions: trace [n][iterations+1][3] (entry 0 = start), steps [n] = updates done
 * (meanshift.rs:336-395). which as above. */
int dh_debug_meanshift(dh_predictor *p, int which, int32_t *trace, uint32_t *steps);
/* Per-frame number of (patch, leaf) hit records kept for voting. */
int dh_debug_hit_counts(dh_predictor *p, uint32_t *out);
/* Tiling the runtime chose for the current workspace (tests construct edge cases from it):
 * out[0..9] = path (bit 0: uniform-rectangle path; bit 8: its walks use the walk table (always, unless the forest has more than 4096
 * nodes with an ambiguity band); bits 16..: tree levels walked from LDS), tile px, py, tiles_x, tiles_y, column de-interleave log2,
 * plane stride q, LDS row stride, rectangle w, rectangle h. */
int dh_debug_geometry(dh_predictor *p, int32_t out[10]);

#ifdef __cplusplus
}
#endif
#endif /* DEPTHHEAD_HIP_H */
