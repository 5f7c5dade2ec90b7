/*
 * dh_oracle.h -- CPU ORACLE for the depthhead Hough-forest head-pose path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path (depthhead_amd/ +
 * libdepthhead_hip.so) never links, imports or calls anything in oracle/.
 *
 * It is a plain-C restatement, in the reference's evaluation order and numeric types,
 * of /root/reference (Entscheider/depthhead):
 *   src/hough/prediction.rs:270-286, 310-317, 397-753
 *   src/hough/houghforest.rs:63-78, 185-193
 *   src/types.rs:33-61, 253-261, 314-340, 405-446
 *   src/meanshift.rs:14-68, 71-138, 228-252, 274-301, 322-408
 *   src/meancov_estimation.rs:76-148, 162-216, 260-282, 290-307, 335-378
 *
 * PINNING STATUS: the small-linear-algebra / geometry helpers are pinned against the
 * reference's own known-answer tests (src/types.rs:454-488, src/meancov_estimation.rs:450-533;
 * see tests/test_oracle_kat.py).  Everything downstream of the tree walk is
 * "PARITY UNPINNED": the reference holds no golden vector for leaf indices, vote
 * accumulators, mean shift or the final pose, its tree walk lives in the un-vendored
 * crate stamm 0.2.0 (Cargo.toml:17, Cargo.lock:1154-1162), and no Rust toolchain exists
 * in the build image, so the reference cannot be run.  The flat forest format below
 * names both children explicitly (child_zero / child_one) so the stamm Binar->child
 * convention is an importer concern, not an arithmetic one.
 *
 * Build with: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#ifndef DH_ORACLE_H
#define DH_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One split node: houghforest.rs:63-68 NodeParam{r1,r2,threshold}; Rect = topleft/bottomright
 * (types.rs:33-37) stored as x0,y0,x1,y1 relative to the patch.  child >= 0: node index,
 * child < 0: leaf index = ~child.  Binar::One -> child_one, Binar::Zero -> child_zero. */
typedef struct {
    uint16_t r1[4];
    uint16_t r2[4];
    double   threshold;
    int32_t  child_zero;
    int32_t  child_one;
} orc_node;

/* houghforest.rs:73-78 LeafParam{prob, offsets: Vec<Vec3<f32>>, rotations: Vec<Vec3<f64>>},
 * flattened CSR-style. */
typedef struct {
    uint32_t        n_trees;
    const int32_t  *roots;       /* per tree: node index, or ~leaf when the tree is a single leaf */
    uint32_t        n_nodes;
    const orc_node *nodes;
    uint32_t        n_leaves;
    const double   *leaf_prob;
    const uint32_t *off_begin;   /* n_leaves+1 */
    const uint32_t *rot_begin;   /* n_leaves+1 */
    const float    *offsets;     /* 3 per vote */
    const double   *rotations;   /* 3 per vote, degrees */
} orc_forest;

/* prediction.rs:239-256 HoughPrediction's serialised scalars. */
typedef struct {
    uint32_t stepwidth;
    uint32_t subimage_width;
    uint32_t subimage_height;
    float    gaussian_sigma;
    uint32_t meanshift_iterations;
} orc_model;

enum { ORC_RECT_FAITHFUL = 0,  /* O(area) pixel loops, as types.rs:317-339 */
       ORC_RECT_SAT      = 1 };/* summed-area table, identical sums, O(1)   */

/* Optional taps; every pointer may be NULL. */
typedef struct {
    int32_t  *leaf_idx;       /* [n_patches * n_trees], -1 for background patches          */
    uint8_t  *patch_flags;    /* [n_patches] bit0 = non-background, bit1 = prob gate passed */
    uint32_t *pos_grid;       /* [400]  */
    uint32_t *rot_grid;       /* [8000] */
    int32_t  *guess_mid;      /* [3] cell coordinates after optional override               */
    double   *guess_rot_deg;  /* [3] degrees (prediction.rs:745-747 / :448-450)             */
    int32_t  *guess_rot;      /* [3] grid coordinates (prediction.rs:458-460)               */
    int32_t  *mid_cells;      /* [mid_cap*4] (x,y,z,value) sorted lexicographically         */
    uint32_t  mid_cap;
    uint32_t *mid_count;      /* number of distinct cells (may exceed cap; then truncated)  */
    int32_t  *rot_cells;
    uint32_t  rot_cap;
    uint32_t *rot_count;
    int32_t  *ms_trace_mid;   /* [(iterations+1)*3] positions, entry 0 = init               */
    uint32_t *ms_steps_mid;   /* number of position updates performed                       */
    int32_t  *ms_trace_rot;
    uint32_t *ms_steps_rot;
} orc_taps;

typedef struct {
    float  mid_point[3];
    double rotation[3];
} orc_pose;

/* Number of sliding-window positions (prediction.rs:535-548, 684-686). */
int orc_patch_grid(uint32_t w, uint32_t h, const orc_model *m, uint32_t *nx, uint32_t *ny);

/* predict_parameter / predict_parameter_parallel (prediction.rs:376-409; identical results by
 * construction).  K is row-major 3x3.  midp_guess / rot_guess may be NULL (= None).
 * Returns 0, or negative on invalid arguments. */
int orc_predict(const orc_forest *f, const orc_model *m, const uint16_t *img, uint32_t w, uint32_t h,
                const float K[9], const float *midp_guess, const double *rot_guess, int rect_mode,
                orc_pose *out, const orc_taps *taps);

/* Frame-parallel batch (OpenMP), used as the timed CPU baseline. guesses: n*3 or NULL. */
int orc_predict_batch(const orc_forest *f, const orc_model *m, const uint16_t *imgs, uint32_t n,
                      uint32_t w, uint32_t h, const float K[9], const float *midp_guess,
                      const double *rot_guess, int rect_mode, int threads, orc_pose *out);

/* ---- sibling consumers of the tree walk ---- */
/* HoughPrediction::predict_mask (prediction.rs:850-905): per-patch mean leaf probability as u8,
 * painted into a stepwidth x stepwidth block.  mask: w*h bytes, fully written. */
int orc_predict_mask(const orc_forest *f, const orc_model *m, const uint16_t *img, uint32_t w, uint32_t h,
                     int rect_mode, uint8_t *mask);
/* The voting stage of HoughPrediction::build_hough_image (prediction.rs:760-840), i.e. the u16 image
 * BEFORE imageproc's gaussian_blur_f32 (:844; see orc_build_hough_image).  out: w*h u16. */
int orc_hough_image(const orc_forest *f, const orc_model *m, const uint16_t *img, uint32_t w, uint32_t h,
                    const float K[9], int rect_mode, uint16_t *out);

/* imageproc 0.12.0 filter::gaussian_blur_f32 on a u16 image (external crate, source NOT in the container: restated from
 * its published algorithm, PARITY UNPINNED -- details in dh_oracle.c).  out: w*h u16. */
int orc_gaussian_kernel_f32(float sigma, float *out, uint32_t cap, uint32_t *len);
int orc_gaussian_blur_u16(const uint16_t *in, uint32_t w, uint32_t h, float sigma, uint16_t *out);
/* HoughPrediction::build_hough_image in full (prediction.rs:760-845) and predict_parameter_from2dhough (:343-367). */
int orc_build_hough_image(const orc_forest *f, const orc_model *m, const uint16_t *img, uint32_t w, uint32_t h,
                          const float K[9], int rect_mode, uint16_t *out);
int orc_predict_from2dhough(const orc_forest *f, const orc_model *m, const uint16_t *img, uint32_t w, uint32_t h,
                            const float K[9], int rect_mode, orc_pose *out);

/* ---- helpers exported for the known-answer tests ---- */
void   orc_mat3_inv_f64(const double m[9], double out[9]);        /* meancov_estimation.rs:344-352 */
void   orc_mat3_inv_f32(const float m[9], float out[9]);
double orc_mat3_det_f64(const double m[9]);                       /* :339-343 */
void   orc_mat3_vec_f64(const double m[9], const double v[3], double out[3]); /* :201-216 */
void   orc_outer_f64(const double v[3], double out[9]);           /* :267-282 */
int    orc_mean_cov_f64(const double *set, uint32_t n, double mean[3], double cov[9]); /* :359-378 */
int    orc_mean_cov_f32(const float *set, uint32_t n, float mean[3], float cov[9]);
double orc_trace_f64(const double m[9]);                          /* :260-265 */
void   orc_space_to_img(const float K[9], const float p[3], float out[2]);          /* types.rs:424-428 */
void   orc_img_to_space(const float K[9], const float px[2], float z, float out[3]);/* types.rs:432-445 */
double orc_average_value_in_rect(const uint16_t *img, uint32_t w, uint32_t ox, uint32_t oy,
                                 const uint16_t r[4]);            /* types.rs:317-339 */
void   orc_build_kernel(uint32_t size, float variance, float *out);                 /* meanshift.rs:228-252 */

#ifdef __cplusplus
}
#endif
#endif
