/*
 * dh_oracle.c -- CPU ORACLE (test infrastructure only; see dh_oracle.h for scope,
 * reference citations and the "parity unpinned" statement).
 *
 * Numeric contract: IEEE binary32/binary64, separate multiply and add (compile with
 * -ffp-contract=off), Rust `as` cast semantics (float->int truncates toward zero,
 * saturates, NaN -> 0), wrapping u32/i32 arithmetic as in a Rust release build.
 */
#include "dh_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ Rust `as` casts */
static inline int32_t f32_as_i32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
static inline int32_t f64_as_i32(double v) {
    if (v != v) return 0;
    if (v >= 2147483648.0) return INT32_MAX;
    if (v <= -2147483648.0) return INT32_MIN;
    return (int32_t)v;
}
static inline uint64_t f64_as_usize(double v) {
    if (v != v || v <= 0.0) return 0;
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)v;
}
static inline uint64_t f32_as_usize(float v) {
    if (v != v || v <= 0.0f) return 0;
    if (v >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)v;
}
static inline int32_t wrap_add_i32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

/* ------------------------------------------------------------------ Mat3 / Vec3
 * Generated for float and double so the text that is KAT-pinned in f64
 * (meancov_estimation.rs:450-533) is the text that runs in f32 on the hot path. */
#define DEFINE_LINALG(T, S)                                                                        \
    /* meancov_estimation.rs:339-343 */                                                            \
    static T det3_##S(const T m[9]) {                                                              \
        return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[3] * (m[1] * m[8] - m[2] * m[7]) +           \
               m[6] * (m[1] * m[5] - m[2] * m[4]);                                                 \
    }                                                                                              \
    /* :344-352 adjugate / det ; Div<T> for Mat (:162-173) divides element-wise */                \
    static void inv3_##S(const T m[9], T o[9]) {                                                   \
        T a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7],          \
          i = m[8];                                                                                \
        T dt = det3_##S(m);                                                                        \
        o[0] = (e * i - f * h) / dt; o[1] = (c * h - b * i) / dt; o[2] = (b * f - c * e) / dt;     \
        o[3] = (f * g - d * i) / dt; o[4] = (a * i - c * g) / dt; o[5] = (c * d - a * f) / dt;     \
        o[6] = (d * h - e * g) / dt; o[7] = (b * g - a * h) / dt; o[8] = (a * e - b * d) / dt;     \
    }                                                                                              \
    /* :201-216 tmp = rhs[0]*m[j][0]; tmp = tmp + rhs[i]*m[j][i] */                                \
    static void matvec3_##S(const T m[9], const T v[3], T o[3]) {                                  \
        for (int j = 0; j < 3; ++j) {                                                              \
            T tmp = v[0] * m[j * 3 + 0];                                                           \
            for (int i = 1; i < 3; ++i) tmp = tmp + v[i] * m[j * 3 + i];                           \
            o[j] = tmp;                                                                            \
        }                                                                                          \
    }                                                                                              \
    /* :267-282 */                                                                                 \
    static void outer3_##S(const T v[3], T o[9]) {                                                 \
        for (int i = 0; i < 3; ++i)                                                                \
            for (int j = 0; j < 3; ++j) o[i * 3 + j] = v[i] * v[j];                                \
    }                                                                                              \
    /* :260-265 (0..n).map(|i| m[i][i]).sum() */                                                   \
    static T trace3_##S(const T m[9]) {                                                            \
        T s = (T)0;                                                                                \
        s = s + m[0]; s = s + m[4]; s = s + m[8];                                                  \
        return s;                                                                                  \
    }                                                                                              \
    /* :359-378 ; the `/ n as f64` is Div<f64>: for f32 it divides by (n as f64) as f32 (:290-297) */ \
    static int mean_cov3_##S(const T *set, uint32_t n, T mean[3], T cov[9]) {                      \
        if (n == 0) return 0;                                                                      \
        T mu[3] = {set[0], set[1], set[2]};                                                        \
        for (uint32_t i = 1; i < n; ++i)                                                           \
            for (int k = 0; k < 3; ++k) mu[k] = mu[k] + set[i * 3 + k];                            \
        T dn = (T)(double)n;                                                                       \
        for (int k = 0; k < 3; ++k) mu[k] = mu[k] / dn;                                            \
        T d[3], t[9], c[9];                                                                        \
        for (int k = 0; k < 3; ++k) d[k] = set[k] - mu[k];                                         \
        outer3_##S(d, c);                                                                          \
        for (uint32_t i = 1; i < n; ++i) {                                                         \
            for (int k = 0; k < 3; ++k) d[k] = set[i * 3 + k] - mu[k];                             \
            outer3_##S(d, t);                                                                      \
            for (int k = 0; k < 9; ++k) c[k] = c[k] + t[k];                                        \
        }                                                                                          \
        T dn1 = (T)(double)(n - 1);                                                                \
        for (int k = 0; k < 9; ++k) c[k] = c[k] / dn1;                                             \
        for (int k = 0; k < 3; ++k) mean[k] = mu[k];                                               \
        for (int k = 0; k < 9; ++k) cov[k] = c[k];                                                 \
        return 1;                                                                                  \
    }

DEFINE_LINALG(float, f32)
DEFINE_LINALG(double, f64)

void orc_mat3_inv_f64(const double m[9], double o[9]) { inv3_f64(m, o); }
void orc_mat3_inv_f32(const float m[9], float o[9]) { inv3_f32(m, o); }
double orc_mat3_det_f64(const double m[9]) { return det3_f64(m); }
void orc_mat3_vec_f64(const double m[9], const double v[3], double o[3]) { matvec3_f64(m, v, o); }
void orc_outer_f64(const double v[3], double o[9]) { outer3_f64(v, o); }
int orc_mean_cov_f64(const double *s, uint32_t n, double mean[3], double cov[9]) { return mean_cov3_f64(s, n, mean, cov); }
int orc_mean_cov_f32(const float *s, uint32_t n, float mean[3], float cov[9]) { return mean_cov3_f32(s, n, mean, cov); }
double orc_trace_f64(const double m[9]) { return trace3_f64(m); }

/* ------------------------------------------------------------------ intrinsics (types.rs:405-446) */
static void to2d(const float K[9], const float p[3], float out[2]) {
    float r[3];
    matvec3_f32(K, p, r);       /* types.rs:425 */
    float c = r[2];
    out[0] = r[0] / c;          /* :427 */
    out[1] = r[1] / c;
}
static void to3d(const float Kinv[9], float px, float py, float z, float out[3]) {
    float v3[3] = {px, py, 1.0f}; /* types.rs:435 */
    float r[3];
    matvec3_f32(Kinv, v3, r);     /* :442 */
    float c = z / r[2];           /* :443 */
    out[0] = r[0] * c;            /* :444 Vec * scalar, element-wise (:138-148) */
    out[1] = r[1] * c;
    out[2] = r[2] * c;
}
void orc_space_to_img(const float K[9], const float p[3], float out[2]) { to2d(K, p, out); }
void orc_img_to_space(const float K[9], const float px[2], float z, float out[3]) {
    float Kinv[9];
    inv3_f32(K, Kinv);            /* types.rs:436-441 (cached there) */
    to3d(Kinv, px[0], px[1], z, out);
}

/* ------------------------------------------------------------------ rectangle means (types.rs:317-339) */
double orc_average_value_in_rect(const uint16_t *img, uint32_t w, uint32_t ox, uint32_t oy,
                                 const uint16_t r[4]) {
    uint64_t sum = 0, count = 0;
    for (uint32_t y = r[1] + oy; y < oy + r[3]; ++y)
        for (uint32_t x = r[0] + ox; x < ox + r[2]; ++x) {
            count += 1;
            sum += img[(size_t)y * w + x];
        }
    if (count == 0) return 0.0;
    return (double)sum / (double)count;
}

typedef struct {
    const uint16_t *img;
    uint32_t w, h;
    const uint64_t *sat; /* (w+1)*(h+1) or NULL */
} frame_view;

static double rect_avg(const frame_view *fv, uint32_t ox, uint32_t oy, const uint16_t r[4]) {
    if (!fv->sat) return orc_average_value_in_rect(fv->img, fv->w, ox, oy, r);
    if (r[2] <= r[0] || r[3] <= r[1]) return 0.0; /* empty range -> count == 0 */
    uint32_t x0 = ox + r[0], y0 = oy + r[1], x1 = ox + r[2], y1 = oy + r[3];
    size_t s = (size_t)fv->w + 1;
    uint64_t sum = fv->sat[y1 * s + x1] - fv->sat[y0 * s + x1] - fv->sat[y1 * s + x0] + fv->sat[y0 * s + x0];
    uint64_t count = (uint64_t)(x1 - x0) * (uint64_t)(y1 - y0);
    return (double)sum / (double)count;
}

static uint64_t *build_sat(const uint16_t *img, uint32_t w, uint32_t h) {
    size_t s = (size_t)w + 1;
    uint64_t *sat = (uint64_t *)calloc(s * ((size_t)h + 1), sizeof(uint64_t));
    if (!sat) return NULL;
    for (uint32_t y = 0; y < h; ++y) {
        uint64_t run = 0;
        for (uint32_t x = 0; x < w; ++x) {
            run += img[(size_t)y * w + x];
            sat[(y + 1) * s + x + 1] = sat[y * s + x + 1] + run;
        }
    }
    return sat;
}

/* ------------------------------------------------------------------ SparseArray3D<u32> (meanshift.rs:14-68) */
typedef struct {
    int32_t *keys; /* 3 per slot */
    uint32_t *vals;
    uint8_t *used;
    size_t cap, len;
} sparse3;

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static size_t sp_hash(int32_t x, int32_t y, int32_t z) {
    uint64_t a = ((uint64_t)(uint32_t)x << 32) | (uint32_t)y;
    return (size_t)mix64(mix64(a) ^ (uint64_t)(uint32_t)z);
}
static int sp_init(sparse3 *s, size_t cap) {
    s->cap = cap; s->len = 0;
    s->keys = (int32_t *)malloc(cap * 3 * sizeof(int32_t));
    s->vals = (uint32_t *)malloc(cap * sizeof(uint32_t));
    s->used = (uint8_t *)calloc(cap, 1);
    return s->keys && s->vals && s->used;
}
static void sp_free(sparse3 *s) { free(s->keys); free(s->vals); free(s->used); }
static uint32_t *sp_slot(sparse3 *s, int32_t x, int32_t y, int32_t z, int insert);
static void sp_grow(sparse3 *s) {
    sparse3 n;
    sp_init(&n, s->cap * 2);
    for (size_t i = 0; i < s->cap; ++i)
        if (s->used[i]) *sp_slot(&n, s->keys[i * 3], s->keys[i * 3 + 1], s->keys[i * 3 + 2], 1) = s->vals[i];
    sp_free(s);
    *s = n;
}
static uint32_t *sp_slot(sparse3 *s, int32_t x, int32_t y, int32_t z, int insert) {
    if (insert && (s->len + 1) * 2 > s->cap) sp_grow(s);
    size_t m = s->cap - 1, i = sp_hash(x, y, z) & m;
    while (s->used[i]) {
        if (s->keys[i * 3] == x && s->keys[i * 3 + 1] == y && s->keys[i * 3 + 2] == z) return &s->vals[i];
        i = (i + 1) & m;
    }
    if (!insert) return NULL;
    s->used[i] = 1; s->keys[i * 3] = x; s->keys[i * 3 + 1] = y; s->keys[i * 3 + 2] = z; s->vals[i] = 0; /* :59-62 default 0 */
    s->len++;
    return &s->vals[i];
}
/* Index: missing key reads the default WITHOUT inserting (meanshift.rs:25-27). */
static inline uint32_t sp_get(sparse3 *s, int32_t x, int32_t y, int32_t z) {
    uint32_t *p = sp_slot(s, x, y, z, 0);
    return p ? *p : 0u;
}
/* IndexMut + `+=` (meanshift.rs:34-40); u32 wraps as in a release build. */
static inline void sp_add(sparse3 *s, int32_t x, int32_t y, int32_t z, uint32_t v) { *sp_slot(s, x, y, z, 1) += v; }

static int cell_cmp(const void *a, const void *b) {
    const int32_t *p = (const int32_t *)a, *q = (const int32_t *)b;
    for (int k = 0; k < 3; ++k)
        if (p[k] != q[k]) return p[k] < q[k] ? -1 : 1;
    return 0;
}
static void sp_export(const sparse3 *s, int32_t *out, uint32_t cap, uint32_t *count) {
    if (count) *count = (uint32_t)s->len;
    if (!out) return;
    int32_t *tmp = (int32_t *)malloc((s->len ? s->len : 1) * 4 * sizeof(int32_t));
    size_t n = 0;
    for (size_t i = 0; i < s->cap; ++i)
        if (s->used[i]) {
            tmp[n * 4] = s->keys[i * 3]; tmp[n * 4 + 1] = s->keys[i * 3 + 1]; tmp[n * 4 + 2] = s->keys[i * 3 + 2];
            tmp[n * 4 + 3] = (int32_t)s->vals[i];
            n++;
        }
    qsort(tmp, n, 4 * sizeof(int32_t), cell_cmp);
    if (n > cap) n = cap;
    memcpy(out, tmp, n * 4 * sizeof(int32_t));
    free(tmp);
}

/* ------------------------------------------------------------------ mean shift kernel (meanshift.rs:228-252) */
void orc_build_kernel(uint32_t size, float variance, float *out) {
    int32_t half = (int32_t)(size / 2);
    for (uint32_t i = 0; i < size * size * size; ++i) { /* from_fn :155-173 */
        uint32_t z = i / (size * size), rest = i % (size * size), y = rest / size, x = rest % size;
        int32_t dx = (int32_t)x - half, dy = (int32_t)y - half, dz = (int32_t)z - half;
        int32_t norm = dx * dx + dy * dy + dz * dz;
        out[i] = expf(-1.0f * (float)norm / (2.0f * variance)); /* :231 */
    }
}

/* meanshift.rs:328-407 for Idx = i32 (min_value checks are no-ops). */
static void meanshift(sparse3 *acc, const int32_t init[3], const float *kern, uint32_t ks, uint32_t iterations,
                      int32_t out[3], int32_t *trace, uint32_t *steps) {
    int32_t pos[3] = {init[0], init[1], init[2]};
    uint32_t done = 0;
    if (trace) { trace[0] = pos[0]; trace[1] = pos[1]; trace[2] = pos[2]; }
    int32_t w = (int32_t)ks, half = w / 2;
    for (uint32_t it = 0; it < iterations; ++it) {
        float num[3] = {0.0f, 0.0f, 0.0f}, den = 0.0f;
        for (int32_t x = -half; x < w - half; ++x)
            for (int32_t y = -half; y < w - half; ++y)
                for (int32_t z = -half; z < w - half; ++z) {
                    int32_t ax = wrap_add_i32(pos[0], x), ay = wrap_add_i32(pos[1], y), az = wrap_add_i32(pos[2], z);
                    uint32_t factor = sp_get(acc, ax, ay, az);       /* :360-362 */
                    if (factor == 0) continue;                        /* :364 */
                    float influence = kern[(uint32_t)(z + half) * ks * ks + (uint32_t)(y + half) * ks + (uint32_t)(x + half)]; /* :371, index :78-88 */
                    float fa[3] = {(float)ax, (float)ay, (float)az};  /* :373-375 */
                    float ff = (float)factor;                          /* :377 */
                    float wgt = influence * ff;
                    for (int k = 0; k < 3; ++k) num[k] = num[k] + fa[k] * wgt; /* :378 */
                    den = den + influence * ff;                        /* :379 */
                }
        if (den == 0.0f) break;                                        /* :385-388 */
        for (int k = 0; k < 3; ++k) pos[k] = f32_as_i32(num[k] / den); /* :391-394 */
        done++;
        if (trace) { trace[done * 3] = pos[0]; trace[done * 3 + 1] = pos[1]; trace[done * 3 + 2] = pos[2]; }
    }
    out[0] = pos[0]; out[1] = pos[1]; out[2] = pos[2];                 /* :397-406 clamp to i32::MIN is a no-op */
    if (steps) *steps = done;
}

/* ------------------------------------------------------------------ constants (prediction.rs:270-286) */
#define ZSCALEFACTOR 1
#define GUESS_GRID_PARTS 20
#define ROT_GRID_PARTS 120
static const double MAX_VARIANCE_ROT = 400.0;
static const float MAX_VARIANCE_OFFSET = 5200.0f;

int orc_patch_grid(uint32_t w, uint32_t h, const orc_model *m, uint32_t *nx, uint32_t *ny) {
    if (!m || m->stepwidth == 0 || m->subimage_width == 0 || m->subimage_height == 0) return -1;
    if (w < m->subimage_width || h < m->subimage_height) return -1; /* reference underflows / panics (:546,:548,:565) */
    uint32_t lw = m->subimage_width / 2, rw = m->subimage_width - lw;
    uint32_t lh = m->subimage_height / 2, rh = m->subimage_height - lh;
    uint32_t cx = 0, cy = 0;
    for (uint32_t x = lw; x < w - rw; x += m->stepwidth) cx++;
    for (uint32_t y = lh; y < h - rh; y += m->stepwidth) cy++;
    if (nx) *nx = cx;
    if (ny) *ny = cy;
    return 0;
}

static int32_t walk_tree(const orc_forest *f, uint32_t t, const frame_view *fv, uint32_t ox, uint32_t oy) {
    int32_t cur = f->roots[t];
    while (cur >= 0) {
        const orc_node *nd = &f->nodes[cur];
        double avg1 = rect_avg(fv, ox, oy, nd->r1);           /* houghforest.rs:186 */
        double avg2 = rect_avg(fv, ox, oy, nd->r2);           /* :187 */
        cur = (avg1 - avg2 > nd->threshold) ? nd->child_one : nd->child_zero; /* :188-192 */
    }
    return ~cur;
}

int orc_predict(const orc_forest *f, const orc_model *m, const uint16_t *img, uint32_t w, uint32_t h,
                const float K[9], const float *midp_guess, const double *rot_guess, int rect_mode,
                orc_pose *out, const orc_taps *taps) {
    uint32_t nx, ny;
    if (!f || !img || !K || !out || orc_patch_grid(w, h, m, &nx, &ny)) return -1;
    const uint32_t T = f->n_trees;
    if (T == 0) return -1;
    const uint32_t sw = m->subimage_width, sh = m->subimage_height;
    if (sw > 65535 || sh > 65535) return -1;
    const uint32_t left_w = sw / 2, right_w = sw - left_w, left_h = sh / 2, right_h = sh - left_h; /* :535-538 */

    frame_view fv = {img, w, h, NULL};
    uint64_t *sat = NULL;
    if (rect_mode == ORC_RECT_SAT) { sat = build_sat(img, w, h); if (!sat) return -2; fv.sat = sat; }

    float Kinv[9];
    inv3_f32(K, Kinv);

    uint32_t guess_pos_grid[GUESS_GRID_PARTS * GUESS_GRID_PARTS];                    /* :529 */
    uint32_t *guess_rot_grid = (uint32_t *)calloc(GUESS_GRID_PARTS * GUESS_GRID_PARTS * GUESS_GRID_PARTS, 4); /* :530-533 */
    memset(guess_pos_grid, 0, sizeof guess_pos_grid);
    sparse3 mid, rot;                                                                  /* :541-542 */
    sp_init(&mid, 1024); sp_init(&rot, 1024);
    int32_t *leafs = (int32_t *)malloc(sizeof(int32_t) * T);
    const uint16_t whole[4] = {0, 0, (uint16_t)sw, (uint16_t)sh};

    uint32_t pidx = 0;
    for (uint32_t y = left_h; y < h - right_h; y += m->stepwidth)                     /* :544-546 */
        for (uint32_t x = left_w; x < w - right_w; x += m->stepwidth, ++pidx) {       /* :547-548 */
            uint16_t z = img[(size_t)y * w + x];                                       /* :551 */
            float p3[3];
            to3d(Kinv, (float)x, (float)y, (float)z, p3);                              /* :554 */
            uint32_t ox = x - left_w, oy = y - left_h;                                 /* :560-565 */
            int nonbg = rect_avg(&fv, ox, oy, whole) > 0.0;                            /* :567-571 */
            if (taps && taps->patch_flags) taps->patch_flags[pidx] = (uint8_t)nonbg;
            if (!nonbg) {
                if (taps && taps->leaf_idx) for (uint32_t t = 0; t < T; ++t) taps->leaf_idx[(size_t)pidx * T + t] = -1;
                continue;
            }
            for (uint32_t t = 0; t < T; ++t) leafs[t] = walk_tree(f, t, &fv, ox, oy); /* :572 -> stamm forest_predictions */
            if (taps && taps->leaf_idx) memcpy(&taps->leaf_idx[(size_t)pidx * T], leafs, sizeof(int32_t) * T);

            double prob = 0.0;                                                         /* :582 */
            for (uint32_t t = 0; t < T; ++t) prob = prob + f->leaf_prob[leafs[t]];
            prob = prob / (double)T;
            if (!(prob > 0.7)) continue;                                               /* :584 */
            if (taps && taps->patch_flags) taps->patch_flags[pidx] |= 2;

            for (uint32_t t = 0; t < T; ++t) {                                         /* :586 */
                int32_t L = leafs[t];
                double lp = f->leaf_prob[L];
                if (!(lp > 0.0)) continue;                                             /* :590 */
                uint32_t ob = f->off_begin[L], oe = f->off_begin[L + 1];
                uint32_t rb = f->rot_begin[L], re = f->rot_begin[L + 1];
                if (oe == ob || re == rb) { free(leafs); free(guess_rot_grid); sp_free(&mid); sp_free(&rot); free(sat); return -3; } /* reference panics (:594 div by 0, :600 unwrap) */
                uint32_t valtoadd = (uint32_t)(f64_as_usize(1000.0 * lp) / (uint64_t)(oe - ob)); /* :594-595 */

                double rmean[3], rcov[9];
                mean_cov3_f64(&f->rotations[(size_t)rb * 3], re - rb, rmean, rcov);
                if (trace3_f64(rcov) <= MAX_VARIANCE_ROT) {                            /* :600 */
                    for (uint32_t i = rb; i < re; ++i) {                               /* :601 */
                        int32_t r[3];
                        uint32_t ru[3], rough[3];
                        for (int k = 0; k < 3; ++k) {
                            r[k] = f64_as_i32(f->rotations[(size_t)i * 3 + k] * (double)ROT_GRID_PARTS / 360.0) + ROT_GRID_PARTS / 2; /* :605-613 */
                            if (r[k] >= ROT_GRID_PARTS) r[k] = r[k] - ROT_GRID_PARTS;  /* :616-627 single step */
                            else if (r[k] < 0) r[k] = ROT_GRID_PARTS + r[k];
                            ru[k] = (uint32_t)r[k];
                            rough[k] = ru[k] * GUESS_GRID_PARTS / ROT_GRID_PARTS;       /* :630-632 (u32 wrapping mul) */
                        }
                        if (rough[0] >= GUESS_GRID_PARTS || rough[1] >= GUESS_GRID_PARTS || rough[2] >= GUESS_GRID_PARTS) {
                            free(leafs); free(guess_rot_grid); sp_free(&mid); sp_free(&rot); free(sat); return -4; /* reference: index out of bounds panic (:636) */
                        }
                        sp_add(&rot, (int32_t)ru[0], (int32_t)ru[1], (int32_t)ru[2], valtoadd);  /* :635 */
                        guess_rot_grid[rough[2] * 400 + rough[1] * 20 + rough[0]] += valtoadd;  /* :636, index meanshift.rs:78-88 */
                    }
                }

                float omean[3], ocov[9];
                mean_cov3_f32(&f->offsets[(size_t)ob * 3], oe - ob, omean, ocov);
                if (trace3_f32(ocov) <= MAX_VARIANCE_OFFSET) {                         /* :643 */
                    for (uint32_t i = ob; i < oe; ++i) {                               /* :644 */
                        float np[3];
                        for (int k = 0; k < 3; ++k) np[k] = p3[k] - f->offsets[(size_t)i * 3 + k]; /* :647 */
                        if (np[2] < 0.0f) continue;                                    /* :650 */
                        float p2[2];
                        to2d(K, np, p2);                                               /* :661 */
                        float x2d = p2[0] > 0.0f ? p2[0] : 0.0f;                       /* max! :19-21 */
                        x2d = x2d < (float)(w - 1) ? x2d : (float)(w - 1);             /* min! :23-25, :662 */
                        float y2d = p2[1] > 0.0f ? p2[1] : 0.0f;
                        y2d = y2d < (float)(h - 1) ? y2d : (float)(h - 1);             /* :663 */
                        float z3d = np[2] / (float)ZSCALEFACTOR;                       /* :666 */
                        sp_add(&mid, f32_as_i32(np[0]), f32_as_i32(np[1]), f32_as_i32(z3d), valtoadd); /* :667 */
                        uint64_t gx = f32_as_usize(x2d) * GUESS_GRID_PARTS / (uint64_t)w; /* :671-672 */
                        uint64_t gy = f32_as_usize(y2d) * GUESS_GRID_PARTS / (uint64_t)h; /* :673-674 */
                        guess_pos_grid[gy * GUESS_GRID_PARTS + gx] += valtoadd;        /* :675-676 */
                    }
                }
            }
        }
    free(leafs);

    /* best 2d grid position: first strictly-greater wins, start (0,0) (:694-702) */
    uint32_t prev_max = 0; size_t best_idx = 0;
    for (size_t i = 0; i < GUESS_GRID_PARTS * GUESS_GRID_PARTS; ++i)
        if (guess_pos_grid[i] > prev_max) { prev_max = guess_pos_grid[i]; best_idx = i; }
    size_t gpw = w / GUESS_GRID_PARTS, gph = h / GUESS_GRID_PARTS;                     /* :706-707 */
    size_t mxg = best_idx % GUESS_GRID_PARTS, myg = best_idx / GUESS_GRID_PARTS;       /* :708-709 */
    uint64_t zsum = 0; size_t zcnt = 0;                                                /* :717-720 */
    for (size_t yy = gph * myg; yy < gph * myg + gph; ++yy)
        for (size_t xx = gpw * mxg; xx < gpw * mxg + gpw; ++xx) {
            uint16_t v = img[yy * w + xx];
            if (v > 0) { zsum += v; zcnt += 1; }
        }
    float meanz = zcnt > 0 ? (float)((double)zsum / (double)zcnt) : 0.0f;             /* :721-725 */
    float max_x = ((float)mxg + 0.5f) * (float)gpw, max_y = ((float)myg + 0.5f) * (float)gph; /* :727-728 */
    float max3d[3];
    to3d(Kinv, max_x, max_y, meanz, max3d);                                            /* :729 */
    int32_t guessmid[3] = {f32_as_i32(max3d[0]), f32_as_i32(max3d[1]), f32_as_i32(max3d[2]) / ZSCALEFACTOR}; /* :750 */

    /* best rotation guess: iteration x fastest, then y, then z (meanshift.rs:114-138); strictly greater (:733-742) */
    uint32_t rbest[3] = {0, 0, 0}, oldc = 0;
    for (uint32_t z = 0; z < GUESS_GRID_PARTS; ++z)
        for (uint32_t y = 0; y < GUESS_GRID_PARTS; ++y)
            for (uint32_t x = 0; x < GUESS_GRID_PARTS; ++x) {
                uint32_t c = guess_rot_grid[z * 400 + y * 20 + x];
                if (c > 0 && c > oldc) { rbest[0] = x; rbest[1] = y; rbest[2] = z; oldc = c; }
            }
    double guessrot_deg[3];
    for (int k = 0; k < 3; ++k) guessrot_deg[k] = ((double)rbest[k] * 360.0 + 180.0) / (double)GUESS_GRID_PARTS; /* :745-747 */

    if (midp_guess) {                                                                  /* :437-441 */
        guessmid[0] = f32_as_i32(midp_guess[0]); guessmid[1] = f32_as_i32(midp_guess[1]);
        guessmid[2] = f32_as_i32(midp_guess[2]) / ZSCALEFACTOR;
    }
    if (rot_guess)                                                                     /* :444-453 */
        for (int k = 0; k < 3; ++k) guessrot_deg[k] = rot_guess[k] * 180.0 / 3.14159 + 180.0;
    int32_t guessrot[3];
    for (int k = 0; k < 3; ++k) guessrot[k] = f64_as_i32(guessrot_deg[k] * (double)ROT_GRID_PARTS / 360.0); /* :458-460 */

    float *kern = (float *)malloc(sizeof(float) * 8000);
    orc_build_kernel(20, m->gaussian_sigma, kern);                                     /* :314 sigma passed as the variance */

    int32_t res_mid[3], res_rot[3];
    meanshift(&mid, guessmid, kern, 20, m->meanshift_iterations, res_mid,
              taps ? taps->ms_trace_mid : NULL, taps ? taps->ms_steps_mid : NULL);     /* :469 */
    meanshift(&rot, guessrot, kern, 20, m->meanshift_iterations, res_rot,
              taps ? taps->ms_trace_rot : NULL, taps ? taps->ms_steps_rot : NULL);     /* :472 */
    free(kern);

    for (int k = 0; k < 3; ++k)
        out->rotation[k] = ((double)res_rot[k] - (double)ROT_GRID_PARTS / 2.0) / (double)(ROT_GRID_PARTS / 2) * 3.14159; /* :477-482 */
    out->mid_point[0] = (float)res_mid[0];                                             /* :486-488 */
    out->mid_point[1] = (float)res_mid[1];
    out->mid_point[2] = (float)(int32_t)((uint32_t)res_mid[2] * (uint32_t)ZSCALEFACTOR);

    if (taps) {
        if (taps->pos_grid) memcpy(taps->pos_grid, guess_pos_grid, sizeof guess_pos_grid);
        if (taps->rot_grid) memcpy(taps->rot_grid, guess_rot_grid, 8000 * 4);
        if (taps->guess_mid) memcpy(taps->guess_mid, guessmid, 12);
        if (taps->guess_rot) memcpy(taps->guess_rot, guessrot, 12);
        if (taps->guess_rot_deg) memcpy(taps->guess_rot_deg, guessrot_deg, 24);
        sp_export(&mid, taps->mid_cells, taps->mid_cap, taps->mid_count);
        sp_export(&rot, taps->rot_cells, taps->rot_cap, taps->rot_count);
    }
    free(guess_rot_grid); sp_free(&mid); sp_free(&rot); free(sat);
    return 0;
}

/* ------------------------------------------------------------------ predict_mask (prediction.rs:850-905) */
static inline uint8_t f64_as_u8(double v) {
    if (v != v || v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

int orc_predict_mask(const orc_forest *f, const orc_model *m, const uint16_t *img, uint32_t w, uint32_t h,
                     int rect_mode, uint8_t *mask) {
    uint32_t nx, ny;
    if (!f || !img || !mask || orc_patch_grid(w, h, m, &nx, &ny) || f->n_trees == 0) return -1;
    const uint32_t T = f->n_trees, sw = m->subimage_width, sh = m->subimage_height, step = m->stepwidth;
    const uint32_t left_w = sw / 2, right_w = sw - left_w, left_h = sh / 2, right_h = sh - left_h;   /* :853-856 */
    frame_view fv = {img, w, h, NULL};
    uint64_t *sat = NULL;
    if (rect_mode == ORC_RECT_SAT) { sat = build_sat(img, w, h); if (!sat) return -2; fv.sat = sat; }
    const uint16_t whole[4] = {0, 0, (uint16_t)sw, (uint16_t)sh};
    memset(mask, 0, (size_t)w * h);                                                     /* ImageBuffer::new zero-fills (:852) */
    for (uint32_t y = left_h; y < h - right_h; y += step)                               /* :858 */
        for (uint32_t x = left_w; x < w - right_w; x += step) {                         /* :860 */
            uint32_t ox = x - left_w, oy = y - left_h;
            if (!(rect_avg(&fv, ox, oy, whole) > 0.0)) continue;                        /* :870-878 */
            double prob = 0.0;                                                          /* :881-882 */
            for (uint32_t t = 0; t < T; ++t) prob = prob + f->leaf_prob[walk_tree(f, t, &fv, ox, oy)];
            prob = prob / (double)T;
            uint8_t pv = f64_as_u8(prob * 255.0);                                       /* :883 */
            for (uint32_t i = 0; i < step; ++i)                                         /* :884 */
                for (uint32_t j = 0; j < step; ++j) {
                    if (x + i < step / 2 || y + j < step / 2) continue;                 /* :886 */
                    if (x + i - step / 2 >= w || y + j - step / 2 >= h) continue;       /* :889 */
                    mask[(size_t)(y + j - step / 2) * w + (x + i - step / 2)] = pv;     /* :892-896 */
                }
        }
    free(sat);
    return 0;
}

/* ------------------------------------------------------------------ build_hough_image, voting stage (prediction.rs:760-840) */
int orc_hough_image(const orc_forest *f, const orc_model *m, const uint16_t *img, uint32_t w, uint32_t h,
                    const float K[9], int rect_mode, uint16_t *out) {
    uint32_t nx, ny;
    if (!f || !img || !out || !K || orc_patch_grid(w, h, m, &nx, &ny) || f->n_trees == 0) return -1;
    const uint32_t T = f->n_trees, sw = m->subimage_width, sh = m->subimage_height, step = m->stepwidth;
    const uint32_t left_w = sw / 2, right_w = sw - left_w, left_h = sh / 2, right_h = sh - left_h;   /* :767-770 */
    frame_view fv = {img, w, h, NULL};
    uint64_t *sat = NULL;
    if (rect_mode == ORC_RECT_SAT) { sat = build_sat(img, w, h); if (!sat) return -2; fv.sat = sat; }
    float Kinv[9];
    inv3_f32(K, Kinv);
    const uint16_t whole[4] = {0, 0, (uint16_t)sw, (uint16_t)sh};
    memset(out, 0, (size_t)w * h * 2);                                                  /* :766 */
    for (uint32_t y = left_h; y < h - right_h; y += step)                               /* :773 */
        for (uint32_t x = left_w; x < w - right_w; x += step) {                         /* :775 */
            uint16_t z = img[(size_t)y * w + x];                                        /* :777 */
            float p3[3];
            to3d(Kinv, (float)x, (float)y, (float)z, p3);                               /* :779 */
            uint32_t ox = x - left_w, oy = y - left_h;
            if (!(rect_avg(&fv, ox, oy, whole) > 0.0)) continue;                        /* :790-798 */
            for (uint32_t t = 0; t < T; ++t) {                                          /* :803 */
                int32_t L = walk_tree(f, t, &fv, ox, oy);
                double lp = f->leaf_prob[L];
                if (!(lp >= 0.95)) continue;                                            /* :805 */
                uint32_t ob = f->off_begin[L], oe = f->off_begin[L + 1];
                if (oe == ob) { free(sat); return -3; }                                 /* reference divides by zero (:807) */
                uint16_t valtoadd = (uint16_t)(f64_as_usize(255.0 * lp) / (uint64_t)(oe - ob));   /* :807-808 */
                for (uint32_t i = ob; i < oe; ++i) {                                    /* :813 */
                    float np[3], p2[2];
                    for (int k = 0; k < 3; ++k) np[k] = p3[k] - f->offsets[(size_t)i * 3 + k];     /* :814 */
                    to2d(K, np, p2);                                                    /* :815 */
                    int32_t vx = f32_as_i32(p2[0]), vy = f32_as_i32(p2[1]);             /* :816 */
                    if (vx < 0 || (uint32_t)vx >= w || vy < 0 || (uint32_t)vy >= h) continue;       /* :818-831 */
                    out[(size_t)vy * w + vx] = (uint16_t)(out[(size_t)vy * w + vx] + valtoadd);     /* :832, u16 wraps */
                }
            }
        }
    free(sat);
    return 0;
}

/* ---- imageproc 0.12.0 filter::gaussian_blur_f32 (external crate, /root/reference Cargo.toml:26, Cargo.lock:555-, call at
 * src/hough/prediction.rs:844).  ITS SOURCE IS NOT IN THE CONTAINER: this restates the crate's published algorithm --
 * PARITY UNPINNED.
 *   gaussian_blur_f32(image, sigma): assert!(sigma > 0.0); separable_filter_equal(image, &gaussian_kernel_f32(sigma))
 *   gaussian_kernel_f32: kernel_radius = (2.0 * sigma).ceil() as usize; data[radius +- i] = gaussian(i as f32, sigma),
 *       i = 0..=radius (the kernel is NOT renormalised)
 *   gaussian(x, r) = ((2.0 * PI).sqrt() * r).recip() * (-x.powi(2) / (2.0 * r.powi(2))).exp()          (all f32)
 *   separable_filter: horizontal_filter -> an image of the SAME pixel type (u16), then vertical_filter over that;
 *       per output pixel: acc = 0; for (i, k) in kernel.enumerate(): p = clamp(pos + i - len/2, 0, side-1);
 *       acc = acc + (pixel as f32) * k;   result = <u16 as Clamp<f32>>::clamp(acc):
 *       if x < 65535.0 { if x > 0.0 { x as u16 } else { 0 } } else { 65535 }                                */
static uint16_t clamp_f32_u16(float x) {
    if (x < 65535.0f) return x > 0.0f ? (uint16_t)x : (uint16_t)0;
    return 65535;
}

int orc_gaussian_kernel_f32(float sigma, float *out, uint32_t cap, uint32_t *len) {
    if (!(sigma > 0.0f) || !len) return -1;                              /* assert!(sigma > 0.0) */
    float r2 = ceilf(2.0f * sigma);
    if (!(r2 <= 1.0e6f)) return -1;
    uint32_t radius = (uint32_t)r2;
    *len = 2 * radius + 1;
    if (!out) return 0;
    if (cap < *len) return -1;
    const float pi = 3.14159274101257324f;                               /* std::f32::consts::PI */
    const float norm = 1.0f / (sqrtf(2.0f * pi) * sigma);                /* ((2.0 * PI).sqrt() * r).recip() */
    for (uint32_t i = 0; i <= radius; ++i) {
        float x = (float)i;
        float v = norm * expf(-(x * x) / (2.0f * (sigma * sigma)));
        out[radius + i] = v;
        out[radius - i] = v;
    }
    return 0;
}

int orc_gaussian_blur_u16(const uint16_t *in, uint32_t w, uint32_t h, float sigma, uint16_t *out) {
    uint32_t klen = 0;
    if (!in || !out || orc_gaussian_kernel_f32(sigma, NULL, 0, &klen)) return -1;
    if (w == 0 || h == 0) return 0;
    float *k = (float *)malloc((size_t)klen * sizeof(float));
    uint16_t *tmp = (uint16_t *)malloc((size_t)w * h * sizeof(uint16_t));
    if (!k || !tmp) { free(k); free(tmp); return -2; }
    orc_gaussian_kernel_f32(sigma, k, klen, &klen);
    const int64_t half = (int64_t)(klen / 2);
    for (uint32_t y = 0; y < h; ++y)                                      /* horizontal_filter */
        for (uint32_t x = 0; x < w; ++x) {
            float acc = 0.0f;
            for (uint32_t i = 0; i < klen; ++i) {
                int64_t xp = (int64_t)x + (int64_t)i - half;
                if (xp < 0) xp = 0;
                if (xp > (int64_t)w - 1) xp = (int64_t)w - 1;
                acc = acc + (float)in[(size_t)y * w + (size_t)xp] * k[i];
            }
            tmp[(size_t)y * w + x] = clamp_f32_u16(acc);
        }
    for (uint32_t y = 0; y < h; ++y)                                      /* vertical_filter */
        for (uint32_t x = 0; x < w; ++x) {
            float acc = 0.0f;
            for (uint32_t i = 0; i < klen; ++i) {
                int64_t yp = (int64_t)y + (int64_t)i - half;
                if (yp < 0) yp = 0;
                if (yp > (int64_t)h - 1) yp = (int64_t)h - 1;
                acc = acc + (float)tmp[(size_t)yp * w + x] * k[i];
            }
            out[(size_t)y * w + x] = clamp_f32_u16(acc);
        }
    free(k); free(tmp);
    return 0;
}

/* HoughPrediction::build_hough_image in full (prediction.rs:760-845): votes (:760-840) + blur (:844). */
int orc_build_hough_image(const orc_forest *f, const orc_model *m, const uint16_t *img, uint32_t w, uint32_t h,
                          const float K[9], int rect_mode, uint16_t *out) {
    uint16_t *votes = (uint16_t *)malloc((size_t)w * h * sizeof(uint16_t));
    if (!votes) return -2;
    int rc = orc_hough_image(f, m, img, w, h, K, rect_mode, votes);
    if (rc == 0) rc = orc_gaussian_blur_u16(votes, w, h, m->gaussian_sigma, out);          /* :844 */
    free(votes);
    return rc;
}

/* HoughPrediction::predict_parameter_from2dhough (prediction.rs:343-367). */
int orc_predict_from2dhough(const orc_forest *f, const orc_model *m, const uint16_t *img, uint32_t w, uint32_t h,
                            const float K[9], int rect_mode, orc_pose *out) {
    if (!out) return -1;
    uint16_t *hough = (uint16_t *)malloc((size_t)w * h * sizeof(uint16_t));
    if (!hough) return -2;
    int rc = orc_build_hough_image(f, m, img, w, h, K, rect_mode, hough);                  /* :348 */
    if (rc) { free(hough); return rc; }
    /* (0..w*h).max_by_key(|i| hough[(i % w, i / w)]): Iterator::max_by_key returns the LAST maximal element (:351-356) */
    uint32_t best = 0;
    for (uint32_t i = 0; i < w * h; ++i)
        if (hough[i] >= hough[best]) best = i;
    uint32_t x = best % w, y = best / w;                                                   /* :357-358 */
    uint16_t z = img[(size_t)y * w + x];                                                   /* :359 */
    float Kinv[9], p[3];
    inv3_f32(K, Kinv);
    to3d(Kinv, (float)x, (float)y, (float)z, p);                                           /* :360 */
    memset(out, 0, sizeof *out);
    out->mid_point[0] = p[0]; out->mid_point[1] = p[1]; out->mid_point[2] = p[2];           /* :361-365 */
    free(hough);
    return 0;
}

int orc_predict_batch(const orc_forest *f, const orc_model *m, const uint16_t *imgs, uint32_t n,
                      uint32_t w, uint32_t h, const float K[9], const float *midp_guess,
                      const double *rot_guess, int rect_mode, int threads, orc_pose *out) {
    int rc = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        int r = orc_predict(f, m, imgs + (size_t)i * w * h, w, h, K, midp_guess ? midp_guess + i * 3 : NULL,
                            rot_guess ? rot_guess + i * 3 : NULL, rect_mode, &out[i], NULL);
        if (r) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}
