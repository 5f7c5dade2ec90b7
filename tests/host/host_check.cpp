// host_check.cpp -- drives the host-only logic of the runtime (depthhead_amd/csrc/dh_host.cpp, dh_biwi.cpp) under the CPU
// sanitizers: built by tests/test_host_sanitize.py with g++ -fsanitize=address,undefined and once more with
// -fsanitize=thread.  Prints "host_check ok" and exits 0; any check that fails prints its line and exits 1.
// Covers: forest validation (good / every refused kind), patch grids, tile selection across geometries (with the invariants
// the kernels rely on), upload chunk plans, the run-length payload scanner / packer on well-formed, redundant and malformed
// payloads (incl. every truncation point of a payload), the host BIWI parsers, knob parsing, the exception guard, and
// dh_parallel_for_ / the thread-local error slots under concurrency.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../depthhead_amd/csrc/dh_host.h"

static int g_fail = 0;
#define CHECK(c)                                                               \
    do {                                                                       \
        if (!(c)) { fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #c); ++g_fail; } \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }
static uint32_t rnd(uint32_t n) { return n ? rnd() % n : 0; }

// ------------------------------------------------------------------ a small forest: T complete trees of depth D
struct Flat {
    std::vector<int32_t> roots;
    std::vector<dh_node> nodes;
    std::vector<double> prob;
    std::vector<uint32_t> ob, rb;
    std::vector<float> off;
    std::vector<double> rot;
    dh_forest_desc desc() const {
        dh_forest_desc d{};
        d.n_trees = (uint32_t)roots.size(); d.roots = roots.data(); d.n_nodes = (uint32_t)nodes.size(); d.nodes = nodes.data();
        d.n_leaves = (uint32_t)prob.size(); d.leaf_prob = prob.data(); d.off_begin = ob.data(); d.rot_begin = rb.data();
        d.offsets = off.data(); d.rotations = rot.data();
        return d;
    }
};
static Flat make_forest(int T, int D, bool uniform) {
    Flat f;
    f.ob.push_back(0); f.rb.push_back(0);
    for (int t = 0; t < T; ++t) {
        const int base = (int)f.nodes.size(), inner = (1 << D) - 1;
        f.roots.push_back(base);
        for (int i = 0; i < inner; ++i) {
            dh_node n{};
            const int w1 = uniform ? 24 : 1 + (int)rnd(40), h1 = uniform ? 24 : 1 + (int)rnd(40);
            const int w2 = uniform ? 24 : (int)rnd(40), h2 = uniform ? 24 : (int)rnd(40);
            n.r1[0] = (uint16_t)rnd(80 - w1 + 1); n.r1[1] = (uint16_t)rnd(80 - h1 + 1); n.r1[2] = (uint16_t)(n.r1[0] + w1); n.r1[3] = (uint16_t)(n.r1[1] + h1);
            n.r2[0] = (uint16_t)rnd(80 - w2 + 1); n.r2[1] = (uint16_t)rnd(80 - h2 + 1); n.r2[2] = (uint16_t)(n.r2[0] + w2); n.r2[3] = (uint16_t)(n.r2[1] + h2);
            n.threshold = (double)rnd(512) - 256.0 + (rnd(2) ? 0.5 : 0.0);
            const int l = 2 * i + 1, r = 2 * i + 2;
            auto child = [&](int c) -> int32_t {
                if (c < inner) return base + c;
                const int leaf = (int)f.prob.size();
                const bool pos = rnd(2);
                const uint32_t nv = pos ? 2 + rnd(6) : 0;
                f.prob.push_back(pos ? 0.5 + 0.5 * (rnd(100) / 100.0) : 0.0);
                for (uint32_t k = 0; k < nv; ++k) {
                    for (int j = 0; j < 3; ++j) { f.off.push_back((float)rnd(300) - 150.0f); f.rot.push_back((double)rnd(120) - 60.0); }
                }
                f.ob.push_back(f.ob.back() + nv); f.rb.push_back(f.rb.back() + nv);
                return ~leaf;
            };
            n.child_zero = child(l); n.child_one = child(r);
            f.nodes.push_back(n);
        }
    }
    return f;
}

static int create(const Flat &f, std::string *msg = nullptr) {
    dh_forest_desc d = f.desc();
    dh_forest *h = nullptr;
    int rc = dh_forest_build_(&d, &h);
    if (msg) *msg = dh_err_get_();
    CHECK((rc == DH_OK) == (h != nullptr));
    delete h;
    return rc;
}

static void test_forest() {
    Flat good = make_forest(3, 5, true);
    CHECK(create(good) == DH_OK);
    {
        dh_forest_desc d = good.desc();
        dh_forest *h = nullptr;
        CHECK(dh_forest_build_(&d, &h) == DH_OK && h && h->uniform && h->rw == 24 && h->rh == 24 && h->max_depth == 5);
        std::vector<uint32_t> b4; std::vector<float> o4;
        dh_pack_off4_(*h, b4, o4);
        CHECK(b4.size() == good.prob.size() + 1 && o4.size() == ((size_t)b4.back() + 4) * 4);
        for (size_t L = 0; L + 1 < b4.size(); ++L) CHECK(b4[L] % 4 == 0 && b4[L + 1] - b4[L] >= good.ob[L + 1] - good.ob[L]);
        std::vector<NodeG> ng;
        dh_build_nodes_g_(*h, ng);
        CHECK(ng.size() == good.nodes.size());
        for (size_t i = 0; i < ng.size(); ++i) CHECK((ng[i].cc & 0xffffu) == 576u && (ng[i].cc >> 16) == 576u && ng[i].amb <= 4u + 1u);
        delete h;
    }
    CHECK(create(make_forest(2, 4, false)) == DH_OK);
    std::string msg;
    { Flat f = good; f.nodes[0].child_one = (int32_t)f.nodes.size() + 5; CHECK(create(f, &msg) == DH_EFOREST && msg.find("child out of range") != std::string::npos); }
    { Flat f = good; f.nodes[0].child_zero = 0; CHECK(create(f, &msg) == DH_EFOREST && msg.find("twice") != std::string::npos); }
    { Flat f = good; f.nodes[1].child_zero = f.nodes[1].child_one; CHECK(create(f) == DH_EFOREST); }
    { Flat f = good; f.nodes[2].r1[0] = 30; f.nodes[2].r1[2] = 20; CHECK(create(f, &msg) == DH_EFOREST && msg.find("negative extent") != std::string::npos); }
    { Flat f = good; f.roots[1] = ~((int32_t)f.prob.size() + 1); CHECK(create(f, &msg) == DH_EFOREST && msg.find("root") != std::string::npos); }
    { Flat f = good; f.nodes[3].threshold = 0.0 / 0.0; CHECK(create(f) == DH_EFOREST); }
    { Flat f = good; f.ob[0] = 1; CHECK(create(f) == DH_EFOREST); }
    { Flat f = good; f.nodes.back().child_one = ~(int32_t)f.prob.size(); CHECK(create(f, &msg) == DH_EFOREST && msg.find("leaf out of range") != std::string::npos); }
    {   // a voting leaf without votes (the reference divides by zero / unwraps None)
        Flat f = good;
        for (size_t L = 0; L < f.prob.size(); ++L) if (f.ob[L + 1] == f.ob[L]) { f.prob[L] = 0.9; break; }
        CHECK(create(f, &msg) == DH_EFOREST && msg.find("no offsets") != std::string::npos);
    }
    {   // a rotation whose bin leaves [0, 120) after the single wrap
        Flat f = good;
        for (size_t L = 0; L < f.prob.size(); ++L) if (f.prob[L] > 0.0) { f.rot[(size_t)f.rb[L] * 3] = 1000.0; break; }
        CHECK(create(f, &msg) == DH_EFOREST && msg.find("rotation bin") != std::string::npos);
    }
    { dh_forest_desc d = good.desc(); d.n_trees = 0; dh_forest *h = nullptr; CHECK(dh_forest_build_(&d, &h) == DH_EINVAL && !h); }
    { dh_forest *h = nullptr; CHECK(dh_forest_build_(nullptr, &h) == DH_EINVAL); }
    {   // a single-leaf tree (root < 0) beside a normal one
        Flat f = good; f.roots.push_back(~0);
        CHECK(create(f) == DH_OK);
    }
}

// ------------------------------------------------------------------ geometry
static void ref_grid(int w, int h, int s, int sw, int sh, int *nx, int *ny) {   // the loops of prediction.rs:535-548, 684-686
    const int lw = sw / 2, rw = sw - lw, lh = sh / 2, rh = sh - lh;
    *nx = *ny = 0;
    for (int x = lw; x < w - rw; x += s) ++*nx;
    for (int y = lh; y < h - rh; y += s) ++*ny;
}
static void test_geometry() {
    const int iters = getenv("HOST_CHECK_LIGHT") ? 200 : 3000;     // (the ThreadSanitizer build is after the threaded parts)
    for (int it = 0; it < iters; ++it) {
        dh_params p{};
        p.stepwidth = 1 + rnd(12); p.subimage_width = 8 + rnd(120); p.subimage_height = 8 + rnd(120); p.gaussian_sigma = 8.0f; p.meanshift_iterations = 20;
        const int w = (int)p.subimage_width + (int)rnd(700), h = (int)p.subimage_height + (int)rnd(500);
        int nx = -1, ny = -1, rx, ry;
        CHECK(dh_patch_grid_(p, w, h, &nx, &ny) == DH_OK);
        ref_grid(w, h, (int)p.stepwidth, (int)p.subimage_width, (int)p.subimage_height, &rx, &ry);
        CHECK(nx == rx && ny == ry);
        if (nx == 0 || ny == 0) continue;
        for (int uniform = 0; uniform < 2; ++uniform) {
            TileQuery q;
            q.params = p; q.n_trees = 1 + rnd(60); q.absorb_ok = uniform && rnd(4) != 0; q.top_levels = rnd(3) ? -1 : (int)rnd(9);
            const int rmax = (int)std::min<uint32_t>(std::min(p.subimage_width, p.subimage_height), 96u);
            q.f_rw = uniform ? 1 + (int)rnd(rmax) : 0; q.f_rh = uniform ? 1 + (int)rnd(rmax) : 0;
            q.lds_budget_kb = rnd(3) ? 0 : 20 + (int)rnd(140);
            if (rnd(8) == 0) { q.tile_x = 1 + (int)rnd(30); q.tile_y = 1 + (int)rnd(30); }

            Geom g;
            g.w = w; g.h = h; g.nx = nx; g.ny = ny; g.npatch = nx * ny; g.uniform = uniform;
            int rc = dh_choose_tile_(q, g);
            if (rc == 1) { CHECK(uniform); continue; }             // "retry on the general path"
            if (rc != DH_OK) { CHECK(rc == DH_ESIZE); continue; }
            // what the kernels rely on
            CHECK(g.px >= 1 && g.py >= 1 && g.px * g.py <= 1024 && g.px <= nx && g.py <= ny);
            CHECK(g.tiles_x * g.px >= nx && (g.tiles_x - 1) * g.px < nx && g.tiles_y * g.py >= ny && (g.tiles_y - 1) * g.py < ny);
            CHECK(g.tiles_x * g.tiles_y <= 65535 && g.win_cap == g.tiles_x * g.tiles_y * g.px * g.py && g.flag_words * 4 >= g.tiles_x * g.tiles_y);
            CHECK(g.lds <= 158u * 1024u || (q.tile_x > 0));
            const int step = (int)p.stepwidth, sw = (int)p.subimage_width, sh = (int)p.subimage_height;
            if (uniform) {
                const int m = 1 << g.swz_log2, bw = (g.px - 1) * step + sw - q.f_rw + 1, bh = (g.py - 1) * step + sh - q.f_rh + 1;
                CHECK(step % m == 0 && m <= 8 && g.swz_q % 4 == 0 && g.swz_q * m >= bw && g.ss_row >= g.swz_q * m && g.ss_row % 4 == 0);
                CHECK(g.ss_max == g.ss_row * bh);
                CHECK((long)(sh - q.f_rh + 1) * g.ss_row < 16384 + (q.tile_x > 0 ? 1 << 30 : 0));          // 14-bit packed rectangle offsets
                CHECK(g.box_rows == h - q.f_rh + 1 && g.box_plane % 4 == 0 && g.box_plane * m >= w - q.f_rw + 1 + 4);
                CHECK(g.box_ow % 4 == 0 && g.box_ow <= 256 - q.f_rw && g.box_parts * g.box_ow >= w - q.f_rw + 1);
                CHECK(g.box_oh % 32 == 0 && g.box_bands * g.box_oh >= g.box_rows && (g.box_bands - 1) * g.box_oh < g.box_rows);
                CHECK(g.top_levels >= 0 && g.top_levels <= 8 && (q.absorb_ok || g.top_levels == 0));
                const size_t top_words = q.absorb_ok ? (size_t)q.n_trees * (1u << g.top_levels) * 3 : 0;
                CHECK(g.lds == ((size_t)g.ss_max + (size_t)g.px * g.py * 2 + 16 + top_words) * 4);
            } else {
                CHECK((g.ss_row & 1) == 1 && g.ss_row >= (g.px - 1) * step + sw + 1);
                CHECK(g.ss_max == g.ss_row * ((g.py - 1) * step + sh + 1));
                CHECK(g.lds == ((size_t)g.ss_max + (size_t)g.px * g.py * 2 + 16) * 4);
            }
        }
    }
    dh_params p{4, 80, 80, 8.0f, 20};
    int nx, ny;
    CHECK(dh_patch_grid_(p, 79, 480, &nx, &ny) == DH_ESIZE);
    p.stepwidth = 0;
    CHECK(dh_patch_grid_(p, 640, 480, &nx, &ny) == DH_EINVAL);
    // the bench geometry keeps its measured tile
    {
        TileQuery q; q.params = dh_params{4, 80, 80, 8.0f, 20}; q.f_rw = q.f_rh = 24; q.n_trees = 10; q.absorb_ok = true;
        Geom g; g.w = 640; g.h = 480; g.uniform = true;
        CHECK(dh_patch_grid_(q.params, 640, 480, &g.nx, &g.ny) == DH_OK && g.nx == 140 && g.ny == 100);
        g.npatch = g.nx * g.ny;
        CHECK(dh_choose_tile_(q, g) == DH_OK && g.px % 4 == 0 && g.lds <= 79u * 1024u);
    }
}

static void test_chunk_plan() {
    int cs[DH_STAGE_EVENTS + 1];
    for (int m = 1; m <= 1200; ++m)
        for (int sc : {1, 2, 7, 16, 64, 100, 4096})
            for (int single = 0; single < 2; ++single) {
                const int n = dh_chunk_plan_(m, sc, single != 0, cs);
                CHECK(n >= 1 && n <= DH_STAGE_EVENTS && cs[0] == 0 && cs[n] == m);
                for (int k = 0; k < n; ++k) CHECK(cs[k + 1] > cs[k]);
                if (single) CHECK(n == 1);
            }
    CHECK(dh_chunk_plan_(0, 64, false, cs) == 0);
    // with the taps on a host batch of 40 frames is ONE device batch (the taps index it from frame 0)
    CHECK(dh_chunk_plan_(40, 64, true, cs) == 1 && dh_chunk_plan_(40, 64, false, cs) == 2);
}

// ------------------------------------------------------------------ run-length coded payloads
static void put32(std::vector<uint8_t> &b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
static std::vector<uint8_t> encode(const std::vector<uint16_t> &img, uint32_t W, uint32_t H, bool redundant) {
    std::vector<uint8_t> b;
    put32(b, W); put32(b, H);
    size_t p = 0;
    const size_t total = (size_t)W * H;
    if (redundant) { put32(b, 0); put32(b, 0); }
    while (p < total) {
        size_t e = p;
        while (e < total && img[e] == 0) ++e;
        size_t f = e;
        while (f < total && img[f] != 0 && (!redundant || f - e < 5)) ++f;
        put32(b, (uint32_t)(e - p)); put32(b, (uint32_t)(f - e));
        for (size_t i = e; i < f; ++i) { b.push_back((uint8_t)img[i]); b.push_back((uint8_t)(img[i] >> 8)); }
        p = f;
    }
    if (redundant) for (int i = 0; i < 7; ++i) b.push_back(0xff);      // trailing bytes are ignored (read_depth stops at w*h pixels)
    return b;
}
extern "C" int dh_biwi_decode_depth(const uint8_t *buf, size_t len, uint16_t *out, size_t cap_px, uint32_t *w, uint32_t *h);
extern "C" int dh_biwi_parse_cal(const char *text, size_t len, float K[9]);
extern "C" int dh_biwi_parse_pose(const uint8_t *buf, size_t len, const float K[9], float pos3d[3], float pos2d[2], float rot[3]);

static void test_rle() {
    const int iters = getenv("HOST_CHECK_LIGHT") ? 25 : 60;
    for (int it = 0; it < iters; ++it) {
        const uint32_t W = 1 + rnd(70), H = 1 + rnd(50);
        const int n = 1 + (int)rnd(9);
        std::vector<std::vector<uint16_t>> imgs(n);
        std::vector<std::vector<uint8_t>> pay(n);
        std::vector<const uint8_t *> bufs(n);
        std::vector<size_t> lens(n);
        for (int i = 0; i < n; ++i) {
            imgs[i].assign((size_t)W * H, 0);
            const uint32_t mode = rnd(5);
            for (auto &v : imgs[i]) v = mode == 0 ? 0 : mode == 1 ? (uint16_t)(1 + rnd(65535)) : (rnd(mode == 2 ? 2 : 6) ? 0 : (uint16_t)(1 + rnd(65535)));
            pay[i] = encode(imgs[i], W, H, rnd(3) == 0);
            // exact-size heap copies: a read past the payload is an ASan error
            uint8_t *q = (uint8_t *)malloc(pay[i].size());
            memcpy(q, pay[i].data(), pay[i].size());
            bufs[i] = q; lens[i] = pay[i].size();
        }
        for (int threads : {1, 4}) {
            RlePlan plan;
            CHECK(dh_rle_plan_(bufs.data(), lens.data(), n, threads, plan) == DH_OK);
            CHECK(plan.W == W && plan.H == H && plan.blob_off.size() == (size_t)n + 1 && plan.run_begin.size() == (size_t)n + 1);
            std::vector<uint8_t> blob(plan.blob_off[n]);
            std::vector<DhRun> runs(plan.nruns);
            dh_rle_pack_(bufs.data(), lens.data(), n, threads, plan, blob.data(), runs.data());
            // decode from (blob, runs) exactly as k_rle_decode does and compare with the images
            std::vector<uint16_t> out((size_t)n * W * H, 0);
            const uint16_t *b16 = (const uint16_t *)blob.data();
            for (int i = 0; i < n; ++i) {
                CHECK(plan.blob_off[i] % 16 == 0);
                for (uint32_t r = plan.run_begin[i]; r < plan.run_begin[i + 1]; ++r) {
                    const uint32_t nf = (uint32_t)b16[runs[r].src - 2] | ((uint32_t)b16[runs[r].src - 1] << 16);
                    CHECK(nf > 0 && (size_t)runs[r].dst + nf <= (size_t)(i + 1) * W * H && runs[r].dst >= (size_t)i * W * H);
                    for (uint32_t k = 0; k < nf; ++k) out[runs[r].dst + k] = b16[runs[r].src + k];
                }
                CHECK(memcmp(out.data() + (size_t)i * W * H, imgs[i].data(), (size_t)W * H * 2) == 0);
                // the host decoder of the C ABI agrees
                std::vector<uint16_t> hd((size_t)W * H, 0x5a5a);
                uint32_t w2 = 0, h2 = 0;
                CHECK(dh_biwi_decode_depth(bufs[i], lens[i], hd.data(), hd.size(), &w2, &h2) == DH_OK && w2 == W && h2 == H);
                CHECK(memcmp(hd.data(), imgs[i].data(), hd.size() * 2) == 0);
            }
        }
        // every truncation of one payload is refused by both, and nothing is read past its end
        {
            const int v = (int)rnd(n);
            std::vector<uint16_t> hd((size_t)W * H);
            uint32_t w2, h2;
            const size_t need = encode(imgs[v], W, H, false).size();
            uint8_t *full = (uint8_t *)malloc(need);
            { auto e = encode(imgs[v], W, H, false); memcpy(full, e.data(), need); }
            for (size_t cut = 0; cut < need; cut += 1 + rnd(3)) {
                uint8_t *q = (uint8_t *)malloc(cut ? cut : 1);
                memcpy(q, full, cut);
                const uint8_t *save = bufs[v]; const size_t slen = lens[v];
                bufs[v] = q; lens[v] = cut;
                RlePlan plan;
                int rc = dh_rle_plan_(bufs.data(), lens.data(), n, 2, plan);
                CHECK(rc == DH_EINVAL);
                CHECK(dh_biwi_decode_depth(q, cut, hd.data(), hd.size(), &w2, &h2) == DH_EINVAL);
                bufs[v] = save; lens[v] = slen;
                free(q);
            }
            free(full);
        }
        // runs that overrun the image, other sizes, NULL payloads
        {
            std::vector<uint8_t> b; put32(b, W); put32(b, H); put32(b, W * H + 1); put32(b, 0);
            const uint8_t *save = bufs[0]; const size_t slen = lens[0];
            bufs[0] = b.data(); lens[0] = b.size();
            RlePlan plan;
            CHECK(dh_rle_plan_(bufs.data(), lens.data(), n, 1, plan) == DH_EINVAL);
            std::vector<uint8_t> c; put32(c, W); put32(c, H); put32(c, W * H - 1); put32(c, 2); put32(c, 0x00020001u);
            bufs[0] = c.data(); lens[0] = c.size();
            CHECK(dh_rle_plan_(bufs.data(), lens.data(), n, 1, plan) == DH_EINVAL);
            if (n > 1) {
                std::vector<uint8_t> d2; put32(d2, W + 1); put32(d2, H); put32(d2, (W + 1) * H); put32(d2, 0);
                bufs[0] = save; lens[0] = slen;
                const uint8_t *s1 = bufs[1]; const size_t l1 = lens[1];
                bufs[1] = d2.data(); lens[1] = d2.size();
                CHECK(dh_rle_plan_(bufs.data(), lens.data(), n, 1, plan) == DH_EINVAL && strstr(dh_err_get_(), "frame 1"));
                bufs[1] = nullptr;
                CHECK(dh_rle_plan_(bufs.data(), lens.data(), n, 1, plan) == DH_EINVAL);
                bufs[1] = s1; lens[1] = l1;
            }
            bufs[0] = save; lens[0] = slen;
        }
        for (int i = 0; i < n; ++i) free((void *)bufs[i]);
    }
    RlePlan plan;
    CHECK(dh_rle_plan_(nullptr, nullptr, 0, 1, plan) == DH_EINVAL);
}

static void test_biwi_text() {
    const char *cal = "575.816 0 320\n0 575.816 240\n0 0 1\n\n0 0 0 0\n";
    float K[9];
    CHECK(dh_biwi_parse_cal(cal, strlen(cal), K) == DH_OK && K[0] == 575.816f && K[2] == 320.0f && K[8] == 1.0f);
    CHECK(dh_biwi_parse_cal("1 2\n3 4 5\n6 7 8\n", 16, K) == DH_EINVAL);
    CHECK(dh_biwi_parse_cal("1 2 3 4\n", 8, K) == DH_EINVAL);
    CHECK(dh_biwi_parse_cal("1 2 3", 5, K) == DH_EINVAL);                       // fewer than three lines
    CHECK(dh_biwi_parse_cal("1.5.5 2 3\n4 5 6\n7 8 9\n", 22, K) == DH_EINVAL);   // regex matches "1.5.5", f32::from_str refuses it
    uint8_t pose[24];
    const float v[6] = {10.f, -20.f, 900.f, 1.f, 2.f, 3.f};
    memcpy(pose, v, 24);
    float p3[3], p2[2], r[3];
    const float Kd[9] = {560, 0, 320, 0, 560, 240, 0, 0, 1};
    CHECK(dh_biwi_parse_pose(pose, 24, Kd, p3, p2, r) == DH_OK && p3[2] == 900.f && r[1] == 2.f);
    CHECK(dh_biwi_parse_pose(pose, 23, Kd, p3, p2, r) == DH_EINVAL);
}

static void test_tables_and_knobs() {
    std::vector<float> k;
    dh_build_kernel_table_(8.0f, k);
    CHECK(k.size() == 8000 && k[(10 * 20 + 10) * 20 + 10] == 1.0f && k[0] > 0.0f && k[0] < 1e-7f);
    // the table by squared distance (what k_cluster keeps in LDS) holds the same bits as every cell of the 20^3 kernel
    for (float sigma : {8.0f, 0.37f, 1234.5f}) {
        std::vector<float> full, r2;
        dh_build_kernel_table_(sigma, full);
        dh_build_kernel_r2_(sigma, r2, 304);
        CHECK(r2.size() == 304);
        for (int x = 0; x < 20; ++x) for (int y = 0; y < 20; ++y) for (int z = 0; z < 20; ++z) {
            const int n = (x - 10) * (x - 10) + (y - 10) * (y - 10) + (z - 10) * (z - 10);
            CHECK(n <= 300 && memcmp(&full[((size_t)x * 20 + y) * 20 + z], &r2[(size_t)n], 4) == 0);
        }
    }
    // k_boxsum's bands: whole mask blocks, every row covered once, and the choices measured on MI355X (profiles/r03_experiments.md)
    {
        int oh = 0;
        CHECK(dh_box_bands_(256, 3, 457, 32, 24, 768, &oh) == 4 && oh == 128);      // the headline batch: 768 workgroups, one round
        CHECK(dh_box_bands_(512, 3, 457, 32, 24, 768, &oh) == 4 && oh == 128);      // (the wave-count rule took 2: 0.28 instead of 0.19 ms)
        CHECK(dh_box_bands_(320, 3, 457, 32, 24, 768, &oh) == 5 && oh == 96);
        CHECK(dh_box_bands_(128, 3, 457, 32, 24, 768, &oh) == 8 && oh == 64);
        CHECK(dh_box_bands_(1, 1, 217, 8, 24, 768, &oh) == 28 && oh == 8);           // one 320 x 240 frame: the shortest bands
        for (int n : {1, 2, 7, 64, 100, 256, 300, 512, 4096})
            for (int rows : {1, 5, 31, 32, 33, 217, 457, 1000})
                for (int blk : {8, 16, 32})
                    for (int parts : {1, 3, 5}) {
                        const int b = dh_box_bands_(n, parts, rows, blk, 24, 768, &oh);
                        CHECK(b >= 1 && oh >= blk && oh % blk == 0 && b == (rows + oh - 1) / oh && (long)b * oh >= rows && (long)(b - 1) * oh < rows);
                    }
        CHECK(dh_box_bands_(0, 0, 0, 0, 0, 0, &oh) == 1 && oh == 1);                 // degenerate arguments are clamped
    }
    CHECK(dh_blur_taps_(8.0f, k) == DH_OK && k.size() == 33 && k[16] > k[15] && k[0] == k[32]);
    CHECK(dh_blur_taps_(0.0f, k) == DH_EINVAL && dh_blur_taps_(5000.0f, k) == DH_ESIZE);
    const float m[9] = {560, 0, 320, 0, 560, 240, 0, 0, 1};
    float o[9];
    dh_mat3_inv_f32_(m, o);
    CHECK(o[0] == 1.0f / 560.0f && o[8] == 1.0f);
    setenv("DH_TILE", "12,7", 1); setenv("DH_CHUNKS", "99", 1); setenv("DH_STAGE_CHUNK", "-5", 1); setenv("DH_HOST_THREADS", "1000", 1);
    setenv("DH_TOP_LEVELS", "77", 1); setenv("DH_NO_ABSORB", "", 1);
    Knobs kn = dh_read_knobs_();
    CHECK(kn.tile_x == 12 && kn.tile_y == 7 && kn.chunks == 8 && kn.stage_chunk == 1 && kn.host_threads == 64 && kn.top_levels == 8 && kn.no_absorb);
    setenv("DH_TILE", "12", 1);
    kn = dh_read_knobs_();
    CHECK(kn.tile_x == 0 && kn.tile_y == 0);                                     // malformed: ignored
    setenv("DH_TILE", "0,-3", 1);
    kn = dh_read_knobs_();
    CHECK(kn.tile_x == 0 && kn.tile_y == 0);
    for (const char *e : {"DH_TILE", "DH_CHUNKS", "DH_STAGE_CHUNK", "DH_HOST_THREADS", "DH_TOP_LEVELS", "DH_NO_ABSORB"}) unsetenv(e);
    kn = dh_read_knobs_();
    CHECK(kn.max_resident == 512 && kn.stage_chunk == 64 && kn.top_levels == -1 && !kn.no_absorb);
}

static void test_guard_and_threads() {
    CHECK(dh_guard_("x", []() -> int { throw std::bad_alloc(); }) == DH_ENOMEM && strstr(dh_err_get_(), "out of host memory"));
    CHECK(dh_guard_("x", []() -> int { throw std::runtime_error("boom"); }) == DH_EINVAL && strstr(dh_err_get_(), "boom"));
    CHECK(dh_guard_("x", []() -> int { throw 7; }) == DH_EINVAL);
    CHECK(dh_guard_("x", []() -> int { return 42; }) == 42);
    // parallel_for: every index exactly once, from many caller threads at once; error slots are per thread
    std::vector<std::thread> ths;
    std::atomic<int> bad{0};
    for (int t = 0; t < 6; ++t)
        ths.emplace_back([t, &bad] {
            for (int rep = 0; rep < 20; ++rep) {
                const int n = 1 + (t * 37 + rep * 11) % 200;
                std::vector<std::atomic<int>> hits(n);
                for (auto &h : hits) h = 0;
                dh_parallel_for_(n, 1 + (rep % 8), [&](int i) { hits[i].fetch_add(1); });
                for (int i = 0; i < n; ++i) if (hits[i].load() != 1) bad.fetch_add(1);
                dh_fail_(DH_EINVAL, "thread %d rep %d", t, rep);
                char want[64];
                snprintf(want, sizeof want, "thread %d rep %d", t, rep);
                if (strcmp(dh_err_get_(), want) != 0) bad.fetch_add(1);
            }
        });
    for (auto &t : ths) t.join();
    CHECK(bad.load() == 0);
    dh_parallel_for_(0, 4, [](int) { abort(); });
}

int main() {
    test_forest();
    test_geometry();
    test_chunk_plan();
    test_rle();
    test_biwi_text();
    test_tables_and_knobs();
    test_guard_and_threads();
    if (g_fail) { fprintf(stderr, "host_check: %d check(s) failed\n", g_fail); return 1; }
    printf("host_check ok\n");
    return 0;
}
