// dh_biwi.cpp -- BIWI Kinect Head Pose Database file formats:
// host-side decoders feeding the frame batches of dh_predict_batch.  Restates the parsers of
// /root/reference/src/db_reader/biwi.rs: read_depth (:81-103), read_cal (:27-60), read_gt (:63-77).
// Byte / integer work and three f32 operations; no device code: plain C++ (builds with g++ for the sanitizer tests, tests/host/).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "dh_host.h"

static inline uint32_t rd_u32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

// read_depth (biwi.rs:81-103): u32 width, u32 height, then runs { u32 n_empty; u32 n_full; u16 depth[n_full] }
// (little endian) until width*height pixels are produced; empty pixels are 0.
static int decode_depth(const uint8_t *buf, size_t len, uint16_t *out, size_t cap_px, uint32_t *w, uint32_t *h) {
    if (!buf || !w || !h) return dh_fail_(DH_EINVAL, "dh_biwi_decode_depth: NULL argument");
    if (len < 8) return dh_fail_(DH_EINVAL, "depth file truncated in the header");          // read_u32 fails (:83-84)
    const uint32_t W = rd_u32(buf), H = rd_u32(buf + 4);
    *w = W; *h = H;
    const size_t total = (size_t)((uint32_t)(W * H));   // `(width * height) as usize` is a u32 product (:85-88)
    if ((uint64_t)W * H != total) return dh_fail_(DH_ESIZE, "depth image %ux%u overflows u32", W, H);
    if (!out) return DH_OK;                              // size query
    if (cap_px < total) return dh_fail_(DH_EINVAL, "output holds %zu pixels, image has %zu", cap_px, total);
    size_t p = 0, pos = 8;
    while (p < total) {
        if (len - pos < 4) return dh_fail_(DH_EINVAL, "depth file truncated at byte %zu", pos);
        const uint32_t n_empty = rd_u32(buf + pos); pos += 4;                                 // :90
        if ((size_t)n_empty > total - p) return dh_fail_(DH_EINVAL, "run of %u empty pixels overruns the image (reference panics, :92)", n_empty);
        memset(out + p, 0, (size_t)n_empty * 2);
        p += n_empty;
        if (len - pos < 4) return dh_fail_(DH_EINVAL, "depth file truncated at byte %zu", pos);
        const uint32_t n_full = rd_u32(buf + pos); pos += 4;                                  // :94
        if ((size_t)n_full > total - p) return dh_fail_(DH_EINVAL, "run of %u pixels overruns the image (reference panics, :97)", n_full);
        if ((len - pos) / 2 < n_full) return dh_fail_(DH_EINVAL, "depth file truncated inside a run at byte %zu", pos);
        for (uint32_t i = 0; i < n_full; ++i) out[p + i] = (uint16_t)(buf[pos + 2 * i] | buf[pos + 2 * i + 1] << 8);   // :96
        pos += (size_t)n_full * 2;
        p += n_full;
    }
    return DH_OK;
}

// read_cal (biwi.rs:27-60): the first three lines must each hold exactly three matches of the
// regex (\d+[\.\d+]*) -- a digit followed by any run of digits, '.' and '+'; signs and exponents are
// not part of a match -- each parsed by f32::from_str.
static int parse_cal(const char *text, size_t len, float K[9]) {
    if (!text || !K) return dh_fail_(DH_EINVAL, "dh_biwi_parse_cal: NULL argument");
    size_t pos = 0;
    for (int j = 0; j < 3; ++j) {
        size_t eol = pos;
        while (eol < len && text[eol] != '\n') ++eol;                                       // read_line (:37)
        int found = 0;
        for (size_t i = pos; i < eol;) {
            if (text[i] < '0' || text[i] > '9') { ++i; continue; }
            size_t e = i + 1;
            while (e < eol && ((text[e] >= '0' && text[e] <= '9') || text[e] == '.' || text[e] == '+')) ++e;
            if (found == 3) return dh_fail_(DH_EINVAL, "Unsupported Calibration-File: more than 3 numbers on line %d", j + 1);   // :48-51
            // f32::from_str on the token: digits [ '.' digits* ]
            std::string tok(text + i, e - i);
            size_t k = 0;
            while (k < tok.size() && tok[k] >= '0' && tok[k] <= '9') ++k;
            if (k < tok.size() && tok[k] == '.') { ++k; while (k < tok.size() && tok[k] >= '0' && tok[k] <= '9') ++k; }
            if (k != tok.size()) return dh_fail_(DH_EINVAL, "calibration token '%s' is not a float (f32::from_str fails, :46)", tok.c_str());
            K[j * 3 + found] = strtof(tok.c_str(), nullptr);
            ++found;
            i = e;
        }
        if (found != 3) return dh_fail_(DH_EINVAL, "Unsupported Calibration-File: line %d has %d numbers", j + 1, found);   // :54-56
        pos = eol < len ? eol + 1 : eol;
    }
    return DH_OK;
}

// read_gt (biwi.rs:63-77): six little-endian f32: position (mm), rotation (degrees); the 2-D
// position is IntrinsicMatrix::space_to_img_coord of the 3-D one (types.rs:424-428).
static int parse_pose(const uint8_t *buf, size_t len, const float K[9], float pos3d[3], float pos2d[2], float rot[3]) {
    if (!buf || !K || !pos3d || !pos2d || !rot) return dh_fail_(DH_EINVAL, "dh_biwi_parse_pose: NULL argument");
    if (len < 24) return dh_fail_(DH_EINVAL, "pose file truncated (%zu of 24 bytes)", len);
    float v[6];
    for (int i = 0; i < 6; ++i) { uint32_t u = rd_u32(buf + 4 * i); memcpy(&v[i], &u, 4); }
    for (int i = 0; i < 3; ++i) { pos3d[i] = v[i]; rot[i] = v[3 + i]; }
    float r[3];
    for (int j = 0; j < 3; ++j) {                       // Mat3 * Vec3 (meancov_estimation.rs:201-216); this TU is built -ffp-contract=off
        float t = v[0] * K[j * 3 + 0];
        t = t + v[1] * K[j * 3 + 1];
        t = t + v[2] * K[j * 3 + 2];
        r[j] = t;
    }
    pos2d[0] = r[0] / r[2];
    pos2d[1] = r[1] / r[2];
    return DH_OK;
}

// ---- the C ABI (nothing throws across it: parse_cal builds std::string tokens)
extern "C" int dh_biwi_decode_depth(const uint8_t *buf, size_t len, uint16_t *out, size_t cap_px, uint32_t *w, uint32_t *h) {
    return dh_guard_("dh_biwi_decode_depth", [&] { return decode_depth(buf, len, out, cap_px, w, h); });
}
extern "C" int dh_biwi_parse_cal(const char *text, size_t len, float K[9]) {
    return dh_guard_("dh_biwi_parse_cal", [&] { return parse_cal(text, len, K); });
}
extern "C" int dh_biwi_parse_pose(const uint8_t *buf, size_t len, const float K[9], float pos3d[3], float pos2d[2], float rot[3]) {
    return dh_guard_("dh_biwi_parse_pose", [&] { return parse_pose(buf, len, K, pos3d, pos2d, rot); });
}
