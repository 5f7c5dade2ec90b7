// Minimal C++ host of the C ABI (include/depthhead_hip.h): what a compiled caller -- the Rust shim of
// INTEGRATION.md, or any C/C++ program -- does per frame.  No Python, no PyTorch.
//
//   g++ -std=c++17 -Iinclude examples/predict_frame.cpp -Ldepthhead_amd -ldepthhead_hip \
//       -Wl,-rpath,$PWD/depthhead_amd -o predict_frame && ./predict_frame
//
// It builds a two-tree forest by hand (the layout of dh_forest_desc), a synthetic 640x480 depth frame with a
// head-sized blob, calls dh_predict_batch the way HoughPrediction::predict_parameter_parallel would
// (prediction.rs:397-409: one frame, default Kinect intrinsic, no guesses) and prints the pose.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "depthhead_hip.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != DH_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dh_last_error()); return 1; } \
    } while (0)

int main() {
    // ---- forest: two single-split trees; every leaf holds three offset votes and three rotation votes
    std::vector<dh_node> nodes(2);
    nodes[0] = {{4, 4, 28, 28}, {40, 40, 64, 64}, 0.0, ~0, ~1};        // avg(r1) - avg(r2) > 0 ? leaf 1 : leaf 0
    nodes[1] = {{10, 30, 34, 54}, {44, 6, 68, 30}, 10.0, ~2, ~3};
    std::vector<int32_t> roots = {0, 1};
    std::vector<double> prob = {1.0, 0.9, 0.95, 1.0};
    std::vector<uint32_t> begin = {0, 3, 6, 9, 12};
    std::vector<float> offsets;
    std::vector<double> rotations;
    for (int leaf = 0; leaf < 4; ++leaf)
        for (int k = 0; k < 3; ++k) {
            offsets.insert(offsets.end(), {10.0f * leaf + k, -5.0f * leaf - k, 20.0f + 2.0f * k});
            rotations.insert(rotations.end(), {5.0 * leaf + k, -3.0 * leaf, 1.0 * k});
        }
    dh_forest_desc desc{};
    desc.n_trees = 2; desc.roots = roots.data();
    desc.n_nodes = 2; desc.nodes = nodes.data();
    desc.n_leaves = 4; desc.leaf_prob = prob.data();
    desc.off_begin = begin.data(); desc.rot_begin = begin.data();
    desc.offsets = offsets.data(); desc.rotations = rotations.data();
    dh_forest *forest = nullptr;
    CHECK(dh_forest_create(&desc, &forest));

    // ---- predictor: the scalars HoughPrediction serialises (prediction.rs:239-256); the trainer's step width
    dh_params params{10, 80, 80, 8.0f, 20};
    dh_predictor *pred = nullptr;
    CHECK(dh_predictor_create(forest, &params, /*device=*/0, &pred));

    // ---- one depth frame: background 0, a 95 mm sphere at 900 mm in the image centre
    const int W = 640, H = 480;
    std::vector<uint16_t> frame((size_t)W * H, 0);
    const float fx = 560.0f, z0 = 900.0f, R = 95.0f;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float dx = (x - W / 2) * z0 / fx, dy = (y - H / 2) * z0 / fx, d2 = dx * dx + dy * dy;
            if (d2 < R * R) frame[(size_t)y * W + x] = (uint16_t)std::lround(z0 - std::sqrt(R * R - d2));
        }
    const float K[9] = {fx, 0, W / 2.0f, 0, fx, H / 2.0f, 0, 0, 1};   // IntrinsicMatrix::default_kinect_intrinsic shape (types.rs:418-420)

    dh_pose pose{};
    CHECK(dh_predict_batch(pred, frame.data(), 1, W, H, K, nullptr, nullptr, nullptr, &pose));
    std::printf("mid_point = (%.0f, %.0f, %.0f) mm   rotation = (%.6f, %.6f, %.6f) rad\n", pose.mid_point[0], pose.mid_point[1],
                pose.mid_point[2], pose.rotation[0], pose.rotation[1], pose.rotation[2]);

    CHECK(dh_predictor_destroy(pred));
    CHECK(dh_forest_destroy(forest));
    return 0;
}
