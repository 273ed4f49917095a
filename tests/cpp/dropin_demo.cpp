// Drop-in check of the C++ surface (include/GPisMap3.h, include/GPisMap.h): this file uses the map classes the
// way the reference's mex gateways do (mexGPisMap3.cpp:49-166, mexGPisMap.cpp:40-131) and is compiled with the
// gateways' own flags (-std=c++11 -pthread -fPIC) against the library instead of the reference sources.
// Output: one line per check, parsed by tests/test_host.py (build) and tests/test_gpu_dropin.py (run).
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>
#include "GPisMap.h"
#include "GPisMap3.h"

// ---- ABI of the by-value public structs --------------------------------------------------------------------
// The reference gives both parameter structs a user-provided copy constructor taking a NON-CONST reference
// (cpp/include/GPisMap3.h:71-80, cpp/include/GPisMap.h:56-66).  That makes them non-trivial for the purposes of calls: the
// constructors GPisMap3(GPisMap3Param), GPisMap3(GPisMap3Param, camParam) and GPisMap(GPisMapParam) take `par` through a hidden
// pointer to a caller-made copy.  A header without the copy constructors would produce the same mangled names with a different
// calling convention (the struct in registers / on the stack), so these properties are part of the drop-in surface.
static_assert(!std::is_trivially_copy_constructible<GPisMap3Param>::value, "reference cpp/include/GPisMap3.h:71-80");
static_assert(!std::is_trivially_copyable<GPisMap3Param>::value, "reference cpp/include/GPisMap3.h:71-80");
static_assert(!std::is_constructible<GPisMap3Param, const GPisMap3Param&>::value, "the reference's copy constructor binds lvalues only");
static_assert(!std::is_trivially_copy_constructible<GPisMapParam>::value, "reference cpp/include/GPisMap.h:56-66");
static_assert(!std::is_trivially_copyable<GPisMapParam>::value, "reference cpp/include/GPisMap.h:56-66");
// camParam has no user-provided copy constructor in the reference (cpp/include/GPisMap3.h:29-46): trivially copyable, by value
static_assert(std::is_trivially_copy_constructible<camParam>::value, "reference cpp/include/GPisMap3.h:29-46");
// field order / offsets / sizes: reference cpp/include/GPisMap3.h:30-35, :49-59 and cpp/include/GPisMap.h:30-42.  offsetof on
// these (standard-layout, non-POD because of the constructors) types is conditionally supported; g++ and clang accept it.
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Winvalid-offsetof"
static_assert(std::is_standard_layout<camParam>::value && std::is_standard_layout<GPisMap3Param>::value &&
              std::is_standard_layout<GPisMapParam>::value, "plain structs");
static_assert(sizeof(camParam) == 24 && offsetof(camParam, fx) == 0 && offsetof(camParam, fy) == 4 && offsetof(camParam, cx) == 8 &&
              offsetof(camParam, cy) == 12 && offsetof(camParam, width) == 16 && offsetof(camParam, height) == 20, "camParam layout");
static_assert(sizeof(GPisMap3Param) == 32 && offsetof(GPisMap3Param, delx) == 0 && offsetof(GPisMap3Param, fbias) == 4 &&
              offsetof(GPisMap3Param, obs_var_thre) == 8 && offsetof(GPisMap3Param, obs_skip) == 12 &&
              offsetof(GPisMap3Param, min_position_noise) == 16 && offsetof(GPisMap3Param, min_grad_noise) == 20 &&
              offsetof(GPisMap3Param, map_scale_param) == 24 && offsetof(GPisMap3Param, map_noise_param) == 28, "GPisMap3Param layout");
static_assert(sizeof(GPisMapParam) == 44 && offsetof(GPisMapParam, delx) == 0 && offsetof(GPisMapParam, fbias) == 4 &&
              offsetof(GPisMapParam, sensor_offset) == 8 && offsetof(GPisMapParam, angle_obs_limit) == 16 &&
              offsetof(GPisMapParam, obs_var_thre) == 24 && offsetof(GPisMapParam, min_position_noise) == 28 &&
              offsetof(GPisMapParam, min_grad_noise) == 32 && offsetof(GPisMapParam, map_scale_param) == 36 &&
              offsetof(GPisMapParam, map_noise_param) == 40, "GPisMapParam layout");
#pragma GCC diagnostic pop

static unsigned checksum(const std::vector<float>& v) {
    unsigned c = 2166136261u;
    for (size_t i = 0; i < v.size(); ++i) { unsigned u; std::memcpy(&u, &v[i], 4); c = (c ^ u) * 16777619u; }
    return c;
}

static bool read_bin(const char* dir, const char* name, std::vector<float>& v) {
    if (!dir) return false;
    char path[1024];
    std::snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    size_t n = std::fread(v.data(), sizeof(float), v.size(), f);
    std::fclose(f);
    return n == v.size();
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : 0;   // optional: inputs written by the test (depth0/1.bin, x.bin, th/rg/x2.bin)
    // ---- 3-D: 'setCamera' + 'update' x2 + 'test' + 'getAllPoints' + 'reset' ----
    camParam c(568.0f, 568.0f, 310.0f, 224.0f, 640, 480);
    GPisMap3Param p3;                          // an lvalue, as mexGPisMap3.cpp:133-140 passes it (the copy constructor binds nothing else)
    GPisMap3* gpm = new GPisMap3(p3, c);
    const int W = 640, H = 480;
    std::vector<float> depth((size_t)W * H);
    std::vector<float> pose = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::vector<float> x;
    const int G = 24;
    for (int k = 0; k < G; ++k) for (int j = 0; j < G; ++j) for (int i = 0; i < G; ++i) {
        x.push_back((float)(-0.60 + 1.20 * i / (G - 1)));
        x.push_back((float)(-0.45 + 0.90 * j / (G - 1)));
        x.push_back((float)(0.85 + 0.30 * k / (G - 1)));
    }
    std::vector<float> res((size_t)8 * G * G * G, 0.f);
    bool before = gpm->test(x.data(), 3, G * G * G, res.data());      // no update yet: must be refused
    std::printf("test_before_update %d\n", before ? 1 : 0);
    for (int f = 0; f < 2; ++f) {
        for (int col = 0; col < W; ++col)
            for (int row = 0; row < H; ++row) {
                double u = (col - 310.0) / 568.0, v = (row - 224.0) / 568.0;
                depth[(size_t)col * H + row] = (float)(1.0 + 0.05 * std::sin(6.0 * (u + 0.01 * f)) * std::cos(5.0 * v));
            }
        read_bin(dir, f == 0 ? "depth0.bin" : "depth1.bin", depth);
        gpm->update(depth.data(), W * H, pose);
    }
    read_bin(dir, "x.bin", x);
    bool ok = gpm->test(x.data(), 3, G * G * G, res.data());
    std::vector<float> pts;
    gpm->getAllPoints(pts);
    std::printf("map3 ok %d points %zu res_checksum %08x\n", ok ? 1 : 0, pts.size() / 3, checksum(res));
    bool bad = gpm->test(x.data(), 2, G * G * G, res.data());          // wrong dimension: false (GPisMap3.cpp:905)
    std::printf("test_wrong_dim %d\n", bad ? 1 : 0);
    gpm->reset();
    delete gpm;

    // ---- 2-D: 'update' + 'test' + 'reset' ----
    {   // the by-value constructor once (GPisMap.cpp:69), then the default one like mexGPisMap.cpp:43
        GPisMapParam p2;
        p2.angle_obs_limit[0] = -2.0f;
        GPisMap* gtmp = new GPisMap(p2);
        std::printf("map2_by_value %d\n", gtmp->getMapDimension());
        delete gtmp;
    }
    GPisMap* g2 = new GPisMap();
    std::printf("map_dimension %d\n", g2->getMapDimension());
    const int NB = 270;
    std::vector<float> th(NB), rg(NB);
    for (int i = 0; i < NB; ++i) {
        th[i] = (float)((-135.0 + i) * M_PI / 180.0);
        rg[i] = (float)(4.0 + 0.5 * std::sin(0.07 * i));
    }
    std::vector<float> pose2 = {0.f, 0.f, 1.f, 0.f, 0.f, 1.f};
    read_bin(dir, "th.bin", th); read_bin(dir, "rg.bin", rg);
    g2->update(th.data(), rg.data(), NB, pose2);
    std::vector<float> x2;
    for (int j = 0; j < 40; ++j) for (int i = 0; i < 40; ++i) { x2.push_back(-5.f + 0.25f * i); x2.push_back(-5.f + 0.25f * j); }
    std::vector<float> res2((size_t)6 * 1600, 0.f);
    read_bin(dir, "x2.bin", x2);
    bool ok2 = g2->test(x2.data(), 2, 1600, res2.data());
    std::printf("map2 ok %d res_checksum %08x\n", ok2 ? 1 : 0, checksum(res2));
    g2->reset();
    delete g2;
    return 0;
}
