// Drop-in check of the C++ surface (include/GPisMap3.h, include/GPisMap.h): this file uses the map classes the
// way the reference's mex gateways do (mexGPisMap3.cpp:49-166, mexGPisMap.cpp:40-131) and is compiled with the
// gateways' own flags (-std=c++11 -pthread -fPIC) against the library instead of the reference sources.
// Output: one line per check, parsed by tests/test_host.py (build) and tests/test_gpu_dropin.py (run).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "GPisMap.h"
#include "GPisMap3.h"

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
    GPisMap3* gpm = new GPisMap3(GPisMap3Param(), c);
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
