/* gpismap_amd -- MI355X-native drop-in for the reference's 2-D map class.
 *
 * Mirrors the public surface of reference cpp/include/GPisMap.h:
 *   GPisMapParam  :29-67   (same members, same defaults)
 *   class GPisMap :69-119  (same public methods and argument meaning)
 */
#ifndef GPISMAP_AMD_GPISMAP_H_
#define GPISMAP_AMD_GPISMAP_H_

/* Standard headers the reference's own header exposed to its includers (reference cpp/include/GPisMap.h:23-26 -> ObsGP.h:24-28, OnGPIS.h:24-28, quadtree.h, strct.h
 * and, through Eigen, <cstring>/<cmath>/<algorithm>): the mex gateways rely on them
 * (mex/mexGPisMap3.cpp:154 calls memcpy without including <cstring>). */
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <unordered_set>
#include <vector>

typedef struct GPisMapParam_ {
    float delx;          // numerical step delta (surface normal sampling)
    float fbias;         // constant map bias (mean of the GP)
    float sensor_offset[2];
    float angle_obs_limit[2];
    float obs_var_thre;  // ObsGP variance above which a prediction is not trusted
    float min_position_noise;
    float min_grad_noise;
    float map_scale_param;
    float map_noise_param;

    GPisMapParam_() {
        delx = 1e-2;
        fbias = 0.2;
        obs_var_thre = 0.1;
        sensor_offset[0] = 0.08;
        sensor_offset[1] = 0.0;
        angle_obs_limit[0] = (-135.0 * 3.14159265358979323846 / 180.0);
        angle_obs_limit[1] = (135.0 * 3.14159265358979323846 / 180.0);
        min_position_noise = 1e-2;
        min_grad_noise = 1e-2;
        map_scale_param = 1.2;
        map_noise_param = 1e-2;
    }
    /* The reference declares this copy constructor (cpp/include/GPisMap.h:56-66, non-const reference); it decides the calling
     * convention of GPisMap(GPisMapParam) (hidden pointer instead of 44 bytes on the stack), see include/GPisMap3.h.  The
     * reference's body leaves angle_obs_limit of the copy uninitialised; here it is copied as well:
     * a caller that sets the limits gets them, a caller that never touches them sees the same defaults either way. */
    GPisMapParam_(GPisMapParam_& par)
        : delx(par.delx), fbias(par.fbias), obs_var_thre(par.obs_var_thre),
          min_position_noise(par.min_position_noise), min_grad_noise(par.min_grad_noise),
          map_scale_param(par.map_scale_param), map_noise_param(par.map_noise_param) {
        sensor_offset[0] = par.sensor_offset[0];
        sensor_offset[1] = par.sensor_offset[1];
        angle_obs_limit[0] = par.angle_obs_limit[0];
        angle_obs_limit[1] = par.angle_obs_limit[1];
    }
} GPisMapParam;

class GPisMap {
public:
    GPisMap();
    GPisMap(GPisMapParam par);
    ~GPisMap();
    void reset();

    void update(float* datax, float* dataf, int N, std::vector<float>& pose);
    bool test(float* x, int dim, int leng, float* res);
    int getMapDimension() { return 2; }

    /* Extensions (not in the reference) */
    bool testDevice(const float* d_x, int leng, float* d_res, void* hip_stream);
    void getAllNodes(std::vector<float>& out7);  /* pos2 grad2 val sigx sigg, tree order */
    struct Impl;
    Impl* impl() { return p_; }

private:
    GPisMap(const GPisMap&);
    GPisMap& operator=(const GPisMap&);
    Impl* p_;
};

#endif
