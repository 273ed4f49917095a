// C-ABI (include/gpismap_amd.h) over the C++ classes.  Nothing throws across it.
#include <cstring>
#include <new>
#include <vector>
#include "../../include/GPisMap.h"
#include "../../include/GPisMap3.h"
#include "../../include/gpismap_amd.h"
#include "map_query.h"
#include "obsgp.h"
#include "ongpis.h"

using namespace gpis;

// accessors implemented in gpismap3.cpp
void gpis3_impl_stats(GPisMap3* m, double* out, int n);
void gpis3_impl_profile(GPisMap3* m, int on);
void gpis2_impl_stats(GPisMap* m, double* out, int n);
int gpis3_impl_fail(GPisMap3* m);
int gpis2_impl_fail(GPisMap* m);
int gpis3_impl_update_fail(GPisMap3* m);
int gpis3_impl_sync(GPisMap3* m);
void gpis3_impl_set_pipeline(GPisMap3* m, int on);
void gpis3_impl_set_host_gather(GPisMap3* m, int on);
void gpis3_impl_set_keep_factors(GPisMap3* m, int on);
void gpis3_impl_set_shard_factors(GPisMap3* m, int mode);
int gpis3_impl_prepare_test(GPisMap3* m);
void gpis3_impl_set_lazy_inverse(GPisMap3* m, int on);
int gpis2_impl_update_fail(GPisMap* m);
int gpis3_impl_device(GPisMap3* m);
int gpis3_impl_num_devices(GPisMap3* m);
GPisMap3* gpis3_impl_create_on(const GPisMap3Param& par, const camParam& c, const int* devices, int n);
int gpis3_impl_set_shard(GPisMap3* m, int rank, int world);
int gpis3_impl_shard_info(GPisMap3* m, int* out, int n);
long long gpis3_impl_shard_bytes(GPisMap3* m, int owner);
int gpis3_impl_shard_pack(GPisMap3* m, void* d_buf, void* stream);
int gpis3_impl_shard_unpack(GPisMap3* m, int owner, const void* d_buf, void* stream);
int gpis3_impl_shard_finish(GPisMap3* m);
int gpis3_impl_set_frame_export(GPisMap3* g, int on);
long long gpis3_impl_frame_record(GPisMap3* g, void* buf, long long cap);
int gpis3_impl_train_deferred(GPisMap3* g);
int gpis3_impl_apply_frame(GPisMap3* g, const void* buf, long long bytes);
int gpis2_impl_device(GPisMap* m);
int gpis2_impl_sync(GPisMap* m);
void gpis2_impl_set_pipeline(GPisMap* m, int on);

namespace gpis { int selftest_ranged_arith(unsigned long long seed, int blocks, int per_thread, int mode, unsigned long long* mismatches); }
extern "C" {

int gpis_selftest_ranged_arith(unsigned long long seed, int blocks, int per_thread, int mode, unsigned long long* mismatches2) {
    try { return gpis::selftest_ranged_arith(seed, blocks, per_thread, mode, mismatches2); } catch (...) { return GPIS_ERR_STATE; }
}
unsigned long long gpis_pool_cache_trim(void) { try { return (unsigned long long)gpis::pool_cache_trim(); } catch (...) { return 0; } }
int gpis_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
const char* gpis_version(void) { return "gpismap_amd 0.2 (gfx950)"; }
int gpis_set_device(int device) {
    int n = gpis_device_count();
    if (device < 0 || device >= n) return GPIS_ERR_ARG;
    return hipSetDevice(device) == hipSuccess ? GPIS_OK : GPIS_ERR_HIP;
}
int gpis_get_device(void) {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) return GPIS_ERR_HIP;
    return d;
}

// ---- 3-D map ----------------------------------------------------------------
void* gpis3_create(const gpis_cam* cam) {
    try {
        GPisMap3Param p;
        if (cam) { camParam c(cam->fx, cam->fy, cam->cx, cam->cy, (float)cam->width, (float)cam->height); return new GPisMap3(p, c); }
        return new GPisMap3(p);
    } catch (...) { return nullptr; }
}
void* gpis3_create_multi(const gpis_cam* cam, const int* devices, int n) {
    if (!devices || n < 1) return nullptr;
    try {
        GPisMap3Param p;
        camParam c;
        if (cam) c = camParam(cam->fx, cam->fy, cam->cx, cam->cy, (float)cam->width, (float)cam->height);
        return gpis3_impl_create_on(p, c, devices, n);
    } catch (...) { return nullptr; }
}
int gpis3_num_devices(void* m) { if (!m) return GPIS_ERR_ARG; return gpis3_impl_num_devices((GPisMap3*)m); }
void gpis3_destroy(void* m) { delete (GPisMap3*)m; }
int gpis3_reset(void* m) { if (!m) return GPIS_ERR_ARG; ((GPisMap3*)m)->reset(); return GPIS_OK; }
int gpis3_set_camera(void* m, const gpis_cam* cam) {
    if (!m || !cam) return GPIS_ERR_ARG;
    camParam c(cam->fx, cam->fy, cam->cx, cam->cy, (float)cam->width, (float)cam->height);
    ((GPisMap3*)m)->resetCam(c);
    return GPIS_OK;
}
int gpis3_update(void* m, const float* depth, int n, const float* pose12) {
    if (!m || !depth || !pose12) return GPIS_ERR_ARG;
    if (gpis_device_count() < 1) return GPIS_ERR_HIP;
    try {
        std::vector<float> pose(pose12, pose12 + 12);
        ((GPisMap3*)m)->update(const_cast<float*>(depth), n, pose);
    } catch (...) { return GPIS_ERR_STATE; }
    return gpis3_impl_update_fail((GPisMap3*)m);   // 0 unless a device step inside failed (bad input is silent, as in the reference)
}
int gpis3_test(void* m, const float* x, int dim, int n, float* res) {
    if (!m) return GPIS_ERR_ARG;
    if (gpis_device_count() < 1) return GPIS_ERR_HIP;
    // false = the reference's own refusal (GPIS_ERR_ARG) unless the device path failed: that is reported as such
    try { if (((GPisMap3*)m)->test(const_cast<float*>(x), dim, n, res)) return GPIS_OK; int e = gpis3_impl_fail((GPisMap3*)m); return e ? e : GPIS_ERR_ARG; }
    catch (...) { return GPIS_ERR_STATE; }
}
int gpis3_test_device(void* m, const float* d_x, int n, float* d_res, void* stream) {
    if (!m) return GPIS_ERR_ARG;
    try { if (((GPisMap3*)m)->testDevice(d_x, n, d_res, stream)) return GPIS_OK; int e = gpis3_impl_fail((GPisMap3*)m); return e ? e : GPIS_ERR_ARG; }
    catch (...) { return GPIS_ERR_STATE; }
}
int gpis3_device(void* m) { if (!m) return GPIS_ERR_ARG; return gpis3_impl_device((GPisMap3*)m); }
int gpis3_set_shard(void* m, int rank, int world) {
    if (!m || world < 1 || rank < 0 || rank >= world) return GPIS_ERR_ARG;
    return gpis3_impl_set_shard((GPisMap3*)m, rank, world);
}
int gpis3_shard_info(void* m, int* out, int n) { if (!m || !out) return GPIS_ERR_ARG; return gpis3_impl_shard_info((GPisMap3*)m, out, n); }
long long gpis3_shard_bytes(void* m, int owner) { if (!m) return GPIS_ERR_ARG; return gpis3_impl_shard_bytes((GPisMap3*)m, owner); }
int gpis3_shard_pack(void* m, void* d_buf, void* stream) {
    if (!m) return GPIS_ERR_ARG;
    return gpis3_impl_shard_pack((GPisMap3*)m, d_buf, stream);
}
int gpis3_shard_unpack(void* m, int owner, const void* d_buf, void* stream) {
    if (!m) return GPIS_ERR_ARG;
    return gpis3_impl_shard_unpack((GPisMap3*)m, owner, d_buf, stream);
}
int gpis3_shard_finish(void* m) { if (!m) return GPIS_ERR_ARG; return gpis3_impl_shard_finish((GPisMap3*)m); }
int gpis3_set_frame_export(void* m, int on) { if (!m) return GPIS_ERR_ARG; return gpis3_impl_set_frame_export((GPisMap3*)m, on); }
long long gpis3_frame_record(void* m, void* buf, long long cap) { if (!m || cap < 0) return GPIS_ERR_ARG; return gpis3_impl_frame_record((GPisMap3*)m, buf, cap); }
int gpis3_train_deferred(void* m) { if (!m) return GPIS_ERR_ARG; try { return gpis3_impl_train_deferred((GPisMap3*)m); } catch (...) { return GPIS_ERR_STATE; } }
int gpis3_apply_frame(void* m, const void* buf, long long bytes) { if (!m) return GPIS_ERR_ARG; return gpis3_impl_apply_frame((GPisMap3*)m, buf, bytes); }
int gpis3_num_points(void* m) {
    if (!m) return GPIS_ERR_ARG;
    std::vector<float> p; ((GPisMap3*)m)->getAllPoints(p); return (int)(p.size() / 3);
}
int gpis3_get_points(void* m, float* out, int cap) {
    if (!m) return GPIS_ERR_ARG;
    std::vector<float> p; ((GPisMap3*)m)->getAllPoints(p);
    int n = (int)(p.size() / 3);
    if (out && n <= cap && n > 0) std::memcpy(out, p.data(), p.size() * sizeof(float));
    return n;
}
int gpis3_get_nodes(void* m, float* out, int cap) {
    if (!m) return GPIS_ERR_ARG;
    std::vector<float> p; ((GPisMap3*)m)->getAllNodes(p);
    int n = (int)(p.size() / 9);
    if (out && n <= cap && n > 0) std::memcpy(out, p.data(), p.size() * sizeof(float));
    return n;
}
int gpis3_save(void* m, const char* path) { if (!m || !path) return GPIS_ERR_ARG; return ((GPisMap3*)m)->saveMap(path) ? GPIS_OK : GPIS_ERR_ARG; }
int gpis3_load(void* m, const char* path) {
    if (!m || !path) return GPIS_ERR_ARG;
    if (((GPisMap3*)m)->loadMap(path)) return GPIS_OK;
    const int rc = gpis3_impl_update_fail((GPisMap3*)m);
    return rc ? rc : GPIS_ERR_ARG;
}
int gpis3_stats(void* m, double* out, int n) { if (!m || !out) return GPIS_ERR_ARG; gpis3_impl_stats((GPisMap3*)m, out, n); return GPIS_OK; }
int gpis3_sync(void* m) { if (!m) return GPIS_ERR_ARG; try { return gpis3_impl_sync((GPisMap3*)m); } catch (...) { return GPIS_ERR_STATE; } }
int gpis3_set_pipeline(void* m, int on) { if (!m) return GPIS_ERR_ARG; try { gpis3_impl_set_pipeline((GPisMap3*)m, on); return GPIS_OK; } catch (...) { return GPIS_ERR_STATE; } }
int gpis3_set_host_gather(void* m, int on) { if (!m) return GPIS_ERR_ARG; gpis3_impl_set_host_gather((GPisMap3*)m, on); return GPIS_OK; }
int gpis3_set_shard_factors(void* m, int mode) { if (!m) return GPIS_ERR_ARG; gpis3_impl_set_shard_factors((GPisMap3*)m, mode); return GPIS_OK; }
int gpis3_set_keep_factors(void* m, int on) { if (!m) return GPIS_ERR_ARG; try { gpis3_impl_set_keep_factors((GPisMap3*)m, on); return GPIS_OK; } catch (...) { return GPIS_ERR_STATE; } }
int gpis3_prepare_test(void* m) { if (!m) return GPIS_ERR_ARG; try { return gpis3_impl_prepare_test((GPisMap3*)m); } catch (...) { return GPIS_ERR_STATE; } }
int gpis3_set_lazy_inverse(void* m, int on) { if (!m) return GPIS_ERR_ARG; gpis3_impl_set_lazy_inverse((GPisMap3*)m, on); return GPIS_OK; }
int gpis3_set_profile(void* m, int on) { if (!m) return GPIS_ERR_ARG; gpis3_impl_profile((GPisMap3*)m, on); return GPIS_OK; }

// ---- 2-D map ----------------------------------------------------------------
void* gpis2_create(void) { try { return new GPisMap(); } catch (...) { return nullptr; } }
void gpis2_destroy(void* m) { delete (GPisMap*)m; }
int gpis2_reset(void* m) { if (!m) return GPIS_ERR_ARG; ((GPisMap*)m)->reset(); return GPIS_OK; }
int gpis2_update(void* m, const float* thetas, const float* ranges, int n, const float* pose6) {
    if (!m || !thetas || !ranges || !pose6) return GPIS_ERR_ARG;
    if (gpis_device_count() < 1) return GPIS_ERR_HIP;
    try {
        std::vector<float> pose(pose6, pose6 + 6);
        ((GPisMap*)m)->update(const_cast<float*>(thetas), const_cast<float*>(ranges), n, pose);
    } catch (...) { return GPIS_ERR_STATE; }
    return gpis2_impl_update_fail((GPisMap*)m);
}
int gpis2_test(void* m, const float* x, int dim, int n, float* res) {
    if (!m) return GPIS_ERR_ARG;
    if (gpis_device_count() < 1) return GPIS_ERR_HIP;
    try { if (((GPisMap*)m)->test(const_cast<float*>(x), dim, n, res)) return GPIS_OK; int e = gpis2_impl_fail((GPisMap*)m); return e ? e : GPIS_ERR_ARG; }
    catch (...) { return GPIS_ERR_STATE; }
}
int gpis2_test_device(void* m, const float* d_x, int n, float* d_res, void* stream) {
    if (!m) return GPIS_ERR_ARG;
    try { if (((GPisMap*)m)->testDevice(d_x, n, d_res, stream)) return GPIS_OK; int e = gpis2_impl_fail((GPisMap*)m); return e ? e : GPIS_ERR_ARG; }
    catch (...) { return GPIS_ERR_STATE; }
}
int gpis2_device(void* m) { if (!m) return GPIS_ERR_ARG; return gpis2_impl_device((GPisMap*)m); }
int gpis2_sync(void* m) { if (!m) return GPIS_ERR_ARG; try { return gpis2_impl_sync((GPisMap*)m); } catch (...) { return GPIS_ERR_STATE; } }
int gpis2_set_pipeline(void* m, int on) { if (!m) return GPIS_ERR_ARG; try { gpis2_impl_set_pipeline((GPisMap*)m, on); } catch (...) { return GPIS_ERR_STATE; } return GPIS_OK; }
int gpis2_get_nodes(void* m, float* out, int cap) {
    if (!m) return GPIS_ERR_ARG;
    std::vector<float> p; ((GPisMap*)m)->getAllNodes(p);
    int n = (int)(p.size() / 7);
    if (out && n <= cap && n > 0) std::memcpy(out, p.data(), p.size() * sizeof(float));
    return n;
}
int gpis2_stats(void* m, double* out, int n) { if (!m || !out) return GPIS_ERR_ARG; gpis2_impl_stats((GPisMap*)m, out, n); return GPIS_OK; }

// ---- ObsGP --------------------------------------------------------------------
struct ObsHandle { int device = -1; ObsGPDevice g; hipStream_t s = nullptr; };
void* gpis_obsgp_create(void) {
    if (gpis_device_count() < 1) { fprintf(stderr, "[gpismap_amd] no HIP device\n"); return nullptr; }
    ObsHandle* h = new (std::nothrow) ObsHandle();
    if (h) (void)hipGetDevice(&h->device);
    if (h && hipStreamCreate(&h->s) != hipSuccess) { delete h; return nullptr; }
    return h;
}
void gpis_obsgp_destroy(void* g) { if (!g) return; ObsHandle* h = (ObsHandle*)g; DeviceScope dev_scope_(h->device); if (h->s) (void)hipStreamDestroy(h->s); delete h; }
int gpis_obsgp_train2d(void* g, const float* vu, const float* f, int ni, int nj) {
    if (!g) return GPIS_ERR_ARG; ObsHandle* h = (ObsHandle*)g; DeviceScope dev_scope_(h->device); return h->g.train2d(vu, f, ni, nj, h->s);
}
int gpis_obsgp_train1d(void* g, const float* th, const float* f, int n) {
    if (!g) return GPIS_ERR_ARG; ObsHandle* h = (ObsHandle*)g; DeviceScope dev_scope_(h->device); return h->g.train1d(th, f, n, h->s);
}
int gpis_obsgp_query(void* g, const float* q, int nq, float* val, float* var) {
    if (!g || !q || !val || !var) return GPIS_ERR_ARG; ObsHandle* h = (ObsHandle*)g; DeviceScope dev_scope_(h->device); return h->g.query(q, nq, val, var, h->s);
}
int gpis_obsgp_num_groups(void* g) { if (!g) return GPIS_ERR_ARG; return ((ObsHandle*)g)->g.ngroups(); }
int gpis_obsgp_get_group(void* g, int group, int* n, float* x, float* alpha, float* L) {
    if (!g || !n) return GPIS_ERR_ARG; ObsHandle* h = (ObsHandle*)g; DeviceScope dev_scope_(h->device); return h->g.get_group(group, n, x, alpha, L, h->s);
}

// ---- OnGPIS -------------------------------------------------------------------
struct OnHandle {
    int device = -1;     // the device current at creation: every entry makes it current for its duration
    OnGPISStore st; hipStream_t s = nullptr;
    float* d_xq = nullptr; float* d_out = nullptr; size_t cap_xq = 0, cap_out = 0;
    OnHandle(int dim, float scale) : st(dim, scale) {}
};
void* gpis_ongpis_create(int dim, float scale) {
    if (dim != 2 && dim != 3) return nullptr;
    if (gpis_device_count() < 1) { fprintf(stderr, "[gpismap_amd] no HIP device\n"); return nullptr; }
    OnHandle* h = new (std::nothrow) OnHandle(dim, scale);
    if (h) (void)hipGetDevice(&h->device);
    if (h && hipStreamCreate(&h->s) != hipSuccess) { delete h; return nullptr; }
    if (h) h->st.profile = true;
    return h;
}
void gpis_ongpis_destroy(void* s) {
    if (!s) return; OnHandle* h = (OnHandle*)s;
    DeviceScope dev_scope_(h->device);
    (void)hipFree(h->d_xq); (void)hipFree(h->d_out);
    if (h->s) (void)hipStreamDestroy(h->s);
    delete h;
}
int gpis_ongpis_train(void* s, const float* soa9, int npts, const int* off, const int* ids, int ncl, int* model_out) {
    if (!s || !soa9 || !off || !ids || ncl < 0) return GPIS_ERR_ARG;
    OnHandle* h = (OnHandle*)s;
    DeviceScope dev_scope_(h->device);
    int dim = h->st.dim();
    int rc = h->st.upload_points(soa9, npts, h->s);
    if (rc) return rc;
    std::vector<TrainJob> jobs;
    std::vector<int> idv(ids, ids + off[ncl]);
    for (int c = 0; c < ncl; ++c) {
        TrainJob j; j.model = h->st.new_slot(); j.off = off[c]; j.n = off[c + 1] - off[c]; j.ng = 0;
        for (int k = j.off; k < j.off + j.n; ++k) {
            int id = ids[k];
            if (id < 0 || id >= npts) return GPIS_ERR_ARG;
            bool tiny = true;
            for (int d = 0; d < dim; ++d) tiny = tiny && ((double)fabsf(soa9[(size_t)(3 + d) * npts + id]) < 1e-6);
            if (!(((double)soa9[(size_t)8 * npts + id] > 0.1001) || tiny)) ++j.ng;
        }
        if (model_out) model_out[c] = j.model;
        jobs.push_back(j);
    }
    return h->st.train_batch(jobs, idv, h->s);
}
int gpis_ongpis_model_dims(void* s, int model, int* d4) {
    if (!s || !d4) return GPIS_ERR_ARG;
    const ClusterModel* m = ((OnHandle*)s)->st.model(model);
    if (!m || !m->base) return GPIS_ERR_ARG;
    d4[0] = m->N; d4[1] = m->ng; d4[2] = m->K; d4[3] = m->ld;
    return GPIS_OK;
}
int gpis_ongpis_get_model(void* s, int model, float* L, float* alpha, int* gidx) {
    if (!s) return GPIS_ERR_ARG;
    OnHandle* h = (OnHandle*)s;
    DeviceScope dev_scope_(h->device);
    const ClusterModel* m = h->st.model(model);
    if (!m || !m->base) return GPIS_ERR_ARG;
    if (!m->L) return GPIS_ERR_STATE;   // imported (predict-only) model: no factor on this rank
    if (L) GPIS_HIP(hipMemcpyAsync(L, m->L, sizeof(float) * (size_t)m->ld * m->ld, hipMemcpyDeviceToHost, h->s));
    if (alpha) GPIS_HIP(hipMemcpyAsync(alpha, m->alpha, sizeof(float) * m->K, hipMemcpyDeviceToHost, h->s));
    if (gidx) GPIS_HIP(hipMemcpyAsync(gidx, m->gidx, sizeof(int) * m->N, hipMemcpyDeviceToHost, h->s));
    GPIS_HIP(hipStreamSynchronize(h->s));
    return GPIS_OK;
}
int gpis_ongpis_eval(void* s, const float* xq, int nq, const int* job_q, const int* job_model, int njobs, float* out8) {
    if (!s || !xq || !job_q || !job_model || !out8 || nq < 1 || njobs < 1) return GPIS_ERR_ARG;
    OnHandle* h = (OnHandle*)s;
    DeviceScope dev_scope_(h->device);
    int dim = h->st.dim();
    std::vector<float> x4((size_t)4 * nq, 0.f);
    for (int i = 0; i < nq; ++i) for (int d = 0; d < dim; ++d) x4[(size_t)4 * i + d] = xq[(size_t)dim * i + d];
    if (x4.size() > h->cap_xq) { (void)hipFree(h->d_xq); h->d_xq = nullptr; GPIS_HIP(hipMalloc(&h->d_xq, sizeof(float) * x4.size())); h->cap_xq = x4.size(); }
    size_t no = (size_t)8 * njobs;
    if (no > h->cap_out) { (void)hipFree(h->d_out); h->d_out = nullptr; GPIS_HIP(hipMalloc(&h->d_out, sizeof(float) * no)); h->cap_out = no; }
    GPIS_HIP(hipMemcpyAsync(h->d_xq, x4.data(), sizeof(float) * x4.size(), hipMemcpyHostToDevice, h->s));
    GPIS_HIP(hipMemsetAsync(h->d_out, 0, sizeof(float) * no, h->s));
    int rc = h->st.eval_jobs(h->d_xq, job_q, job_model, njobs, h->d_out, h->s);
    if (rc && rc != GPIS_ERR_STATE) return rc;
    // (GPIS_ERR_STATE = the kernels' error word: the results still travel -- the affected ones are NaN -- and the call fails)
    GPIS_HIP(hipMemcpyAsync(out8, h->d_out, sizeof(float) * no, hipMemcpyDeviceToHost, h->s));
    GPIS_HIP(hipStreamSynchronize(h->s));
    return rc;
}
long long gpis_ongpis_packed_bytes(void* s, const int* models, int n) {
    if (!s || (!models && n > 0) || n < 0) return GPIS_ERR_ARG;
    return (long long)((OnHandle*)s)->st.packed_bytes(models, n);
}
int gpis_ongpis_pack(void* s, const int* models, int n, void* d_buf, long long stride, void* stream) {
    if (!s || !models || !d_buf || n < 0 || stride < 256 || stride % 256 != 0) return GPIS_ERR_ARG;
    OnHandle* h = (OnHandle*)s;
    DeviceScope dev_scope_(h->device);
    return h->st.pack_models(models, n, d_buf, (size_t)stride, stream ? (hipStream_t)stream : h->s);
}
int gpis_ongpis_unpack(void* s, const void* d_buf, int n, long long stride, int* models_inout, void* stream) {
    if (!s || !d_buf || !models_inout || n < 0 || stride < 256 || stride % 256 != 0) return GPIS_ERR_ARG;
    OnHandle* h = (OnHandle*)s;
    DeviceScope dev_scope_(h->device);
    return h->st.unpack_models(d_buf, n, (size_t)stride, models_inout, stream ? (hipStream_t)stream : h->s);
}
int gpis_ongpis_set_exp_table(void* s, int on) {
    if (!s) return GPIS_ERR_ARG;
    ((OnHandle*)s)->st.use_exp_table = on != 0;
    return GPIS_OK;
}
int gpis_ongpis_kernel_matrix(void* s, const float* x, const int* gidx, const float* sigx, const float* sigg, int n, float* K_out) {
    if (!s) return GPIS_ERR_ARG;
    OnHandle* h = (OnHandle*)s;
    DeviceScope dev_scope_(h->device);
    return h->st.kernel_matrix(x, gidx, sigx, sigg, n, K_out, h->s);
}
int gpis_ongpis_set_keep_factor(void* s, int on) {
    if (!s) return GPIS_ERR_ARG;
    ((OnHandle*)s)->st.keep_factor = on != 0;
    return GPIS_OK;
}
int gpis_ongpis_set_fused(void* s, int on) {
    if (!s) return GPIS_ERR_ARG;
    ((OnHandle*)s)->st.use_fused = on != 0;
    return GPIS_OK;
}
int gpis_ongpis_set_lazy_inverse(void* s, int on) {
    if (!s) return GPIS_ERR_ARG;
    ((OnHandle*)s)->st.lazy_inverse = on != 0;
    return GPIS_OK;
}
int gpis_ongpis_set_debug(void* s, int inject, int wait_limit_ms) {
    if (!s || wait_limit_ms < 0 || wait_limit_ms > 20000) return GPIS_ERR_ARG;
    OnHandle* h = (OnHandle*)s;
    DeviceScope dev_scope_(h->device);
    h->st.debug_inject = inject;
    h->st.wait_limit_ticks = wait_limit_ms * 100000;     // 100 MHz device clock
    return GPIS_OK;
}
int gpis_ongpis_set_cu_reserve(void* s, int n) {
    if (!s || n < 0) return GPIS_ERR_ARG;
    OnHandle* h = (OnHandle*)s;
    DeviceScope dev_scope_(h->device);
    const int rc = h->st.set_cu_reserve(n);      // (joins a batch in flight, re-creates the side streams at the next training)
    if (rc != GPIS_OK) return rc;
    // the handle's own stream carries the largest clusters (the cooperative launch): masked like the others
    hipStream_t ns = nullptr;
    if (int src = ongpis_make_train_stream(&ns, n)) return src;
    if (h->s) { (void)hipStreamSynchronize(h->s); (void)hipStreamDestroy(h->s); }
    h->s = ns;
    return GPIS_OK;
}
int gpis_ongpis_last_ms(void* s, float* t, float* e) {
    if (!s) return GPIS_ERR_ARG;
    OnHandle* h = (OnHandle*)s;
    DeviceScope dev_scope_(h->device);
    if (t) *t = h->st.last_train_ms;
    if (e) *e = h->st.last_eval_ms;
    return GPIS_OK;
}

}  // extern "C"
