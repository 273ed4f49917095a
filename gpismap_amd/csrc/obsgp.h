// Host-side owner of the device-resident observation GP (ObsGP2D / ObsGP1D of
// reference cpp/include/ObsGP.h:77-142) and the kernel-argument view.
#pragma once
#include <vector>
#include "dev_common.h"

namespace gpis {

struct ObsGPView {
    int mode;      // 2: regular 2-D grid tiles (ObsGP2D); 1: 1-D ranges (ObsGP1D)
    int ni, nj;    // grid size (mode 2)
    int ng0, ng1;  // groups per axis (mode 2)
    int ngroups;
    const float* x;  // mode 2: interleaved (v,u) per pixel; mode 1: theta
    const float* f;  // observations (1/z, or 1/sqrt(r))
    const int *i0, *i1, *j0, *j1;  // mode 2 tile index ranges
    const int *ga, *glen;          // mode 1 group start/length
    const float *vali, *valj;      // boundary tables (mode 1: `range` in vali)
    int* tn;        // [ngroups] number of training points (0 = untrained)
    float* tx;      // [ngroups][64][2]
    float* talpha;  // [ngroups][64]
    float* tL;      // [ngroups][64*64] column-major lower factor
};

void obsgp_launch_train(const ObsGPView& v, hipStream_t s);
void obsgp_launch_query(const ObsGPView& v, const float* d_q, int nq, float* d_val, float* d_var, hipStream_t s);
// the same answers with the queries sorted by group on the device first (counting sort; scratch: 2 nq + 2 (ngroups + 1) ints):
// a wave then meets one or two groups instead of every group of its 64 consecutive queries
void obsgp_launch_query_binned(const ObsGPView& v, const float* d_q, int nq, float* d_val, float* d_var, int* scratch, hipStream_t s);

// Device-resident ObsGP.  train*() re-trains every group from host inputs
// (reference GPisMap3::regressObs, GPisMap3.cpp:239-256); query() answers a batch
// of single-point queries (every reference call site queries one point).
class ObsGPDevice {
public:
    ObsGPDevice();
    ~ObsGPDevice();
    // ObsGP2D::train (ObsGP.cpp:331-342).  The partition tables are computed on the
    // first call for a grid size and kept afterwards (the reference never re-partitions,
    // SURVEY B-7/B-15).
    int train2d(const float* vu_grid, const float* f, int ni, int nj, hipStream_t s);
    // ObsGP1D::train (ObsGP.cpp:85-143)
    int train1d(const float* theta, const float* f, int n, hipStream_t s);
    // q: nq*2 (mode 2) or nq (mode 1) host floats; val/var host outputs.  val entries
    // are pre-filled by the caller (untouched when no group answers).
    int query(const float* q, int nq, float* val, float* var, hipStream_t s);
    int query_device(const float* d_q, int nq, float* d_val, float* d_var, hipStream_t s);
    // update()'s batches: the caller writes its queries straight into page-locked staging (stage_q), query_staged()
    // moves them with true DMA transfers, zero-fills the values on the device (every map caller starts from val = 0)
    // and leaves the answers in staged_val()/staged_var() until the next stage_q().
    float* stage_q(int nq);
    int query_staged(int nq, hipStream_t s);
    const float* staged_val() const { return h_val_; }
    const float* staged_var() const { return h_var_; }
    // A second, independent staging set for a batch that runs BESIDE the host (the new-pixel batch of update(), issued on
    // its own stream before the re-evaluation batches and collected after them): stage_qb() / query_staged_b_async() /
    // wait_b(), answers in staged_val_b() / staged_var_b().  Same kernel, same zero-filled values.
    float* stage_qb(int nq);
    int query_staged_b_async(int nq, hipStream_t s);
    int wait_b();
    const float* staged_val_b() const { return hb_val_; }
    const float* staged_var_b() const { return hb_var_; }
    bool trained() const { return trained_; }
    int mode() const { return view_.mode; }
    int ngroups() const { return view_.ngroups; }
    int trained_groups(hipStream_t s);
    // debug / parity: copy one group back (n, x[128], alpha[64], L[4096])
    int get_group(int g, int* n, float* x, float* alpha, float* L, hipStream_t s);
    // map reset(): the reference deletes its ObsGP object (GPisMap3.cpp:105-108), so the partition kept across
    // frames goes with it and the next train2d() partitions afresh
    void reset_trained() { trained_ = false; sz0_ = sz1_ = 0; }

private:
    int ensure_groups(int ngroups);
    int ensure_io(size_t nx, size_t nf);
    int ensure_q(int nq);
    ObsGPView view_{};
    bool trained_ = false;
    int sz0_ = 0, sz1_ = 0;  // szSamples (ObsGP.h:107)
    std::vector<int> h_i0_, h_i1_, h_j0_, h_j1_;
    std::vector<float> h_vali_, h_valj_;
    int cap_groups_ = 0;
    size_t cap_x_ = 0, cap_f_ = 0;
    int cap_q_ = 0;
    float *d_x_ = nullptr, *d_f_ = nullptr;
    int *d_idx_ = nullptr;      // i0,i1,j0,j1 / ga,glen packed
    float* d_tab_ = nullptr;    // vali, valj packed
    int cap_idx_ = 0, cap_tab_ = 0;
    float *d_q_ = nullptr, *d_val_ = nullptr, *d_var_ = nullptr;
    float *h_q_ = nullptr, *h_val_ = nullptr, *h_var_ = nullptr;   // page-locked staging
    int cap_hq_ = 0;
    float *db_q_ = nullptr, *db_val_ = nullptr, *db_var_ = nullptr;   // the second set
    float *hb_q_ = nullptr, *hb_val_ = nullptr, *hb_var_ = nullptr;
    int cap_qb_ = 0, cap_hqb_ = 0;
    hipEvent_t evb_ = nullptr;
    bool b_pending_ = false;
    int* d_bin_[2] = {nullptr, nullptr};     // scratch of the device-side sort by group, one per staging set
    size_t cap_bin_[2] = {0, 0};
    int launch_query(int set, const float* d_q, int nq, float* d_val, float* d_var, hipStream_t s);
};

}  // namespace gpis
