/* gpismap_amd -- C-ABI of the MI355X-native GPisMap hot path.
 *
 * Plain pointers and sizes only; every function returns an int status
 * (0 = GPIS_OK, negative = error) unless stated otherwise and never throws.
 * Host pointers unless a parameter is named d_* (device pointer).
 *
 * Map level: what a binding of the reference's mex gateways would call.
 *   reference mex/mexGPisMap3.cpp:  'update' :49-78, 'test' :79-110,
 *   'setCamera' :111-144, 'getAllPoints' :145-157, 'reset' :158-166
 *   -> class GPisMap3  reference cpp/include/GPisMap3.h:117-127
 *   reference mex/mexGPisMap.cpp:   'update' :40-85, 'test' :86-122, 'reset' :123-131
 *   -> class GPisMap   reference cpp/include/GPisMap.h:97-106
 * Kernel level: the batched GP primitives.
 */
#ifndef GPISMAP_AMD_H_
#define GPISMAP_AMD_H_

#ifdef __cplusplus
extern "C" {
#endif

#define GPIS_OK 0
#define GPIS_ERR_ARG (-1)
#define GPIS_ERR_HIP (-2)
#define GPIS_ERR_STATE (-3)
#define GPIS_ERR_LIMIT (-4)

typedef struct gpis_cam {  /* reference camParam, GPisMap3.h:29-46 */
    float fx, fy, cx, cy;
    int width, height;
} gpis_cam;

/* number of HIP devices visible (0 when no GPU: every compute entry then fails loudly) */
int gpis_device_count(void);
/* The library keeps the standard-size chunks of destroyed device pools per device for the next map of the process (bounded by
 * GPIS_POOL_CACHE_GB, default 16).  gpis_pool_cache_trim() hands them back to the driver -- call it when another library in the
 * process needs the memory; returns the bytes released.  No map may be in use on another thread during the call. */
unsigned long long gpis_pool_cache_trim(void);
const char* gpis_version(void);
/* Device selection (one process per GPU in a multi-GPU job: call once with LOCAL_RANK before creating anything).
 * Every object created afterwards lives on the device current at its creation and makes it current inside each
 * call; d_* pointers and streams handed to *_test_device must belong to that device.  GPIS_ERR_ARG if out of range. */
int gpis_set_device(int device);
int gpis_get_device(void);

/* ---- 3-D map (GPisMap3) -------------------------------------------------- */
void* gpis3_create(const gpis_cam* cam /* NULL = reference defaults */);
/* ONE map object over several devices of this process (a device may be listed more than once: logical shards on one GPU).
 * update() runs on every listed device from its own host thread, each trains its K^3-balanced share of the frame's
 * clusters, the packed models are copied device to device and every device ends up with the whole map; test() deals the
 * queries to the devices in blocks of 65 536 and answers them concurrently.  Results are bit-identical to a one-device
 * map.  The GPisMap3 class does the same when GPIS_DEVICES=0,1,... is set in the environment (so the unchanged mex
 * gateway uses every GPU).  Reference: the fan-out over host threads, GPisMap3.cpp:759-784 (updateGPs), :904-949 (test). */
void* gpis3_create_multi(const gpis_cam* cam /* NULL = reference defaults */, const int* devices, int n);
int   gpis3_num_devices(void* map);
void  gpis3_destroy(void* map);
int   gpis3_reset(void* map);                                   /* GPisMap3::reset    GPisMap3.cpp:99  */
int   gpis3_set_camera(void* map, const gpis_cam* cam);         /* GPisMap3::resetCam GPisMap3.cpp:117 */
/* depth: width*height floats, column-major (index = col*height + row), metres;
 * pose12 = [t(3), R column-major(9)].                          GPisMap3::update GPisMap3.cpp:218 */
int   gpis3_update(void* map, const float* depth, int n, const float* pose12);
/* x: n*3 interleaved; res: n*8 [f gx gy gz vf vgx vgy vgz], pre-filled by the caller; only the
 * entries the reference writes are touched.  Returns GPIS_ERR_ARG where the reference returns
 * false; a failure of the device path is never folded into that: GPIS_ERR_HIP / _STATE / _LIMIT.
 *                                                             GPisMap3::test GPisMap3.cpp:904 */
int   gpis3_test(void* map, const float* x, int dim, int n, float* res);
int   gpis3_test_device(void* map, const float* d_x, int n, float* d_res, void* hip_stream);
int   gpis3_device(void* map);                                 /* device the map lives on, or negative */
/* ---- multi-GPU: sharded cluster training (one process per GPU).  Every rank runs the same update()
 * (host logic is deterministic, so trees, cluster sets and model slots agree), but after gpis3_set_shard(rank, world)
 * it TRAINS only its share of the frame's clusters (greedy longest-processing-time partition by K^3).  The caller then
 * moves the packed models between the ranks and completes the update.  A rank's records (what prediction reads of each of
 * its models: 2 K^2 + 20 K bytes) sit BACK TO BACK at their own sizes in the frame's job order; every rank can size every
 * rank's buffer (gpis3_shard_bytes), so an exchange moves the records' bytes and no padding between them:
 *   gpis3_update(...);                                  trains the local share, defers the cluster table
 *   gpis3_shard_info(map, out, 2 + world)               out[0] = clusters of this frame, out[1] = local ones,
 *                                                       out[2 + r] = clusters rank r trains
 *   gpis3_shard_bytes(map, r)                           bytes of rank r's records (the same answer on every rank)
 *   gpis3_shard_pack(map, d_send, stream)               local models -> d_send[0 .. gpis3_shard_bytes(map, rank))
 *   ... transport: e.g. one RCCL all_gather_into_tensor of buffers padded to the largest RANK total (the K^3-balanced
 *       partition makes the totals near-equal), grouped send/recv, or hipMemcpyPeerAsync inside one process ...
 *   gpis3_shard_unpack(map, r, d_recv_r, stream)        for every other rank r
 *   gpis3_shard_finish(map)                             builds the cluster table: test() is valid again
 * With world = 1 (default) update() is complete on return and none of the calls is needed. */
int   gpis3_set_shard(void* map, int rank, int world);
/* What the records of a sharded update carry (same size either way): mode 1 = FACTOR records for models whose explicit inverse is
 * still pending (lazy inverse: -L re-tiled + alpha; the receiver computes X = L^-1 when it first predicts with the model, so no
 * rank inverts a cluster that is retrained before anybody asks), mode 0 = prediction records (X; packing forces the inverse on
 * the owner), mode -1 (default) = 1 with the lazy inverse, 0 with the eager one.  Also applies to maps over several devices.
 * gpis3_stats out[27] = factor records received in the last exchange. */
int   gpis3_set_shard_factors(void* map, int mode);
int   gpis3_shard_info(void* map, int* out, int n);
long long gpis3_shard_bytes(void* map, int owner);
int   gpis3_shard_pack(void* map, void* d_buf, void* hip_stream);
int   gpis3_shard_unpack(void* map, int owner, const void* d_buf, void* hip_stream);
int   gpis3_shard_finish(void* map);
/* One process per GPU, the host logic of update() run ONCE (round 6).  By default every rank of a sharded run replays the whole
 * frame (tree mutation, ObsGP round trips: the larger half of an update) and only the training is divided.  Alternatively the
 * LEAD rank (rank 0 of gpis3_set_shard) records what its update() decided and the other ranks apply the record:
 *   lead:    gpis3_set_frame_export(map, 1) once;  per frame  gpis3_update(...)           host logic, record written, own share NOT trained yet
 *                                                             n = gpis3_frame_record(map, buf, cap)   (a few hundred KB .. MB: the point mirror dominates)
 *                                                             ... broadcast buf[0..n) to the workers ...
 *                                                             gpis3_train_deferred(map)   the lead's own share
 *   worker:  per frame  gpis3_apply_frame(map, buf, n)        slot operations mirrored (the ids must come out as the lead's), point
 *                                                             mirror uploaded, K6 on its own device, its share trained
 *   all:     the exchange as above (gpis3_shard_info / _bytes / _pack / _unpack / _finish).
 * gpis3_frame_record returns the record's size (copies it when cap suffices); a worker map never calls gpis3_update (it holds no
 * tree: gpis3_get_points answers on the lead only) and is created for that role: gpis3_apply_frame on a map that has replayed
 * frames itself returns GPIS_ERR_STATE.  gpis3_stats out[26] (host replays of the last update) is 1 on the lead and 0 on a worker.
 * reset / loadMap on the lead are not recorded: recreate the workers with it. */
int   gpis3_set_frame_export(void* map, int on);
long long gpis3_frame_record(void* map, void* buf, long long cap);
int   gpis3_train_deferred(void* map);
int   gpis3_apply_frame(void* map, const void* buf, long long bytes);
int   gpis3_num_points(void* map);
int   gpis3_get_points(void* map, float* out3, int cap);        /* GPisMap3::getAllPoints GPisMap3.cpp:951 */
int   gpis3_get_nodes(void* map, float* out9, int cap);         /* pos3 grad3 val sigx sigg, tree order */
/* out[0..27]: obsgp groups trained, obsgp queries, clusters trained (cumulative), late re-evaluations, clusters in table,
 * GP evaluations of last test, ms in K4 of last test (profiling on), device bytes, algorithmic flops of last test, K4 launches,
 * ms in K6+K3+K3b of last update (profiling on), model bytes, update phases ms [preproc, ObsGP train, re-evaluation,
 * new points, updateGPs], algorithmic flops / bytes / clusters / largest K of the last training batch, ms and clusters of the
 * last deferred inverse pass (profiling on), bytes received in the last in-library model exchange, pipelined update on (0/1),
 * CUs the training streams leave free, host replays of the last update() over all devices of the map (1: the host logic ran
 * once, on the lead device, however many devices train), factor records received in the last model exchange (inverses deferred) */
int   gpis3_stats(void* map, double* out, int n);
/* Map checkpoint (optional; the reference keeps its map only in the mex singleton): gpis3_save writes the spatial
 * index, the surface points with their data and every trained model as its packed prediction record (about 2 K^2 bytes per
 * cluster: the file of a 500-cluster map is ~1.2 GB) (GPisMap3::saveMap); gpis3_load replaces the map's state with a file's.
 * Nothing is retrained -- the records are restored verbatim, so test() answers with the same bits as before the save and the
 * next update continues from it; restored models are predict-only until their cluster is trained again.  The file is a raw image
 * of this build's structures with a checksum over the payload: another build's, a truncated or a damaged file is refused
 * (GPIS_ERR_ARG) and the map stays as it was.  The camera is not part of it. */
int   gpis3_save(void* map, const char* path);
int   gpis3_load(void* map, const char* path);
int   gpis3_set_profile(void* map, int on);
/* Pipelined update (the default since round 4; gpis3_set_pipeline(map, 0) or GPIS_PIPELINE_UPDATE=0 in the environment select
 * the reference's synchronous update(), GPisMap3.cpp:218-237): gpis3_update() returns once the frame's OnGPIS training is
 * enqueued; the next update's training, test, the getters, statistics and the sharded exchange join it first, so results never
 * depend on the mode -- only WHEN a training failure is reported does: by the call that joined.  gpis3_sync() joins explicitly
 * and returns the pending update status.  While it is on, the training streams are kept off GPIS_PIPELINE_RESERVE_CUS CUs
 * (default 32, spread evenly over the XCDs) so that the next frame's ObsGP batches never wait for a factorisation workgroup.
 * Maps that shard their training over ranks or devices run synchronously whatever the setting (every frame ends with the
 * exchange). */
int   gpis3_sync(void* map);
int   gpis3_set_pipeline(void* map, int on);
/* Cross-check paths, reachable through these setters only (no environment switch since round 5):
 *  - K6's range part (which points of the touched cells lie in a cluster's range, GPisMap3.cpp:721-735) runs on the device;
 *    gpis3_set_host_gather(map, 1) selects the host walk it replaced -- same training sets;
 *  - a cluster keeps only what prediction reads once its inverse exists; gpis3_set_keep_factors(map, 1) keeps the training side
 *    (factor, re-tiled factor) of every model as well (10 K^2 instead of 2 K^2 bytes per cluster). */
int   gpis3_set_host_gather(void* map, int on);
int   gpis3_set_keep_factors(void* map, int on);
/* Lazy inverse at map level (default on; GPIS_EAGER_INVERSE=1 or gpis3_set_lazy_inverse(map, 0) turn it off): update() trains
 * factors and alpha; the explicit inverses are computed by the first test() after it (or by gpis3_prepare_test(), which also
 * joins a pipelined training) -- once per cluster, however many updates retrained it in between.  With a test() after
 * every update() the total work is unchanged; with several updates per test() the inverses of the intermediate factors
 * are never computed.  Results do not depend on the mode. */
int   gpis3_prepare_test(void* map);
int   gpis3_set_lazy_inverse(void* map, int on);

/* ---- 2-D map (GPisMap) ---------------------------------------------------- */
void* gpis2_create(void);                                       /* GPisMap() GPisMap.cpp:57 */
void  gpis2_destroy(void* map);
int   gpis2_reset(void* map);                                   /* GPisMap::reset GPisMap.cpp:90 */
/* thetas / ranges: n floats each (radians, metres); pose6 = [tx ty R11 R21 R12 R22].
 *                                                               GPisMap::update GPisMap.cpp:151 */
int   gpis2_update(void* map, const float* thetas, const float* ranges, int n, const float* pose6);
/* x: n*2 interleaved; res: n*6 [f gx gy vf vgx vgy], pre-filled by the caller.  GPisMap::test GPisMap.cpp:765 */
int   gpis2_test(void* map, const float* x, int dim, int n, float* res);
int   gpis2_test_device(void* map, const float* d_x, int n, float* d_res, void* hip_stream);
int   gpis2_device(void* map);
/* Round 6: the 2-D update() is pipelined like the 3-D one -- it returns once the frame's OnGPIS training is enqueued; the next
 * update, gpis2_test / _test_device, gpis2_stats and gpis2_sync join it (a failed training surfaces there).  gpis2_set_pipeline(map, 0)
 * or GPIS_PIPELINE_UPDATE=0 restore the synchronous call.  Same map state and test() results in both modes. */
int   gpis2_sync(void* map);
int   gpis2_set_pipeline(void* map, int on);
int   gpis2_get_nodes(void* map, float* out7, int cap);         /* pos2 grad2 val sigx sigg, tree order */
int   gpis2_stats(void* map, double* out, int n);               /* same slots as gpis3_stats */

/* ---- kernel level: observation GP (K1, K2) -------------------------------- */
void* gpis_obsgp_create(void);
void  gpis_obsgp_destroy(void* g);
/* ObsGP2D::train ObsGP.cpp:331: vu = ni*nj interleaved (v,u), f = ni*nj (valid iff > 0) */
int   gpis_obsgp_train2d(void* g, const float* vu, const float* f, int ni, int nj);
/* ObsGP1D::train ObsGP.cpp:85 */
int   gpis_obsgp_train1d(void* g, const float* theta, const float* f, int n);
/* batched single-point queries (ObsGP2D::test ObsGP.cpp:410 / ObsGP1D::test :145);
 * q: nq*2 (2-D) or nq (1-D); val is pre-filled by the caller and left untouched where no
 * group answers (var = 1e6 there) */
int   gpis_obsgp_query(void* g, const float* q, int nq, float* val, float* var);
int   gpis_obsgp_num_groups(void* g);
int   gpis_obsgp_get_group(void* g, int group, int* n, float* x128, float* alpha64, float* L4096);

/* ---- kernel level: OnGPIS batches (K6, K3, K4) ----------------------------- */
void* gpis_ongpis_create(int dim, float scale);
void  gpis_ongpis_destroy(void* s);
/* points: 9 SoA rows of length npts (px py pz gx gy gz val sigx sigg); clusters given as CSR
 * (off[ncl+1], ids[]) of point ids in training order.  model_out[ncl] receives the model slots.
 * OnGPIS::train OnGPIS.cpp:91-149 (2-D :34-89) for every cluster. */
int   gpis_ongpis_train(void* s, const float* points_soa9, int npts, const int* off, const int* ids, int ncl,
                        int* model_out);
/* copy a trained model back: sizes via gpis_ongpis_model_dims (N, ng, K, ld) */
int   gpis_ongpis_model_dims(void* s, int model, int* dims4);
int   gpis_ongpis_get_model(void* s, int model, float* L_ldxld, float* alpha_K, int* gidx_N);
/* OnGPIS::testSinglePoint OnGPIS.cpp:177-216 (2-D test2Dpoint :218) for njobs (query, model) pairs.
 * xq: nq*dim interleaved; out: njobs*8 = mean(4) var(4) (2-D uses 3+3, slots 3 and 7 unused) */
int   gpis_ongpis_eval(void* s, const float* xq, int nq, const int* job_q, const int* job_model, int njobs,
                       float* out8);
/* Rounds 2-4: K4 kept one double-precision exp per (training point, query) in an LDS table when it fit.  The kernel since round 5
 * (B chunks in a three-slot ring) spends that LDS on wider chunks and evaluates the exponential per entry for every cluster:
 * gpis_ongpis_set_exp_table is accepted for compatibility and changes nothing (results were identical either way). */
/* Packed model records for a multi-GPU exchange (what K4 needs from a trained model: 2 K^2 + 20 K bytes): pack the listed
 * models into d_buf (n records of `stride` bytes, stride >= gpis_ongpis_packed_bytes of every sender, a multiple of 256),
 * unpack records into predict-only models (models_inout[i] < 0: a new model is created and its id returned). */
long long gpis_ongpis_packed_bytes(void* s, const int* models, int n);
int   gpis_ongpis_pack(void* s, const int* models, int n, void* d_buf, long long stride, void* hip_stream);
int   gpis_ongpis_unpack(void* s, const void* d_buf, int n, long long stride, int* models_inout, void* hip_stream);
int   gpis_ongpis_set_exp_table(void* s, int on);
/* Clusters of at most 256 rows are trained by one fused on-chip kernel and keep only what prediction reads
 * (row table, points, the re-tiled inverse factor).  keep_factor(1) BEFORE training makes such models carry L, alpha
 * and gidx as well (gpis_ongpis_get_model returns GPIS_ERR_STATE for a model without them); set_fused(0) sends every
 * cluster through the separate gather / build / factorise / invert kernels (same results, bit for bit). */
/* kernel matrix only (the build kernel on caller-given arrays, no gather rule): x [n][dim], gidx [n] running gradient
 * index or -1, sigx / sigg [n]; K_out receives the K x K lower triangle, column-major, K = n + dim * #(gidx >= 0).
 * reference matern32_sparse_deriv1_3D / _2D (train), covFnc.cpp:142-256 / :317-402 */
int   gpis_ongpis_kernel_matrix(void* s, const float* x, const int* gidx, const float* sigx, const float* sigg, int n, float* K_out);
int   gpis_ongpis_set_keep_factor(void* s, int on);
int   gpis_ongpis_set_fused(void* s, int on);
/* Lazy inverse (default on): training of clusters of more than 256 rows stops at the factor and alpha (what
 * OnGPIS::train computes, OnGPIS.cpp:139-143); the explicit inverse the prediction kernel multiplies with is computed at the
 * first prediction / packing after a training, once per cluster however often it was retrained in between.  on = 0: the
 * inverse runs behind the factorisation in every training batch. */
int   gpis_ongpis_set_lazy_inverse(void* s, int on);
/* In-kernel waits (the cooperative factorisation of the largest clusters, the pipelined inverse) are bounded: when one
 * expires the batch's models are dropped and training returns GPIS_ERR_STATE.  wait_limit_ms = 0 keeps the default
 * (2 s); inject is a TEST hook: bit 0 makes one workgroup of every cooperative cluster withhold a hand-over; bit 4 (16)
 * makes the first workgroup of every prediction launch withhold one signal of its LDS ring (and shortens that kernel's
 * bounded wait): its tile's results are NaN, the error word of the launch is raised and gpis_ongpis_eval / test() return
 * GPIS_ERR_STATE -- so that the error paths can be exercised. */
int   gpis_ongpis_set_debug(void* s, int inject, int wait_limit_ms);
/* CUs the training streams of this handle leave free (what the maps do for their pipelined update: gpis3_set_pipeline /
 * GPIS_PIPELINE_RESERVE_CUS).  The streams are created with a CU mask over the first (CUs - n) bits -- bit i is CU i / 8 of XCD
 * i % 8 -- and the cooperative factorisation sizes its workgroup groups for what is left on one XCD.  0 = ordinary streams. */
int   gpis_ongpis_set_cu_reserve(void* s, int n);
/* Device self-test (round 6): the factorisation kernels take their square roots and divisions through range-restricted sequences
 * (csrc/tile_solve.h: the compiler's correctly rounded expansions without the operand scaling and classification the chains'
 * operands never need).  Runs blocks * 256 * per_thread random operand pairs through them and through the compiler's sqrtf and `/`
 * on the current device and returns the number of results whose bits differ: mismatches2[0] square roots, [1] divisions.
 * mode 0: the documented operand ranges; mode 1: operands shaped like the factorisations'; mode 2: the table-driven double-precision
 * exponential of the prediction / query kernels (csrc/exp_tab.h) against the device library's exp on arguments in [-12, 0] and a
 * few deep in the underflow range: mismatches2[0] = results more than one ulp apart, [1] = results that differ at all. */
int   gpis_selftest_ranged_arith(unsigned long long seed, int blocks, int per_thread, int mode, unsigned long long* mismatches2);
int   gpis_ongpis_last_ms(void* s, float* train_ms, float* eval_ms);

#ifdef __cplusplus
}
#endif
#endif
