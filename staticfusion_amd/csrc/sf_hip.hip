// sf_hip.hip — the product: libsf_hip.so, the MI355X (gfx950) implementation of include/sf.h.
//
// One persistent kernel, `sf_frame_kernel`, runs any subset of the frame stages for every
// stream of the batch; workgroups pull stream indices from an atomic queue.  Each C-ABI entry
// point that the reference exposes as a StaticFusion method is one launch of that kernel with
// the corresponding stage mask; sf_process_frame is ONE launch with the whole per-frame
// sequence of the reference drivers (StaticFusion-datasets.cpp:171-184).
//
// There is no CPU fallback and no dependence on the test oracle.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sf_cluster.h"
#include "sf_device_common.h"
#include "sf_input.h"
#include "sf_predict.h"
#include "sf_fusion.h"

// the two builds of the frame kernels (sf_frame_kernels.hip, -DSF_NT=256 / -DSF_NT=1024)
struct FrameVariant {
    int id;  // SF_VARIANT_*
    const char *name;
    void (*geometry)(int *threads, int *blocks_per_cu);
    void (*launch_frame)(int grid, hipStream_t st, const KArgs *ka, const FrameLaunch *fl);
    void (*launch_irls_pass)(int grid, hipStream_t st, const KArgs *ka, int which, int variant, int reps, int slices);
    void (*launch_debug_rows)(int grid, hipStream_t st, const KArgs *ka, int b, float *out);
};
extern "C" __attribute__((visibility("hidden"))) void sf_launch_frame_nt256(int, hipStream_t, const KArgs *, const FrameLaunch *);
extern "C" __attribute__((visibility("hidden"))) void sf_launch_irls_pass_nt256(int, hipStream_t, const KArgs *, int, int, int, int);
extern "C" __attribute__((visibility("hidden"))) void sf_launch_frame_nt256o5(int, hipStream_t, const KArgs *, const FrameLaunch *);
extern "C" __attribute__((visibility("hidden"))) void sf_variant_geometry_nt256o5(int *, int *);
extern "C" __attribute__((visibility("hidden"))) void sf_launch_frame_nt1024(int, hipStream_t, const KArgs *, const FrameLaunch *);
extern "C" __attribute__((visibility("hidden"))) void sf_launch_irls_pass_nt1024(int, hipStream_t, const KArgs *, int, int, int, int);
extern "C" __attribute__((visibility("hidden"))) void sf_launch_debug_rows_nt256(int, hipStream_t, const KArgs *, int, float *);
extern "C" __attribute__((visibility("hidden"))) void sf_launch_debug_rows_nt1024(int, hipStream_t, const KArgs *, int, float *);
extern "C" __attribute__((visibility("hidden"))) void sf_launch_frame_ntcluster(int, hipStream_t, const KArgs *, const FrameLaunch *);
extern "C" __attribute__((visibility("hidden"))) void sf_launch_irls_pass_ntcluster(int, hipStream_t, const KArgs *, int, int, int, int);
extern "C" __attribute__((visibility("hidden"))) void sf_launch_debug_rows_ntcluster(int, hipStream_t, const KArgs *, int, float *);
extern "C" __attribute__((visibility("hidden"))) void sf_variant_geometry_ntcluster(int *, int *);
extern "C" __attribute__((visibility("hidden"))) void sf_variant_geometry_nt256(int *, int *);
extern "C" __attribute__((visibility("hidden"))) void sf_variant_geometry_nt1024(int *, int *);
static const FrameVariant VARIANTS[3] = {
    {SF_VARIANT_THROUGHPUT, "throughput", sf_variant_geometry_nt256, sf_launch_frame_nt256, sf_launch_irls_pass_nt256, sf_launch_debug_rows_nt256},
    {SF_VARIANT_LATENCY, "latency", sf_variant_geometry_nt1024, sf_launch_frame_nt1024, sf_launch_irls_pass_nt1024, sf_launch_debug_rows_nt1024},
    {SF_VARIANT_CLUSTER, "cluster", sf_variant_geometry_ntcluster, sf_launch_frame_ntcluster, sf_launch_irls_pass_ntcluster, sf_launch_debug_rows_ntcluster},
};

// prediction := current; current := pool[frame_index[stream]] for every stream, 16 bytes per lane and plane
// (sf_advance_sequences_device). grid = (slices, batch).
// Longest-expected-first order of the streams of a launch (KArgs::order): a counting sort by the IRLS iterations each
// stream needed for its previous frame, descending. One workgroup; the order inside a bucket is whatever the atomics give
// (it only decides who runs when, never what is computed).
__global__ __launch_bounds__(1024) void sf_order_kernel(const sf_frame_stats *stats, int batch, int *order) {
    __shared__ int bins[256];
    const int tid = threadIdx.x;
    if (tid < 256) bins[tid] = 0;
    __syncthreads();
    for (int i = tid; i < batch; i += 1024) atomicAdd(&bins[255 - min(max(stats[i].n_irls, 0), 255)], 1);
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int q = 0; q < 256; q++) {
            const int n = bins[q];
            bins[q] = run;
            run += n;
        }
    }
    __syncthreads();
    for (int i = tid; i < batch; i += 1024) order[atomicAdd(&bins[255 - min(max(stats[i].n_irls, 0), 255)], 1)] = i;
}

// the nearest K-means seed of every level-1 pixel (KArgs::km_seed_lab): once per handle, with the device arithmetic
// sf_clear_sync_timeout: epochs and the sticky timeout flag of every stream (the granules are zeroed by a memset)
__global__ __launch_bounds__(256) void sf_clear_sync_kernel(StreamState *state, int batch) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < batch) {
        state[b].sync_epoch = 0;
        state[b].sync_failed = 0;
    }
}
__global__ __launch_bounds__(256) void sf_seed_label_kernel(uint8_t *out, int rows_km, int cols_km) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows_km * cols_km) return;
    const int u = idx / rows_km, v = idx - u * rows_km;
    out[idx] = (uint8_t)km_nearest_seed(rows_km, cols_km, (unsigned)u, (unsigned)v);
}

__global__ __launch_bounds__(256) void sf_advance_kernel(float *cur_d, float *cur_i, float *pred_d, float *pred_i, const float *pool_d,
                                                         const float *pool_i, const int *frame_index, int n0, int n_tot) {
    const int b = blockIdx.y;
    const int f = frame_index[b];
    if (f < 0) return;
    typedef float __attribute__((ext_vector_type(4))) f4;
    const size_t so = (size_t)b * n_tot, po = (size_t)f * n0;
    for (int q = (blockIdx.x * 256 + threadIdx.x) * 4; q < n0; q += gridDim.x * 256 * 4) {
        const f4 cd = *(const f4 *)(cur_d + so + q), ci = *(const f4 *)(cur_i + so + q);
        const f4 nd = *(const f4 *)(pool_d + po + q), ni = *(const f4 *)(pool_i + po + q);
        *(f4 *)(pred_d + so + q) = cd;
        *(f4 *)(pred_i + so + q) = ci;
        *(f4 *)(cur_d + so + q) = nd;
        *(f4 *)(cur_i + so + q) = ni;
    }
}

// =============================================================================================
//  host side
// =============================================================================================
struct sf_handle {
    KArgs k{};
    int device = 0;
    int max_blocks = 0;
    int wg_per_cu = 0;
    int *d_order = nullptr;   // KArgs::order storage (more streams than resident workgroups: a launch has a tail)
    int max_blocks_o5 = 0;  // throughput build: resident workgroups of the 5-per-CU kernel (0: not used)
    const FrameVariant *fv = &VARIANTS[0];
    std::vector<struct sf_map *> maps;  // live maps created from this handle: sf_destroy releases their memory and orphans them
    int cluster_grid = 0;  // SF_VARIANT_CLUSTER: blocks per launch (8 XCDs x streams per XCD x workgroups per stream)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evk0 = nullptr, evk1 = nullptr;
    bool solver_timed = false;
    KArgs *d_args = nullptr;  // device copy of k (geometry, parameters, buffer table)
    bool args_dirty = true;
    std::vector<void *> allocs;
    // input stage (sf_input.h), allocated by the first sf_load_frame*
    float depth_cutoff = 4.5f;  // FrontEnd.cpp:168
    bool have_frame = false;
    uint16_t *in_depth_mm = nullptr, *in_filtered_mm = nullptr;
    float *in_depth_metric = nullptr;
    uint8_t *in_color = nullptr;
    uint8_t *stage_color = nullptr;  // one full-resolution frame, for the host-pointer variant
    uint16_t *stage_depth = nullptr;
    size_t stage_px = 0;
    // model prediction (sf_predict.h), allocated by the first sf_predict_from_model
    unsigned long long *pr_keys = nullptr;  // per batched map: low and high key image (2 x n0)
    int *pr_dense = nullptr;                // per batched map: 2 ints (density sum; init-model counts)
    size_t pr_maps = 0;                     // how many maps the two blocks above are sized for
    bool pr_rendered = false;
    std::vector<int> pr_job_of_stream;      // per stream: index of the job of the last predict batch that rendered into it, or -1
    vfloat4 *pr_rays = nullptr;             // view ray per pixel for the intrinsics below (sf_predict_rays_kernel)
    float pr_rays_for[4] = {0.f, 0.f, 0.f, 0.f};
    float *pr_surfels = nullptr;
    size_t pr_floats = 0;                   // capacity of pr_surfels in floats (12 per surfel)
    // argument tables of the batched map kernels (sf_predict.h, sf_fusion.h): device block + the host copy it is filled from
    void *tab_dev = nullptr;
    size_t tab_bytes = 0;
    std::vector<unsigned char> tab_host;
    int *res_dev = nullptr;                 // per batched map: 8 ints of results
    size_t res_maps = 0;
    // overlapped host -> HBM upload of the next frames (sf_upload_current_async): copy stream, staging, event
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done = nullptr, compute_done = nullptr;
    float *up_depth = nullptr, *up_inten = nullptr;
    bool upload_pending = false;
    // sf_advance_sequences_device: per-stream frame numbers, a ring of device + pinned host slots so that calls queue up
    // behind running frame kernels without a host synchronisation (a slot is reused only after its copy has executed)
    static const int SEQ_SLOTS = 8;
    int *seq_index = nullptr;
    int *seq_index_host = nullptr;
    hipEvent_t seq_done[SEQ_SLOTS] = {};
    unsigned seq_calls = 0;
    bool seq_ready = false;  // index ring + events all created
    // multi-frame launches (sf_process_frames / sf_process_sequence_frames_device)
    int *d_frame_done = nullptr;     // [batch]
    int *d_multi_index = nullptr;    // [capacity frames][batch]
    int *h_multi_index = nullptr;    // pinned staging of the same size
    float *d_traj = nullptr;         // [capacity frames][batch][16]
    int multi_capacity = 0;          // frames the three buffers hold
    int solver_timed_frames = 1;     // frames of the launch evk0 / evk1 bracket
};

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(SF_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

template <class T>
static int dev_alloc(sf_handle *h, T **p, size_t count) {
    void *q = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(SF_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemset(q, 0, bytes);
    if (e != hipSuccess) return fail(SF_ERR_DEVICE, std::string("hipMemset: ") + hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = (T *)q;
    return SF_OK;
}

// Grow a device block of the handle to at least `count` elements: geometric growth (a map that gains a few surfels every
// frame must not allocate every frame) and the old block is released -- after the stream has drained, nothing queued still
// reads it -- instead of living on until sf_destroy.
template <class T>
static int dev_grow(sf_handle *h, T **p, size_t *capacity, size_t count) {
    if (*capacity >= count) return SF_OK;
    const size_t want = std::max(count, *capacity + *capacity / 2 + 1024);
    T *old = *p;
    T *fresh = nullptr;
    if (int e = dev_alloc(h, &fresh, want)) return e;
    if (old) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (auto it = h->allocs.begin(); it != h->allocs.end(); ++it)
            if (*it == (void *)old) {
                h->allocs.erase(it);
                break;
            }
        HIP_TRY(hipFree(old));
    }
    *p = fresh;
    *capacity = want;
    return SF_OK;
}

// Cluster launches need every one of their workgroups resident at once (sf_cluster.h): two of them must not share the
// GPU. All cluster launches of the process on one device are therefore chained through an event -- a launch waits (on the
// device) for the previous cluster launch of ANY handle. Kernels of other kinds on other streams are the caller's business:
// if they keep workgroups of a cluster launch from being scheduled, the frame reports SF_STATUS_SYNC_TIMEOUT.
#include <mutex>
static std::mutex g_cluster_mu;
static hipEvent_t g_cluster_done[64] = {};

// throughput build: the 5-workgroups-per-CU compilation of the frame kernel serves the full solver (Makefile: frame_nt256o5.o)
static bool use_five_per_cu(const sf_handle *h) {
    if (!h->max_blocks_o5) return false;
    if (const char *v = std::getenv("SF_THROUGHPUT_WG_PER_CU")) return v[0] == '5';  // pins one of the two (A/B tooling)
    return h->k.p.segmentation_enabled != 0;
}

// One launch of the frame kernel: `n_frames` consecutive frames of every stream (1: the per-call API). ml: the per-launch
// pointers of a multi-frame launch (frame counters, index table, pools, trajectory), or null.
static int launch(sf_handle *h, int mask, int im_count, int n_frames = 1, const FrameLaunch *ml = nullptr) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemsetAsync(h->k.queue, 0, sizeof(int), h->stream));
    if (h->args_dirty) {
        HIP_TRY(hipMemcpyAsync(h->d_args, &h->k, sizeof(KArgs), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->args_dirty = false;
    }
    const long long items = (long long)h->k.batch * n_frames;
    int grid = h->cluster_grid ? h->cluster_grid : (int)std::min<long long>(items, h->max_blocks);
    auto launch_frame = h->fv->launch_frame;
    if (use_five_per_cu(h)) {
        launch_frame = sf_launch_frame_nt256o5;
        grid = (int)std::min<long long>(items, h->max_blocks_o5);
    }
    FrameLaunch fl{};
    if (ml) fl = *ml;
    fl.stage_mask = mask;
    fl.im_count = im_count;
    fl.n_frames = n_frames;
    fl.spin_limit = 1u << 27;
    const bool timed = (mask & ST_SOLVE) != 0;
    if (h->k.order && (mask & ST_SOLVE) && !std::getenv("SF_NO_STREAM_ORDER")) {
        // more streams than resident workgroups: hand the streams out longest-expected-first (their previous frame's IRLS
        // iterations; identical results, a shorter tail when the streams differ)
        hipLaunchKernelGGL(sf_order_kernel, dim3(1), dim3(1024), 0, h->stream, (const sf_frame_stats *)h->k.stats, h->k.batch, h->d_order);
        HIP_TRY(hipGetLastError());
    }
    if (timed) HIP_TRY(hipEventRecord(h->evk0, h->stream));
    if (h->cluster_grid && h->device < 64) {
        std::lock_guard<std::mutex> lock(g_cluster_mu);
        hipEvent_t &ev = g_cluster_done[h->device];
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(h->stream, ev, 0));
        launch_frame(grid, h->stream, (const KArgs *)h->d_args, &fl);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev, h->stream));
    } else {
        launch_frame(grid, h->stream, (const KArgs *)h->d_args, &fl);
        HIP_TRY(hipGetLastError());
    }
    if (timed) {
        HIP_TRY(hipEventRecord(h->evk1, h->stream));
        h->solver_timed = true;
        h->solver_timed_frames = n_frames;
    }
    return SF_OK;
}

static int solve_mask(const sf_handle *h, int create_image_pyr) {
    int m = ST_SOLVE;
    if (create_image_pyr) m |= ST_PYR_NEW;
    if (h->k.p.segmentation_enabled) m |= ST_KMEANS;
    return m;
}

extern "C" {

void sf_ctor_params(sf_params *p) {  // reference FrontEnd.cpp:57-76
    std::memset(p, 0, sizeof(*p));
    p->ctf_levels = 0;
    p->max_iter_per_level = 2;
    p->max_iter_irls = 10;
    p->use_motion_filter = 0;
    p->segmentation_enabled = 1;
    p->debug_planes = 0;
    p->fovh = float(M_PI * 62.5 / 180.0);
    p->k_photometric_res = 0.15f;
    p->irls_delta_threshold = 1e-6f;
    p->previous_speed_const_weight = 0.05f;
    p->previous_speed_eig_weight = 0.5f;
    p->kc_Cauchy = 0.5f;
    p->kb = 1.25f;
    p->kz = 1.5f;
    p->lambda_reg = 0.35f;
    p->lambda_prior = 0.5f;
}

void sf_default_params(sf_params *p) {  // reference StaticFusion-datasets.cpp:79-94
    sf_ctor_params(p);
    p->use_motion_filter = 1;
    p->max_iter_per_level = 3;
    p->previous_speed_const_weight = 0.1f;
    p->previous_speed_eig_weight = 2.f;
    p->k_photometric_res = 0.15f;
    p->irls_delta_threshold = 0.0015f;
    p->max_iter_irls = 6;
    p->lambda_reg = 0.35f;
    p->lambda_prior = 0.5f;
    p->kc_Cauchy = 0.5f;
    p->kb = 1.5f;
    p->kz = 1.5f;
}

const char *sf_last_error(void) { return g_err.c_str(); }
const char *sf_backend(void) { return "hip:gfx950"; }

static int validate_params(const sf_params *p, int levels) {
    // K-means clusters image level 1 (KMeans.cpp:145): a one-level pyramid is only meaningful without segmentation
    if (levels < (p->segmentation_enabled ? 2 : 1) || levels > SF_MAX_LEVELS)
        return fail(SF_ERR_ARG, "ctf_levels must be in [2, 8] (1 is accepted with segmentation_enabled = 0)");
    if (p->max_iter_per_level < 1 || p->max_iter_irls < 1 || levels * p->max_iter_per_level > SF_MAX_OUTER)
        return fail(SF_ERR_ARG, "iteration counts out of range");
    return SF_OK;
}

static void orphan_maps(sf_handle *h);  // defined with struct sf_map below
void sf_destroy(sf_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    orphan_maps(h);  // maps outliving their handle: their memory is freed now, every later call on them fails cleanly
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->evk0) (void)hipEventDestroy(h->evk0);
    if (h->evk1) (void)hipEventDestroy(h->evk1);
    if (h->seq_index_host) (void)hipHostFree(h->seq_index_host);
    if (h->d_multi_index) (void)hipFree(h->d_multi_index);
    if (h->h_multi_index) (void)hipHostFree(h->h_multi_index);
    if (h->d_traj) (void)hipFree(h->d_traj);
    for (auto &e : h->seq_done)
        if (e) (void)hipEventDestroy(e);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->copy_done) (void)hipEventDestroy(h->copy_done);
    if (h->compute_done) (void)hipEventDestroy(h->compute_done);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int sf_create(const sf_params *p, int rows, int cols, int batch, int device, sf_handle **out) {
    return sf_create_ex(p, rows, cols, batch, device, SF_VARIANT_AUTO, out);
}

int sf_get_variant(const sf_handle *h, int *variant, int *threads, int *workgroups_per_stream) {
    if (!h) return fail(SF_ERR_ARG, "null");
    int t = 0, per_cu = 0;
    h->fv->geometry(&t, &per_cu);
    if (variant) *variant = h->fv->id;
    if (threads) *threads = t;
    if (workgroups_per_stream) *workgroups_per_stream = h->k.cluster_g ? h->k.cluster_g : 1;
    return SF_OK;
}

int sf_get_resident_workgroups(const sf_handle *h, int *per_cu, int *total) {
    if (!h) return fail(SF_ERR_ARG, "null");
    const int cus = h->max_blocks / std::max(1, h->wg_per_cu);
    const bool five = use_five_per_cu(h);
    if (per_cu) *per_cu = five ? h->max_blocks_o5 / std::max(1, cus) : h->wg_per_cu;
    if (total) *total = h->cluster_grid ? h->cluster_grid : std::min(h->k.batch, five ? h->max_blocks_o5 : h->max_blocks);
    return SF_OK;
}

int sf_create_ex(const sf_params *p, int rows, int cols, int batch, int device, int variant, sf_handle **out) {
    if (!p || !out || rows < 8 || cols < 8 || batch < 1) return fail(SF_ERR_ARG, "bad argument");
    if (variant < SF_VARIANT_AUTO || variant > SF_VARIANT_CLUSTER) return fail(SF_ERR_ARG, "unknown variant");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(SF_ERR_DEVICE, "no HIP device visible: libsf_hip.so has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(SF_ERR_ARG, "device ordinal out of range");
    if (p->ctf_levels <= 0 && cols < 40) return fail(SF_ERR_ARG, "ctf_levels = 0 (log2(cols/40)+2, FrontEnd.cpp:61) needs cols >= 40");
    int levels = p->ctf_levels > 0 ? p->ctf_levels : int(std::log2(double(cols / 40)) + 2);  // FrontEnd.cpp:61
    if (int e = validate_params(p, levels)) return e;
    if ((rows >> (levels - 1)) < 3 || (cols >> (levels - 1)) < 3)
        return fail(SF_ERR_ARG, "unsupported ctf_levels for this resolution");
    for (int L = 0; L < levels; L++)
        if (((rows >> L) * (cols >> L)) % 4 != 0)
            return fail(SF_ERR_ARG, "every pyramid level must hold a multiple of 4 pixels (vectorised record loads)");

    sf_handle *h = new sf_handle;
    h->device = device;
    KArgs &k = h->k;
    k.rows = rows;
    k.cols = cols;
    k.levels = levels;
    k.batch = batch;
    k.p = *p;
    k.p.ctf_levels = levels;
    k.tan_half_fovh = std::tan(0.5f * p->fovh);  // float overload, as in the reference
    int off = 0;
    for (int L = 0; L < levels; L++) {
        const unsigned s = 1u << L;  // pow(2.f, int(i))
        k.lrows[L] = rows / s;
        k.lcols[L] = cols / s;
        k.ln[L] = k.lrows[L] * k.lcols[L];
        k.loff[L] = off;
        off += k.ln[L];
    }
    k.n_tot = off;
    k.n0 = k.ln[0];

#define TRY_OR_FREE(expr)          \
    do {                           \
        int e_ = (expr);           \
        if (e_ != SF_OK) {         \
            sf_destroy(h);         \
            return e_;             \
        }                          \
    } while (0)
#define HIP_OR_FREE(expr)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            sf_destroy(h);                                                                       \
            return fail(SF_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));       \
        }                                                                                        \
    } while (0)

    HIP_OR_FREE(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_OR_FREE(hipGetDeviceProperties(&prop, device));
    // Few streams: one 1024-thread workgroup per stream and CU gives each stream four times the lanes (1.6-1.8x
    // lower latency, and higher throughput up to ~2 streams per CU); many streams: four 256-thread workgroups per CU.
    // sf_create_ex names the build explicitly; for SF_VARIANT_AUTO the environment variable SF_VARIANT=throughput|latency|cluster
    // may still override the choice by batch size (A/B tooling).
    h->fv = &VARIANTS[(batch <= 2 * prop.multiProcessorCount) ? 1 : 0];
    if (variant == SF_VARIANT_AUTO) {
        if (const char *v = std::getenv("SF_VARIANT")) {
            if (!std::strcmp(v, "throughput")) h->fv = &VARIANTS[0];
            if (!std::strcmp(v, "latency")) h->fv = &VARIANTS[1];
            if (!std::strcmp(v, "cluster")) h->fv = &VARIANTS[2];
        }
    } else {
        h->fv = &VARIANTS[variant == SF_VARIANT_THROUGHPUT ? 0 : (variant == SF_VARIANT_LATENCY ? 1 : 2)];
    }
    int wg_threads = 0, wg_per_cu = 0;
    h->fv->geometry(&wg_threads, &wg_per_cu);
    h->max_blocks = prop.multiProcessorCount * wg_per_cu;
    h->wg_per_cu = wg_per_cu;
    if (h->fv->id == SF_VARIANT_THROUGHPUT) {
        // the full solver runs the same source at 5 workgroups per CU (Makefile: frame_nt256o5.o); SF_THROUGHPUT_WG_PER_CU=4|5
        // pins one of the two for every configuration (A/B tooling)
        int t5 = 0, per_cu5 = 0;
        sf_variant_geometry_nt256o5(&t5, &per_cu5);
        h->max_blocks_o5 = (per_cu5 > wg_per_cu) ? prop.multiProcessorCount * per_cu5 : 0;  // only if a fifth workgroup really fits
        if (std::getenv("SF_DEBUG_GEOMETRY")) std::fprintf(stderr, "sf: throughput build: %d workgroups per CU, full-solver kernel %d\n", wg_per_cu, per_cu5);
    }
    size_t slots = batch;  // record / accumulator slots of n0 pixels
    if (h->fv->id == SF_VARIANT_CLUSTER) {
        // G workgroups (CUs) per stream, all of a launch resident at once: 8 XCDs x (CUs / 8) CUs, the workgroups of a
        // stream on one XCD. Default: as many as fit, at most 24 (one K-means cluster per workgroup; measured best for one QVGA
        // stream: 0.99 ms per frame against 1.02 with 16 or 32); SF_CLUSTER_G overrides.
        const int per_xcd = prop.multiProcessorCount / 8, streams_per_xcd = (batch + 7) / 8;
        int G = std::min(24, per_xcd / streams_per_xcd);
        if (const char *v = std::getenv("SF_CLUSTER_G")) G = std::atoi(v);
        if (G < 1 || G > SF_MAX_CLUSTER || G * streams_per_xcd > per_xcd) {
            sf_destroy(h);
            return fail(SF_ERR_ARG, "SF_VARIANT_CLUSTER: batch too large (or SF_CLUSTER_G out of range): every stream needs its "
                                    "workgroups resident at once, at most CUs / 8 workgroups per XCD");
        }
        k.cluster_g = G;
        k.debug_stall_rank = -1;
        h->cluster_grid = 8 * streams_per_xcd * G;
        slots = (size_t)batch * (1 + G);
    }
    HIP_OR_FREE(hipStreamCreate(&h->own_stream));
    h->stream = h->own_stream;
    HIP_OR_FREE(hipEventCreate(&h->ev0));
    HIP_OR_FREE(hipEventCreate(&h->ev1));
    HIP_OR_FREE(hipEventCreate(&h->evk0));
    HIP_OR_FREE(hipEventCreate(&h->evk1));

    const size_t B = batch, NT = k.n_tot, N0 = k.n0;
    for (int c = 0; c < 2; c++) {  // depth, intensity; xx / yy are recomputed (level_coord), their table entries stay null
        TRY_OR_FREE(dev_alloc(h, &k.pyr_new[c], B * NT));
        TRY_OR_FREE(dev_alloc(h, &k.pyr_pred[c], B * NT));
    }
    if (p->debug_planes)
        for (int c = 0; c < 4; c++) {
            TRY_OR_FREE(dev_alloc(h, &k.dbg_warped[c], B * NT));
            TRY_OR_FREE(dev_alloc(h, &k.dbg_inter[c], B * NT));
        }
    TRY_OR_FREE(dev_alloc(h, &k.labels, B * NT));
    {
        uint8_t *seed = nullptr;  // levels >= 2: level 1 exists (K-means works there)
        const size_t n1 = (k.levels >= 2) ? (size_t)k.ln[1] : 4;
        TRY_OR_FREE(dev_alloc(h, &seed, (n1 + 3) & ~(size_t)3));
        k.km_seed_lab = seed;
    }
    TRY_OR_FREE(dev_alloc(h, &k.acc_d, slots * N0));
    TRY_OR_FREE(dev_alloc(h, &k.acc_i, slots * N0));
    for (int q = 0; q < R_COUNT; q++) TRY_OR_FREE(dev_alloc(h, &k.rec[q], slots * N0));
    TRY_OR_FREE(dev_alloc(h, &k.rec_lab, slots * N0));
    TRY_OR_FREE(dev_alloc(h, &k.rec_null, slots * N0));
    if (k.cluster_g) TRY_OR_FREE(dev_alloc(h, &k.sync, B * 2 * k.cluster_g * SF_SYNC_WORDS));
    TRY_OR_FREE(dev_alloc(h, &k.hist_d, (size_t)SF_HISTORY * B * N0));
    TRY_OR_FREE(dev_alloc(h, &k.hist_i, (size_t)SF_HISTORY * B * N0));
    TRY_OR_FREE(dev_alloc(h, &k.b_img, B * N0));
    TRY_OR_FREE(dev_alloc(h, &k.state, B));
    TRY_OR_FREE(dev_alloc(h, &k.stats, B));
    TRY_OR_FREE(dev_alloc(h, &k.queue, (size_t)1));
    TRY_OR_FREE(dev_alloc(h, &h->d_args, (size_t)1));

    // constructor state (reference FrontEnd.cpp:79-81,110,152-154)
    std::vector<StreamState> st(B);
    std::memset(st.data(), 0, B * sizeof(StreamState));
    for (auto &s : st) {
        for (int q = 0; q < 16; q++) s.T[q] = (q % 5 == 0) ? 1.f : 0.f;
        for (int l = 0; l < SF_NC; l++) {
            s.b_segm[l] = 0.5f;
            s.conn[l] = 1u << l;
            s.cluster_res[l] = std::nanf("");
        }
        for (int i = 0; i < SF_HISTORY; i++)
            for (int q = 0; q < 16; q++) s.hist_T[i][q] = (q % 5 == 0) ? 1.f : 0.f;
        s.kb = p->kb;
    }
    for (size_t b = 0; b < B; b++) st[b].last_slot = (int32_t)b;
    HIP_OR_FREE(hipMemcpy(k.state, st.data(), B * sizeof(StreamState), hipMemcpyHostToDevice));
    if (!k.cluster_g && (int)B > std::max(h->max_blocks, h->max_blocks_o5)) {
        TRY_OR_FREE(dev_alloc(h, &h->d_order, B));
        std::vector<int> iota(B);
        for (size_t i = 0; i < B; i++) iota[i] = (int)i;
        HIP_OR_FREE(hipMemcpy(h->d_order, iota.data(), B * sizeof(int), hipMemcpyHostToDevice));
        k.order = h->d_order;
    }
    if (k.levels >= 2) {
        const int n1 = k.ln[1];
        hipLaunchKernelGGL(sf_seed_label_kernel, dim3((n1 + 255) / 256), dim3(256), 0, 0, const_cast<uint8_t *>(k.km_seed_lab), k.lrows[1], k.lcols[1]);
        HIP_OR_FREE(hipGetLastError());
        HIP_OR_FREE(hipStreamSynchronize(0));
    }
    {
        std::vector<float> half(B * N0, 0.5f);  // b_segm_perpixel.fill(0.5f)
        HIP_OR_FREE(hipMemcpy(k.b_img, half.data(), B * N0 * sizeof(float), hipMemcpyHostToDevice));
    }
    *out = h;
    return SF_OK;
}

int sf_set_params(sf_handle *h, const sf_params *p) {
    if (!h || !p) return fail(SF_ERR_ARG, "null");
    int levels = p->ctf_levels > 0 ? p->ctf_levels : h->k.levels;
    if (levels > h->k.levels) return fail(SF_ERR_ARG, "ctf_levels exceeds the allocated pyramid");
    if (int e = validate_params(p, levels)) return e;
    if (p->debug_planes && !h->k.dbg_warped[0]) return fail(SF_ERR_ARG, "debug_planes must be set at sf_create");
    if (levels != h->k.levels) return fail(SF_ERR_ARG, "ctf_levels cannot change after sf_create");
    const float kb_old = h->k.p.kb;
    h->k.p = *p;
    h->k.p.ctf_levels = levels;
    h->k.tan_half_fovh = std::tan(0.5f * p->fovh);
    h->args_dirty = true;
    if (p->kb != kb_old) return sf_set_kb(h, -1, p->kb);
    return SF_OK;
}
int sf_get_params(const sf_handle *h, sf_params *p) {
    if (!h || !p) return fail(SF_ERR_ARG, "null");
    *p = h->k.p;
    return SF_OK;
}

static int check_stream(const sf_handle *h, int stream) {
    if (!h) return fail(SF_ERR_ARG, "null handle");
    if (stream < 0 || stream >= h->k.batch) return fail(SF_ERR_ARG, "stream out of range");
    return SF_OK;
}

int sf_set_kb(sf_handle *h, int stream, float kb) {
    if (!h || stream < -1 || stream >= h->k.batch) return fail(SF_ERR_ARG, "bad stream");
    HIP_TRY(hipSetDevice(h->device));
    for (int b = 0; b < h->k.batch; b++)
        if (stream < 0 || stream == b)
            HIP_TRY(hipMemcpyAsync(&h->k.state[b].kb, &kb, sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));  // kb is a stack variable
    if (stream < 0) h->k.p.kb = kb;
    return SF_OK;
}
int sf_set_hip_stream(sf_handle *h, void *hip_stream) {
    if (!h) return fail(SF_ERR_ARG, "null");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return SF_OK;
}
int sf_synchronize(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SF_OK;
}

static int upload_pair(sf_handle *h, float *const *set, int stream, const float *depth, const float *intensity) {
    if (int e = check_stream(h, stream)) return e;
    if (!depth || !intensity) return fail(SF_ERR_ARG, "null image");
    HIP_TRY(hipSetDevice(h->device));
    const size_t bytes = sizeof(float) * h->k.n0, o = (size_t)stream * h->k.n_tot;
    HIP_TRY(hipMemcpyAsync(set[0] + o, depth, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(set[1] + o, intensity, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));  // the caller may reuse its buffers on return
    return SF_OK;
}
// depthCurrent / intensityCurrent ARE level 0 of the new pyramid (createImagePyramid copies them there)
int sf_set_current(sf_handle *h, int stream, const float *depth, const float *intensity) {
    return h ? upload_pair(h, h->k.pyr_new, stream, depth, intensity) : fail(SF_ERR_ARG, "null");
}
int sf_set_prediction(sf_handle *h, int stream, const float *depth, const float *intensity) {
    return h ? upload_pair(h, h->k.pyr_pred, stream, depth, intensity) : fail(SF_ERR_ARG, "null");
}
static int copy_batch_device(sf_handle *h, float *const *set, const void *d, const void *i) {
    if (!h || !d || !i) return fail(SF_ERR_ARG, "null");
    HIP_TRY(hipSetDevice(h->device));
    const size_t w = sizeof(float) * h->k.n0;
    HIP_TRY(hipMemcpy2DAsync(set[0], sizeof(float) * h->k.n_tot, d, w, w, h->k.batch, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpy2DAsync(set[1], sizeof(float) * h->k.n_tot, i, w, w, h->k.batch, hipMemcpyDeviceToDevice, h->stream));
    return SF_OK;
}
int sf_set_current_device(sf_handle *h, const void *d, const void *i) {
    return h ? copy_batch_device(h, h->k.pyr_new, d, i) : fail(SF_ERR_ARG, "null");
}
int sf_set_prediction_device(sf_handle *h, const void *d, const void *i) {
    return h ? copy_batch_device(h, h->k.pyr_pred, d, i) : fail(SF_ERR_ARG, "null");
}
int sf_advance_sequences_device(sf_handle *h, const void *pool_depth, const void *pool_intensity, const int32_t *frame_index, int pool_frames) {
    if (!h || !pool_depth || !pool_intensity || !frame_index) return fail(SF_ERR_ARG, "null");
    if (pool_frames < 1) return fail(SF_ERR_ARG, "pool_frames < 1");
    if (h->k.n0 % 4 || h->k.n_tot % 4) return fail(SF_ERR_ARG, "level sizes must be multiples of 4 pixels");
    if (((uintptr_t)pool_depth | (uintptr_t)pool_intensity) & 15u) return fail(SF_ERR_ARG, "the frame pools must be 16-byte aligned (16-byte loads)");
    const size_t B = (size_t)h->k.batch;
    for (size_t b = 0; b < B; b++)
        if (frame_index[b] >= pool_frames) return fail(SF_ERR_ARG, "frame_index entry outside the pool");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->seq_ready) {  // all or nothing: a failure leaves nothing half-initialised behind (the next call starts over)
        if (!h->seq_index)
            if (int e = dev_alloc(h, &h->seq_index, B * sf_handle::SEQ_SLOTS)) return e;
        if (!h->seq_index_host) HIP_TRY(hipHostMalloc((void **)&h->seq_index_host, sizeof(int) * B * sf_handle::SEQ_SLOTS, hipHostMallocDefault));
        for (auto &e : h->seq_done)
            if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->seq_ready = true;
    }
    const unsigned slot = h->seq_calls % sf_handle::SEQ_SLOTS;
    if (h->seq_calls >= (unsigned)sf_handle::SEQ_SLOTS) HIP_TRY(hipEventSynchronize(h->seq_done[slot]));  // eight calls ago
    h->seq_calls++;
    int *host = h->seq_index_host + slot * B, *dev = h->seq_index + slot * B;
    std::memcpy(host, frame_index, sizeof(int) * B);
    HIP_TRY(hipMemcpyAsync(dev, host, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    const dim3 grid((unsigned)std::min(16, (h->k.n0 / 4 + 255) / 256), (unsigned)h->k.batch);
    hipLaunchKernelGGL(sf_advance_kernel, grid, dim3(256), 0, h->stream, h->k.pyr_new[0], h->k.pyr_new[1], h->k.pyr_pred[0], h->k.pyr_pred[1],
                       (const float *)pool_depth, (const float *)pool_intensity, (const int *)dev, h->k.n0, h->k.n_tot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->seq_done[slot], h->stream));
    return SF_OK;
}

// ---- overlapped upload: the next batch of frames crosses PCIe on a second HIP stream while the solver runs ----
int sf_upload_current_async(sf_handle *h, const float *depth_batch, const float *intensity_batch) {
    if (!h || !depth_batch || !intensity_batch) return fail(SF_ERR_ARG, "null");
    if (h->upload_pending) return fail(SF_ERR_STATE, "an upload is already pending: call sf_commit_upload first");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = (size_t)h->k.n0 * h->k.batch;
    if (!h->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->copy_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->compute_done, hipEventDisableTiming));
        if (int e = dev_alloc(h, &h->up_depth, n)) return e;
        if (int e = dev_alloc(h, &h->up_inten, n)) return e;
    }
    // the staging block may still be read by the previous commit's copy on the compute stream
    HIP_TRY(hipStreamWaitEvent(h->copy_stream, h->compute_done, 0));
    HIP_TRY(hipMemcpyAsync(h->up_depth, depth_batch, n * sizeof(float), hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipMemcpyAsync(h->up_inten, intensity_batch, n * sizeof(float), hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipEventRecord(h->copy_done, h->copy_stream));
    h->upload_pending = true;
    return SF_OK;
}
int sf_commit_upload(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (!h->upload_pending) return fail(SF_ERR_STATE, "no upload pending");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamWaitEvent(h->stream, h->copy_done, 0));  // device-side dependency: the host does not block
    if (int e = copy_batch_device(h, h->k.pyr_new, h->up_depth, h->up_inten)) return e;
    HIP_TRY(hipEventRecord(h->compute_done, h->stream));
    h->upload_pending = false;
    return SF_OK;
}
int sf_alloc_pinned(size_t bytes, void **out) {
    if (!out) return fail(SF_ERR_ARG, "null");
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return SF_OK;
}
int sf_free_pinned(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return SF_OK;
}

int sf_current_to_prediction(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    HIP_TRY(hipSetDevice(h->device));
    const size_t w = sizeof(float) * h->k.n0, pitch = sizeof(float) * h->k.n_tot;
    for (int c = 0; c < 2; c++)
        HIP_TRY(hipMemcpy2DAsync(h->k.pyr_pred[c], pitch, h->k.pyr_new[c], pitch, w, h->k.batch, hipMemcpyDeviceToDevice,
                                 h->stream));
    return SF_OK;
}
int sf_set_segm_state(sf_handle *h, int stream, const int32_t *labels0, const float *b_segm, const float *cluster_res) {
    if (int e = check_stream(h, stream)) return e;
    HIP_TRY(hipSetDevice(h->device));
    std::vector<uint8_t> lab;
    if (labels0) {
        lab.resize(h->k.n0);
        for (int q = 0; q < h->k.n0; q++) {
            if (labels0[q] < 0 || labels0[q] > SF_NC) return fail(SF_ERR_ARG, "label out of range");
            lab[q] = (uint8_t)labels0[q];
        }
        HIP_TRY(hipMemcpyAsync(h->k.labels + (size_t)stream * h->k.n_tot, lab.data(), lab.size(), hipMemcpyHostToDevice, h->stream));
    }
    if (b_segm) HIP_TRY(hipMemcpyAsync(h->k.state[stream].b_segm, b_segm, SF_NC * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (cluster_res)
        HIP_TRY(hipMemcpyAsync(h->k.state[stream].cluster_res, cluster_res, SF_NC * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SF_OK;
}
int sf_set_twist_old(sf_handle *h, int stream, const float twist[6]) {
    if (int e = check_stream(h, stream)) return e;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->k.state[stream].twist_old, twist, 6 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SF_OK;
}

int sf_build_pyramid(sf_handle *h, int old_im) {
    if (!h) return fail(SF_ERR_ARG, "null");
    return launch(h, old_im ? ST_PYR_OLD : ST_PYR_NEW, 0);
}
int sf_kmeans(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    return launch(h, ST_KMEANS, 0);
}
int sf_run_solver(sf_handle *h, int create_image_pyr) {
    if (!h) return fail(SF_ERR_ARG, "null");
    return launch(h, solve_mask(h, create_image_pyr), 0);
}
int sf_push_history(sf_handle *h, int im_count) {
    if (!h || im_count < 0) return fail(SF_ERR_ARG, "bad argument");
    return launch(h, ST_PUSH_HISTORY, im_count);
}
int sf_residuals_vs_history(sf_handle *h, int index) {
    if (!h || index < SF_HISTORY) return fail(SF_ERR_ARG, "index must be >= 5");
    return launch(h, ST_RESIDUALS, index);
}
int sf_build_segm_image(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    return launch(h, ST_SEGM_IMAGE, 0);
}
int sf_process_frame(sf_handle *h, int im_count) {
    if (!h || im_count < 0) return fail(SF_ERR_ARG, "bad argument");
    int m = ST_PYR_OLD | solve_mask(h, 1) | ST_SEGM_IMAGE | ST_PUSH_HISTORY;
    if (im_count - SF_HISTORY >= 0) m |= ST_RESIDUALS;
    return launch(h, m, im_count);
}

// ---- several frames per launch ----------------------------------------------------------------
static int multi_buffers(sf_handle *h, int n_frames, bool want_index, bool want_traj) {
    const size_t B = (size_t)h->k.batch;
    if (!h->d_frame_done)
        if (int e = dev_alloc(h, &h->d_frame_done, B)) return e;
    if (n_frames > h->multi_capacity) {  // grow: the old buffers may still be in use by a queued launch
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->d_multi_index) (void)hipFree(h->d_multi_index);
        if (h->h_multi_index) (void)hipHostFree(h->h_multi_index);
        if (h->d_traj) (void)hipFree(h->d_traj);
        h->d_multi_index = nullptr; h->h_multi_index = nullptr; h->d_traj = nullptr; h->multi_capacity = 0;
        HIP_TRY(hipMalloc((void **)&h->d_multi_index, sizeof(int) * B * n_frames));
        HIP_TRY(hipHostMalloc((void **)&h->h_multi_index, sizeof(int) * B * n_frames, hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **)&h->d_traj, sizeof(float) * 16 * B * n_frames));
        h->multi_capacity = n_frames;
    }
    (void)want_index; (void)want_traj;
    return SF_OK;
}
static int process_frames(sf_handle *h, const void *pool_depth, const void *pool_intensity, const int32_t *frame_index, int pool_frames,
                          int im_count0, int n_frames, float *T_out) {
    if (!h || im_count0 < 0 || n_frames < 1 || n_frames > 4096) return fail(SF_ERR_ARG, "bad argument");
    const size_t B = (size_t)h->k.batch;
    const bool seq = pool_depth || pool_intensity || frame_index;
    if (seq) {
        if (!pool_depth || !pool_intensity || !frame_index) return fail(SF_ERR_ARG, "null");
        if (pool_frames < 1) return fail(SF_ERR_ARG, "pool_frames < 1");
        if (h->k.n0 % 4 || h->k.n_tot % 4) return fail(SF_ERR_ARG, "level sizes must be multiples of 4 pixels");
        if (((uintptr_t)pool_depth | (uintptr_t)pool_intensity) & 15u) return fail(SF_ERR_ARG, "the frame pools must be 16-byte aligned (16-byte loads)");
        for (size_t q = 0; q < B * n_frames; q++)
            if (frame_index[q] >= pool_frames) return fail(SF_ERR_ARG, "frame_index entry outside the pool");
    }
    if (h->cluster_grid || n_frames == 1) {
        // the cluster build keeps all workgroups of a stream resident together, one frame per launch: the same calls one by one
        for (int k = 0; k < n_frames; k++) {
            if (seq)
                if (int e = sf_advance_sequences_device(h, pool_depth, pool_intensity, frame_index + (size_t)k * B, pool_frames)) return e;
            if (int e = sf_process_frame(h, im_count0 + k)) return e;
            if (T_out) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                HIP_TRY(hipMemcpy2D(T_out + (size_t)k * B * 16, 16 * sizeof(float), h->k.state, sizeof(StreamState), 16 * sizeof(float), B, hipMemcpyDeviceToHost));
            }
        }
        return SF_OK;
    }
    HIP_TRY(hipSetDevice(h->device));
    if (int e = multi_buffers(h, n_frames, seq, T_out != nullptr)) return e;
    HIP_TRY(hipMemsetAsync(h->d_frame_done, 0, sizeof(int) * B, h->stream));
    FrameLaunch ml{};
    ml.frame_done = h->d_frame_done;
    if (seq) {
        HIP_TRY(hipStreamSynchronize(h->stream));  // the staging block of the previous call has been consumed
        std::memcpy(h->h_multi_index, frame_index, sizeof(int) * B * n_frames);
        HIP_TRY(hipMemcpyAsync(h->d_multi_index, h->h_multi_index, sizeof(int) * B * n_frames, hipMemcpyHostToDevice, h->stream));
        ml.seq_index = h->d_multi_index;
        ml.pool_d = (const float *)pool_depth;
        ml.pool_i = (const float *)pool_intensity;
    }
    if (T_out) ml.traj = h->d_traj;
    const int m = ST_PYR_OLD | solve_mask(h, 1) | ST_SEGM_IMAGE | ST_PUSH_HISTORY | ST_AUTO_RESIDUALS;
    if (int e = launch(h, m, im_count0, n_frames, &ml)) return e;
    if (T_out) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(T_out, h->d_traj, sizeof(float) * 16 * B * n_frames, hipMemcpyDeviceToHost));
    }
    return SF_OK;
}
int sf_process_frames(sf_handle *h, int im_count0, int n_frames, float *T_out) {
    return process_frames(h, nullptr, nullptr, nullptr, 0, im_count0, n_frames, T_out);
}
int sf_process_sequence_frames_device(sf_handle *h, const void *pool_depth, const void *pool_intensity, const int32_t *frame_index, int pool_frames,
                                      int im_count0, int n_frames, float *T_out) {
    if (!pool_depth || !pool_intensity || !frame_index) return fail(SF_ERR_ARG, "null");
    return process_frames(h, pool_depth, pool_intensity, frame_index, pool_frames, im_count0, n_frames, T_out);
}

// ---- getters (synchronise the handle's stream, then copy) ----------------------------------
static int d2h(sf_handle *h, void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return SF_OK;
}
int sf_get_T(sf_handle *h, int stream, float T[16]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, T, h->k.state[stream].T, 16 * sizeof(float));
}
int sf_get_twist(sf_handle *h, int stream, float t[6]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, t, h->k.state[stream].twist, 6 * sizeof(float));
}
int sf_get_twist_old(sf_handle *h, int stream, float t[6]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, t, h->k.state[stream].twist_old, 6 * sizeof(float));
}
int sf_get_b(sf_handle *h, int stream, float b[SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, b, h->k.state[stream].b_segm, SF_NC * sizeof(float));
}
int sf_get_b_image(sf_handle *h, int stream, float *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out) return fail(SF_ERR_ARG, "null");
    return d2h(h, out, h->k.b_img + (size_t)stream * h->k.n0, sizeof(float) * h->k.n0);
}
int sf_get_labels(sf_handle *h, int stream, int level, int32_t *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out || level < 0 || level >= h->k.levels) return fail(SF_ERR_ARG, "bad level");
    std::vector<uint8_t> tmp(h->k.ln[level]);
    if (int e = d2h(h, tmp.data(), h->k.labels + (size_t)stream * h->k.n_tot + h->k.loff[level], tmp.size())) return e;
    for (size_t q = 0; q < tmp.size(); q++) out[q] = tmp[q];
    return SF_OK;
}
int sf_get_kmeans(sf_handle *h, int stream, float c[3 * SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, c, h->k.state[stream].kmeans, 3 * SF_NC * sizeof(float));
}
int sf_get_connectivity(sf_handle *h, int stream, uint8_t conn[SF_NUM_CLUSTERS * SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    uint32_t rows[SF_NC];
    if (int e = d2h(h, rows, h->k.state[stream].conn, sizeof(rows))) return e;
    for (int i = 0; i < SF_NC; i++)
        for (int j = 0; j < SF_NC; j++) conn[i * SF_NC + j] = (rows[i] >> j) & 1u;
    return SF_OK;
}
int sf_get_cluster_residuals(sf_handle *h, int stream, float r[SF_NUM_CLUSTERS]) {
    if (int e = check_stream(h, stream)) return e;
    return d2h(h, r, h->k.state[stream].cluster_res, SF_NC * sizeof(float));
}
int sf_get_stats(sf_handle *h, int stream, sf_frame_stats *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out) return fail(SF_ERR_ARG, "null");
    return d2h(h, out, &h->k.stats[stream], sizeof(sf_frame_stats));
}
int sf_get_batch_results(sf_handle *h, float *T, int32_t *n_irls, int32_t *n_outer, int64_t *pixel_iters) {
    if (!h) return fail(SF_ERR_ARG, "null");
    const int B = h->k.batch;
    if (T) {
        std::vector<StreamState> st(B);
        if (int e = d2h(h, st.data(), h->k.state, B * sizeof(StreamState))) return e;
        for (int b = 0; b < B; b++) std::memcpy(T + 16 * b, st[b].T, 16 * sizeof(float));
    }
    if (n_irls || n_outer || pixel_iters) {
        std::vector<sf_frame_stats> fs(B);
        if (int e = d2h(h, fs.data(), h->k.stats, B * sizeof(sf_frame_stats))) return e;
        for (int b = 0; b < B; b++) {
            if (n_irls) n_irls[b] = fs[b].n_irls;
            if (n_outer) n_outer[b] = fs[b].n_outer;
            if (pixel_iters) pixel_iters[b] = fs[b].pixel_iters;
        }
    }
    return SF_OK;
}

int sf_get_plane(sf_handle *h, int stream, int set, int channel, int level, float *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out || level < 0 || level >= h->k.levels || set < 0 || set > 3 || channel < 0 || channel > 3)
        return fail(SF_ERR_ARG, "bad selector");
    float *const *tab[4] = {h->k.pyr_new, h->k.pyr_pred, h->k.dbg_warped, h->k.dbg_inter};
    const size_t off = (size_t)stream * h->k.n_tot + h->k.loff[level], n = h->k.ln[level];
    if (set <= SF_SET_PRED && channel >= SF_CH_XX) {
        // xx / yy of the pyramids are not stored on the device (every kernel recomputes them from the depth):
        // the same float expression (reference FrontEnd.cpp:385-386) evaluated here
        if (int e = d2h(h, out, tab[set][SF_CH_DEPTH] + off, sizeof(float) * n)) return e;
        const int rows_i = h->k.lrows[level], cols_i = h->k.lcols[level];
        const float inv_f_i = 2.f * h->k.tan_half_fovh / float(cols_i);
        const float disp = (channel == SF_CH_XX) ? 0.5f * (cols_i - 1) : 0.5f * (rows_i - 1);
        for (int u = 0; u < cols_i; u++)
            for (int v = 0; v < rows_i; v++) {
                float &d = out[v + (size_t)u * rows_i];
                d = (inv_f_i * (float(channel == SF_CH_XX ? u : v) - disp)) * d;
            }
        return SF_OK;
    }
    const float *base = tab[set][channel];
    if (!base) return fail(SF_ERR_STATE, "WARPED / INTER planes need params.debug_planes = 1 at sf_create");
    return d2h(h, out, base + off, sizeof(float) * n);
}

int sf_get_jacobian_rows(sf_handle *h, int stream, float *A, float *B, int *n_rows) {
    if (int e = check_stream(h, stream)) return e;
    if (!n_rows) return fail(SF_ERR_ARG, "null");
    if (!h->k.p.debug_planes) return fail(SF_ERR_STATE, "the Jacobian rows need params.debug_planes = 1");
    HIP_TRY(hipSetDevice(h->device));
    StreamState st;
    if (int e = d2h(h, &st, &h->k.state[stream], sizeof(st))) return e;
    const int L = st.last_level;
    if (L < 0 || L >= h->k.levels || st.cum_frames == 0) return fail(SF_ERR_STATE, "no outer iteration executed yet");
    const size_t n = h->k.ln[L];
    float *dev = nullptr;
    HIP_TRY(hipMalloc((void **)&dev, 14 * n * sizeof(float)));
    std::vector<float> planes(14 * n);
    if (h->args_dirty) {
        HIP_TRY(hipMemcpyAsync(h->d_args, &h->k, sizeof(KArgs), hipMemcpyHostToDevice, h->stream));
        h->args_dirty = false;
    }
    h->fv->launch_debug_rows(int((n + 1023) / 1024), h->stream, (const KArgs *)h->d_args, stream, dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(planes.data(), dev, planes.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(SF_ERR_DEVICE, std::string("sf_get_jacobian_rows: ") + hipGetErrorString(e));
    int rows = 0;  // validPixels order of the reference = ascending column-major index (u outer, v inner)
    for (size_t q = 0; q < n; q++) {
        if (std::isnan(planes[q])) continue;
        for (int half = 0; half < 2; half++) {
            if (A)
                for (int c = 0; c < 6; c++) A[(size_t)rows * 6 + c] = planes[(size_t)(7 * half + c) * n + q];
            if (B) B[rows] = planes[(size_t)(7 * half + 6) * n + q];
            rows++;
        }
    }
    *n_rows = rows;
    return SF_OK;
}

int sf_get_lin_plane(sf_handle *h, int stream, int which, float *out, int *rows, int *cols) {
    if (int e = check_stream(h, stream)) return e;
    if (which < 0 || which >= SF_LIN_COUNT) return fail(SF_ERR_ARG, "bad selector");
    StreamState st;
    if (int e = d2h(h, &st, &h->k.state[stream], sizeof(st))) return e;
    const int L = st.last_level;
    if (L < 0 || L >= h->k.levels) return fail(SF_ERR_STATE, "no outer iteration executed yet");
    if (rows) *rows = h->k.lrows[L];
    if (cols) *cols = h->k.lcols[L];
    if (!out) return SF_OK;
    const size_t n = h->k.ln[L], o = (size_t)st.last_slot * h->k.n0;  // the slot the last outer iteration ran on
    if (which == SF_LIN_NULL) {
        if (!h->k.p.debug_planes) return fail(SF_ERR_STATE, "the Null plane needs params.debug_planes = 1");
        std::vector<uint8_t> tmp(n);
        if (int e = d2h(h, tmp.data(), h->k.rec_null + o, n)) return e;
        for (size_t q = 0; q < n; q++) out[q] = tmp[q] ? 1.f : 0.f;
        return SF_OK;
    }
    // dcu..ddv and dct are stored; ddt and the pre-weights are recomputed exactly as the kernels do
    std::vector<float> dn(n), dw(n);
    if (int e = d2h(h, dn.data(), h->k.pyr_new[0] + (size_t)stream * h->k.n_tot + h->k.loff[L], sizeof(float) * n)) return e;
    if (int e = d2h(h, dw.data(), h->k.rec[R_DW] + o, sizeof(float) * n)) return e;
    auto fetch = [&](int plane, std::vector<float> &v) { v.resize(n); return d2h(h, v.data(), h->k.rec[plane] + o, sizeof(float) * n); };
    std::vector<uint8_t> lab(n);  // validPixels: the sign of the stored warped depth (sf_solver.h, linearise)
    for (size_t q = 0; q < n; q++) {
        lab[q] = (dw[q] > 0.f) ? 0 : SF_INVALID_LABEL;
        dw[q] = std::fabs(dw[q]);
    }
    switch (which) {
        case SF_LIN_DCU: return d2h(h, out, h->k.rec[R_DCU] + o, sizeof(float) * n);
        case SF_LIN_DCV: return d2h(h, out, h->k.rec[R_DCV] + o, sizeof(float) * n);
        case SF_LIN_DCT: return d2h(h, out, h->k.rec[R_DCT] + o, sizeof(float) * n);
        case SF_LIN_DDU: return d2h(h, out, h->k.rec[R_DDU] + o, sizeof(float) * n);
        case SF_LIN_DDV: return d2h(h, out, h->k.rec[R_DDV] + o, sizeof(float) * n);
        case SF_LIN_DDT:
            for (size_t q = 0; q < n; q++) out[q] = dn[q] - dw[q];
            return SF_OK;
        default: break;
    }
    std::vector<float> t, gu, gv;
    const bool colour = (which == SF_LIN_WC);
    if (int e = fetch(colour ? R_DCU : R_DDU, gu)) return e;
    if (int e = fetch(colour ? R_DCV : R_DDV, gv)) return e;
    if (colour) {
        if (int e = fetch(R_DCT, t)) return e;
    } else {
        t.resize(n);
        for (size_t q = 0; q < n; q++) t[q] = dn[q] - dw[q];
    }
    for (size_t q = 0; q < n; q++) {
        float w = 0.f;
        if (lab[q] != SF_INVALID_LABEL) {  // weights are 0 outside validPixels (reference :483-484)
            const float err = (colour ? 10.f : 200.f) * (std::fabs(t[q]) + std::fabs(gu[q]) + std::fabs(gv[q]));
            w = std::sqrt(1.f / ((colour ? 1.f : 0.01f) + err));
            w = (colour ? st.inv_max_c : st.inv_max_d) * w;
        }
        out[q] = w;
    }
    return SF_OK;
}

// ---- input stage ------------------------------------------------------------------------------
static int input_alloc(sf_handle *h) {
    if (h->in_depth_mm) return SF_OK;
    const size_t n = (size_t)h->k.n0 * h->k.batch;
    if (int e = dev_alloc(h, &h->in_depth_mm, n)) return e;
    if (int e = dev_alloc(h, &h->in_filtered_mm, n)) return e;
    if (int e = dev_alloc(h, &h->in_depth_metric, n)) return e;
    if (int e = dev_alloc(h, &h->in_color, n * 3)) return e;
    return SF_OK;
}
static int check_full(sf_handle *h, int full_rows, int full_cols, int res) {
    if (res < 1 || full_rows != h->k.rows * res || full_cols != h->k.cols * res)
        return fail(SF_ERR_ARG, "full resolution / res_factor do not match the handle");
    return SF_OK;
}
static int launch_load(sf_handle *h, const uint8_t *d_color, const uint16_t *d_depth, int full_rows, int full_cols, int res,
                       int stream0, int count) {
    const dim3 grid((h->k.cols + LD_T - 1) / LD_T, (h->k.rows + LD_T - 1) / LD_T, count);
    hipLaunchKernelGGL(sf_load_frame_kernel, grid, dim3(256), 0, h->stream, d_color, d_depth, full_cols, (size_t)full_rows * full_cols,
                       res, h->k.rows, h->k.cols, h->k.pyr_new[0], h->k.pyr_new[1], (size_t)h->k.n_tot, h->in_depth_mm, h->in_color,
                       stream0);
    HIP_TRY(hipGetLastError());
    h->have_frame = true;
    return SF_OK;
}
int sf_load_frame(sf_handle *h, int stream, const uint8_t *color_full, const uint16_t *depth_full, int full_rows, int full_cols,
                  int res_factor) {
    if (int e = check_stream(h, stream)) return e;
    if (!color_full || !depth_full) return fail(SF_ERR_ARG, "null image");
    if (int e = check_full(h, full_rows, full_cols, res_factor)) return e;
    HIP_TRY(hipSetDevice(h->device));
    if (int e = input_alloc(h)) return e;
    const size_t px = (size_t)full_rows * full_cols;
    if (h->stage_px < px) {
        if (int e = dev_alloc(h, &h->stage_color, px * 3)) return e;
        if (int e = dev_alloc(h, &h->stage_depth, px)) return e;
        h->stage_px = px;
    }
    HIP_TRY(hipMemcpyAsync(h->stage_color, color_full, px * 3, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->stage_depth, depth_full, px * 2, hipMemcpyHostToDevice, h->stream));
    if (int e = launch_load(h, h->stage_color, h->stage_depth, full_rows, full_cols, res_factor, stream, 1)) return e;
    HIP_TRY(hipStreamSynchronize(h->stream));  // the staging frame is reused by the next call; the host buffers are free again
    return SF_OK;
}
int sf_load_frame_device(sf_handle *h, const void *d_color_full, const void *d_depth_full, int full_rows, int full_cols,
                         int res_factor) {
    if (!h || !d_color_full || !d_depth_full) return fail(SF_ERR_ARG, "null");
    if (int e = check_full(h, full_rows, full_cols, res_factor)) return e;
    HIP_TRY(hipSetDevice(h->device));
    if (int e = input_alloc(h)) return e;
    return launch_load(h, (const uint8_t *)d_color_full, (const uint16_t *)d_depth_full, full_rows, full_cols, res_factor, 0,
                       h->k.batch);
}
int sf_set_depth_cutoff(sf_handle *h, float m) {
    if (!h || !(m > 0.f)) return fail(SF_ERR_ARG, "bad cutoff");
    h->depth_cutoff = m;
    return SF_OK;
}
int sf_filter_depth(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (!h->have_frame) return fail(SF_ERR_STATE, "sf_filter_depth needs sf_load_frame first");
    HIP_TRY(hipSetDevice(h->device));
    const dim3 grid((h->k.cols + BF_TX - 1) / BF_TX, (h->k.rows + BF_TY - 1) / BF_TY, h->k.batch);
    hipLaunchKernelGGL(sf_bilateral_kernel, grid, dim3(256), 0, h->stream, h->in_depth_mm, h->k.rows, h->k.cols, h->depth_cutoff,
                       h->in_filtered_mm, h->in_depth_metric, h->k.pyr_new[0], (size_t)h->k.n_tot);
    HIP_TRY(hipGetLastError());
    return SF_OK;
}
int sf_get_current(sf_handle *h, int stream, float *depth, float *intensity) {
    if (int e = check_stream(h, stream)) return e;
    const size_t bytes = sizeof(float) * h->k.n0;
    if (depth)
        if (int e = d2h(h, depth, h->k.pyr_new[0] + (size_t)stream * h->k.n_tot, bytes)) return e;
    if (intensity)
        if (int e = d2h(h, intensity, h->k.pyr_new[1] + (size_t)stream * h->k.n_tot, bytes)) return e;
    return SF_OK;
}
int sf_get_input_image(sf_handle *h, int stream, int which, void *out) {
    if (int e = check_stream(h, stream)) return e;
    if (!out) return fail(SF_ERR_ARG, "null");
    if (!h->have_frame) return fail(SF_ERR_STATE, "no frame loaded");
    const size_t n = h->k.n0, o = (size_t)stream * n;
    switch (which) {
        case SF_IN_DEPTH_MM: return d2h(h, out, h->in_depth_mm + o, n * 2);
        case SF_IN_DEPTH_FILTERED_MM: return d2h(h, out, h->in_filtered_mm + o, n * 2);
        case SF_IN_DEPTH_METRIC: return d2h(h, out, h->in_depth_metric + o, n * 4);
        case SF_IN_COLOR: return d2h(h, out, h->in_color + o * 3, n * 3);
        default: return fail(SF_ERR_ARG, "bad selector");
    }
}
int sf_timed_input_stage(sf_handle *h, const void *d_color_full, const void *d_depth_full, int full_rows, int full_cols,
                         int res_factor, int calls, float *elapsed_ms) {
    if (!h || calls < 1) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    for (int c = 0; c < calls; c++) {
        if (int e = sf_load_frame_device(h, d_color_full, d_depth_full, full_rows, full_cols, res_factor)) return e;
        if (int e = sf_filter_depth(h)) return e;
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return SF_OK;
}

// ---- frame-to-model prediction -------------------------------------------------------------------
int sf_default_model_params(const sf_handle *h, sf_model_params *p) {
    if (!h || !p) return fail(SF_ERR_ARG, "null");
    const float fovv = float(M_PI * 48.5 / 180.0);                      // FrontEnd.cpp:58
    p->fx = float(0.5 * h->k.cols / std::tan(h->k.p.fovh * 0.5));       // :62 (double arithmetic, then float)
    p->fy = float(0.5 * h->k.rows / std::tan(fovv * 0.5));              // :63
    p->cx = float(h->k.cols / 2);                                        // :165 (integer division)
    p->cy = float(h->k.rows / 2);
    p->max_depth = 20.0f;
    p->conf_low = 0.13f;
    p->conf_high = 0.25f;
    p->time = p->max_time = 0;
    p->time_delta = 2147483647;
    p->extract_max_depth = 4.5f;
    return SF_OK;
}
// 4x4 inverse, double Gauss-Jordan with partial pivoting, rounded to float (the [C5] convention of the solver)
static void invert_pose(const float pose[16], float out[16]) {
    double A[16], Ai[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            A[r * 4 + c] = double(pose[r + 4 * c]);
            Ai[r * 4 + c] = (r == c) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int piv = c;
        double pv = std::fabs(A[c * 4 + c]);
        for (int r = c + 1; r < 4; r++)
            if (std::fabs(A[r * 4 + c]) > pv) {
                pv = std::fabs(A[r * 4 + c]);
                piv = r;
            }
        if (piv != c)
            for (int j = 0; j < 4; j++) {
                std::swap(A[c * 4 + j], A[piv * 4 + j]);
                std::swap(Ai[c * 4 + j], Ai[piv * 4 + j]);
            }
        const double inv = 1.0 / A[c * 4 + c];
        for (int j = 0; j < 4; j++) {
            A[c * 4 + j] *= inv;
            Ai[c * 4 + j] *= inv;
        }
        for (int r = 0; r < 4; r++) {
            if (r == c) continue;
            const double f = A[r * 4 + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 4; j++) {
                A[r * 4 + j] -= f * A[c * 4 + j];
                Ai[r * 4 + j] -= f * Ai[c * 4 + j];
            }
        }
    }
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[r + 4 * c] = float(Ai[r * 4 + c]);
}
// device table for a batched launch: grows on demand, filled from a host copy that lives in the handle
static int upload_table(sf_handle *h, const void *src, size_t bytes, void **dev) {
    if (h->tab_bytes < bytes) {
        unsigned char *q = nullptr;
        if (int e = dev_alloc(h, &q, bytes * 2)) return e;  // the old block is freed with the handle
        h->tab_dev = q;
        h->tab_bytes = bytes * 2;
    }
    h->tab_host.assign((const unsigned char *)src, (const unsigned char *)src + bytes);
    HIP_TRY(hipMemcpyAsync(h->tab_dev, h->tab_host.data(), bytes, hipMemcpyHostToDevice, h->stream));
    *dev = h->tab_dev;
    return SF_OK;
}
static int predict_scratch(sf_handle *h, size_t n_maps) {
    if (h->pr_maps >= n_maps) return SF_OK;
    if (int e = dev_alloc(h, &h->pr_keys, n_maps * 2 * h->k.n0)) return e;
    if (int e = dev_alloc(h, &h->pr_dense, n_maps * 2)) return e;
    h->pr_maps = n_maps;
    return SF_OK;
}
static int results_scratch(sf_handle *h, size_t n_maps) {
    if (h->res_maps >= n_maps) return SF_OK;
    if (int e = dev_alloc(h, &h->res_dev, n_maps * 8)) return e;
    h->res_maps = n_maps;
    return SF_OK;
}
struct PredictJob {
    int stream;
    const float *d_surfels;
    int count;
    const float *pose;
    int time, max_time;
};
// Reconstruction::getPredictedImages for n (stream, surfel buffer, pose) triples in four launches
static int predict_batch(sf_handle *h, const std::vector<PredictJob> &jobs, const sf_model_params *p) {
    const size_t n = h->k.n0;
    if (!(p->conf_low <= p->conf_high)) return fail(SF_ERR_ARG, "conf_low must not exceed conf_high");
    if (jobs.empty()) return SF_OK;
    if (int e = predict_scratch(h, jobs.size())) return e;
    if (!h->pr_rays)
        if (int e = dev_alloc(h, &h->pr_rays, n)) return e;
    if (h->pr_rays_for[0] != p->cx || h->pr_rays_for[1] != p->cy || h->pr_rays_for[2] != p->fx || h->pr_rays_for[3] != p->fy) {
        hipLaunchKernelGGL(sf_predict_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->pr_rays, h->k.rows, h->k.cols, p->cx, p->cy,
                           p->fx, p->fy);
        h->pr_rays_for[0] = p->cx; h->pr_rays_for[1] = p->cy; h->pr_rays_for[2] = p->fx; h->pr_rays_for[3] = p->fy;
    }
    std::vector<PredictArgs> tab(jobs.size());
    int max_count = 0;
    for (size_t q = 0; q < jobs.size(); q++) {
        const PredictJob &j = jobs[q];
        PredictArgs &a = tab[q];
        a.surfels = j.d_surfels;
        a.count = j.count;
        max_count = std::max(max_count, j.count);
        invert_pose(j.pose, a.t_inv);  // t_inv = pose.inverse() (IndexMap.cpp:251)
        a.cx = p->cx; a.cy = p->cy; a.fx = p->fx; a.fy = p->fy;
        a.max_depth = p->max_depth; a.conf_low = p->conf_low; a.conf_high = p->conf_high; a.extract_max_depth = p->extract_max_depth;
        a.time = j.time; a.max_time = j.max_time; a.time_delta = p->time_delta;
        a.rows = h->k.rows; a.cols = h->k.cols;
        a.key_low = h->pr_keys + q * 2 * n; a.key_high = a.key_low + n; a.dense_count = h->pr_dense + q * 2;
        a.filtered_mm = h->in_filtered_mm + (size_t)j.stream * n;
        a.color = h->in_color + (size_t)j.stream * n * 3;
        a.b_img = h->k.b_img + (size_t)j.stream * n;
        a.depth_pred = h->k.pyr_pred[0] + (size_t)j.stream * h->k.n_tot;
        a.inten_pred = h->k.pyr_pred[1] + (size_t)j.stream * h->k.n_tot;
        a.rays = h->pr_rays;
    }
    void *dev = nullptr;
    if (int e = upload_table(h, tab.data(), tab.size() * sizeof(PredictArgs), &dev)) return e;
    const PredictArgs *d_tab = (const PredictArgs *)dev;
    const unsigned nm = (unsigned)jobs.size();
    const unsigned pix_blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(sf_predict_clear_kernel, dim3(pix_blocks, nm), dim3(256), 0, h->stream, d_tab);
    if (max_count) hipLaunchKernelGGL(sf_predict_splat_kernel, dim3((max_count + SF_SPLAT_NT - 1) / SF_SPLAT_NT, nm), dim3(SF_SPLAT_NT), 0, h->stream, d_tab);
    hipLaunchKernelGGL(sf_predict_dense_kernel, dim3(nm), dim3(64), 0, h->stream, d_tab);
    hipLaunchKernelGGL(sf_predict_resolve_kernel, dim3(pix_blocks, nm), dim3(256), 0, h->stream, d_tab);
    HIP_TRY(hipGetLastError());
    h->pr_rendered = true;
    // the density sums of this batch live in pr_dense[2 q] until the next prediction call: which job served which stream
    h->pr_job_of_stream.assign((size_t)h->k.batch, -1);
    for (size_t q = 0; q < jobs.size(); q++) h->pr_job_of_stream[(size_t)jobs[q].stream] = (int)q;
    return SF_OK;
}
static int predict_launch(sf_handle *h, int stream, const float *d_surfels, int count, const float pose[16], const sf_model_params *p) {
    return predict_batch(h, std::vector<PredictJob>{PredictJob{stream, d_surfels, count, pose, p->time, p->max_time}}, p);
}
int sf_predict_from_model(sf_handle *h, int stream, const float *surfels, int count, const float pose[16], const sf_model_params *p) {
    if (int e = check_stream(h, stream)) return e;
    if ((!surfels && count > 0) || count < 0 || !pose || !p) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (int e = input_alloc(h)) return e;
    if (int e = dev_grow(h, &h->pr_surfels, &h->pr_floats, (size_t)count * 12)) return e;
    if (count) HIP_TRY(hipMemcpyAsync(h->pr_surfels, surfels, (size_t)count * 12 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (int e = predict_launch(h, stream, h->pr_surfels, count, pose, p)) return e;
    HIP_TRY(hipStreamSynchronize(h->stream));  // the host surfel buffer is free again; the staging block may be reused
    return SF_OK;
}
int sf_predict_from_model_device(sf_handle *h, int stream, const void *d_surfels, int count, const float pose[16],
                                 const sf_model_params *p) {
    if (int e = check_stream(h, stream)) return e;
    if ((!d_surfels && count > 0) || count < 0 || !pose || !p) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (int e = input_alloc(h)) return e;
    return predict_launch(h, stream, (const float *)d_surfels, count, pose, p);
}
// GlobalModel::initialise for n maps: zero-fill (the feedback buffers start zero-filled), the two ordered compactions, trim
static void init_model_launch(sf_handle *h, const InitModelArgs *d_tab, int n_maps);
static int init_model_batch(sf_handle *h, const InitModelArgs *args, int n_maps) {
    void *dev = nullptr;
    if (int e = upload_table(h, args, (size_t)n_maps * sizeof(InitModelArgs), &dev)) return e;
    init_model_launch(h, (const InitModelArgs *)dev, n_maps);
    HIP_TRY(hipGetLastError());
    return SF_OK;
}
static void init_model_launch(sf_handle *h, const InitModelArgs *d_tab, int n_maps) {
    const unsigned blocks = (unsigned)((h->k.n0 * 12 + 255) / 256);
    hipLaunchKernelGGL(sf_init_model_zero_kernel, dim3(blocks, n_maps), dim3(256), 0, h->stream, d_tab);
    hipLaunchKernelGGL(sf_init_model_kernel, dim3(n_maps), dim3(1024), 0, h->stream, d_tab);
    hipLaunchKernelGGL(sf_init_model_trim_kernel, dim3(blocks, n_maps), dim3(256), 0, h->stream, d_tab);
}
int sf_init_model_from_frame(sf_handle *h, int stream, const float pose[16], const sf_model_params *p, int time, float *surfels_out,
                             int *count) {
    if (int e = check_stream(h, stream)) return e;
    if (!pose || !p || !surfels_out || !count) return fail(SF_ERR_ARG, "null");
    if (!h->have_frame) return fail(SF_ERR_STATE, "sf_init_model_from_frame needs a loaded frame (sf_load_frame + sf_filter_depth)");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n = h->k.n0;
    if (int e = dev_grow(h, &h->pr_surfels, &h->pr_floats, n * 12)) return e;
    if (int e = results_scratch(h, 1)) return e;
    InitModelArgs a;
    a.depth_metric = h->in_depth_metric + (size_t)stream * n;
    a.depth_filtered = h->k.pyr_new[0] + (size_t)stream * h->k.n_tot;
    a.color = h->in_color + (size_t)stream * n * 3;
    a.b_img = h->k.b_img + (size_t)stream * n;
    a.rows = h->k.rows; a.cols = h->k.cols; a.time = time;
    for (int q = 0; q < 16; q++) a.pose[q] = pose[q];
    a.cx = p->cx; a.cy = p->cy; a.fx = p->fx; a.fy = p->fy; a.max_depth = p->max_depth;
    a.out = h->pr_surfels;
    a.count = h->res_dev;
    if (int e = init_model_batch(h, &a, 1)) return e;
    int counts[2] = {0, 0};
    if (int e = d2h(h, counts, h->res_dev, sizeof counts)) return e;
    if (int e = d2h(h, surfels_out, h->pr_surfels, n * 12 * sizeof(float))) return e;
    *count = counts[0];
    return SF_OK;
}
int sf_get_prediction_dense(sf_handle *h, int *dense) {
    if (!h || !dense) return fail(SF_ERR_ARG, "null");
    *dense = 0;
    if (!h->pr_rendered) return SF_OK;  // nothing rendered yet
    int sum = 0;
    if (int e = d2h(h, &sum, h->pr_dense, sizeof sum)) return e;
    const int rw = h->k.cols / 40, rh = h->k.rows / 40;
    *dense = (rw * rh > 0) && (float(sum) / float(rh * rw) > 0.25f);
    return SF_OK;
}
int sf_get_prediction_dense_stream(sf_handle *h, int stream, int *dense) {
    if (int e = check_stream(h, stream)) return e;
    if (!dense) return fail(SF_ERR_ARG, "null");
    *dense = 0;
    if (!h->pr_rendered || h->pr_job_of_stream.empty()) return SF_OK;
    const int q = h->pr_job_of_stream[(size_t)stream];
    if (q < 0) return SF_OK;  // not part of the last prediction call
    int sum = 0;
    if (int e = d2h(h, &sum, h->pr_dense + (size_t)q * 2, sizeof sum)) return e;
    const int rw = h->k.cols / 40, rh = h->k.rows / 40;
    *dense = (rw * rh > 0) && (float(sum) / float(rh * rw) > 0.25f);
    return SF_OK;
}
int sf_get_prediction(sf_handle *h, int stream, float *depth, float *intensity) {
    if (int e = check_stream(h, stream)) return e;
    const size_t bytes = sizeof(float) * h->k.n0;
    if (depth)
        if (int e = d2h(h, depth, h->k.pyr_pred[0] + (size_t)stream * h->k.n_tot, bytes)) return e;
    if (intensity)
        if (int e = d2h(h, intensity, h->k.pyr_pred[1] + (size_t)stream * h->k.n_tot, bytes)) return e;
    return SF_OK;
}

// ---- the surfel map (sf_fusion.h) ------------------------------------------------------------------
struct sf_map {
    sf_handle *h = nullptr;
    int capacity = 0;
    float *buf[2] = {nullptr, nullptr};  // the model lives in buf[0] between calls; buf[1] holds the merged model inside a fuse
    int count = 0, tick = 1;
    float pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int stats[4] = {0, 0, 0, 0};
    unsigned long long *keys = nullptr, *occ = nullptr;
    unsigned *winner = nullptr, *meta = nullptr, *index_export = nullptr;
    float *rec = nullptr;
    unsigned char *flags = nullptr;
    int *block_counts = nullptr;
    bool have_index = false;
    int epoch = 0;  // index images rendered since the key image was last filled with ones; tag = 255 - epoch
    std::vector<void *> allocs;
};
static int map_alloc_bytes(sf_map *m, void **p, size_t bytes) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(SF_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    m->allocs.push_back(q);
    *p = q;
    return SF_OK;
}
#define map_alloc(m, p, count) map_alloc_bytes(m, (void **)(p), (size_t)(count) * sizeof(**(p)))
int sf_map_create(sf_handle *h, int capacity, sf_map **out) {
    if (!h || !out || capacity < 0) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t n0 = h->k.n0;
    const size_t cap = capacity ? (size_t)capacity : (size_t)3072 * 3072;  // GlobalModel.cpp:21-22
    if (cap < n0) return fail(SF_ERR_ARG, "capacity below rows * cols (the first frame alone can need that many surfels)");
    sf_map *m = new sf_map;
    m->h = h;
    m->capacity = (int)cap;
    const size_t n_cand_max = (size_t)((h->k.rows + 1) / 2) * ((h->k.cols + 1) / 2);
    int e = SF_OK;
    if (!e) e = map_alloc(m, &m->buf[0], cap * 12);
    if (!e) e = map_alloc(m, &m->buf[1], cap * 12);
    if (!e) e = map_alloc(m, &m->keys, n0 * 16);
    if (!e && hipMemset(m->keys, 0xff, n0 * 16 * sizeof(unsigned long long)) != hipSuccess) e = fail(SF_ERR_DEVICE, "hipMemset");
    if (!e) e = map_alloc(m, &m->occ, (size_t)h->k.cols * 4 * ((h->k.rows * 4 + 63) / 64));
    if (!e) e = map_alloc(m, &m->index_export, n0 * 16);
    if (!e) e = map_alloc(m, &m->winner, cap);
    if (!e) e = map_alloc(m, &m->rec, n_cand_max * 12);
    if (!e) e = map_alloc(m, &m->meta, n_cand_max * 2);
    if (!e) e = map_alloc(m, &m->flags, cap + n_cand_max);
    if (!e) e = map_alloc(m, &m->block_counts, (cap + n_cand_max + SF_CLEAN_BLOCK - 1) / SF_CLEAN_BLOCK + 1);
    if (e) {
        sf_map_destroy(m);
        return e;
    }
    h->maps.push_back(m);
    *out = m;
    return SF_OK;
}
static void map_release(sf_map *m) {  // device memory of a map (its handle's device is current, its stream drained)
    for (void *q : m->allocs) (void)hipFree(q);
    m->allocs.clear();
}
static void orphan_maps(sf_handle *h) {
    for (sf_map *m : h->maps) {
        map_release(m);
        m->h = nullptr;
    }
    h->maps.clear();
}
void sf_map_destroy(sf_map *m) {
    if (!m) return;
    if (sf_handle *h = m->h) {  // the handle is alive: its device, after its queued work
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        for (auto it = h->maps.begin(); it != h->maps.end(); ++it)
            if (*it == m) {
                h->maps.erase(it);
                break;
            }
        map_release(m);
    }  // else: sf_destroy of the handle already released the memory and left the map as an empty shell
    delete m;
}
static void pose_compose(const float *a, const float *b, float *out) {  // Eigen::Matrix4f product, column-major
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++) {
            float acc = a[rr] * b[4 * c];
            for (int k = 1; k < 4; k++) acc = acc + a[rr + 4 * k] * b[k + 4 * c];
            r[rr + 4 * c] = acc;
        }
    std::memcpy(out, r, sizeof r);
}
// Reconstruction::fuseFrame for n (stream, map) pairs: at most 3 + 9 launches and one read-back for the whole batch
int sf_map_fuse_frames(sf_handle *h, int n, const int *streams, sf_map *const *maps, const float *in_poses, float weight_multiplier,
                       const sf_model_params *p) {
    if (!h || n < 0 || (n && (!streams || !maps)) || !p) return fail(SF_ERR_ARG, "bad argument");
    if (n == 0) return SF_OK;
    if (!h->have_frame) return fail(SF_ERR_STATE, "sf_map_fuse_frame needs a loaded frame (sf_load_frame + sf_filter_depth)");
    for (int q = 0; q < n; q++) {
        if (int e = check_stream(h, streams[q])) return e;
        if (!maps[q] || maps[q]->h != h) return fail(SF_ERR_ARG, "a map belongs to the handle it was created from");
        if (!in_poses && maps[q]->tick != 1) return fail(SF_ERR_ARG, "in_pose may be NULL on the first fuse only");
        for (int r = 0; r < q; r++)
            if (maps[r] == maps[q]) return fail(SF_ERR_ARG, "the same map twice in one batch");
    }
    HIP_TRY(hipSetDevice(h->device));
    if (int e = results_scratch(h, (size_t)n)) return e;
    const size_t npx = h->k.n0;
    std::vector<InitModelArgs> init;
    std::vector<FuseArgs> fuse;
    std::vector<int> init_of, fuse_of;  // batch index of each table entry
    // the maps' new poses / epochs are held here and committed together with count and tick only after the results have
    // been read back: a failed upload, launch or copy leaves every map as it was (a retry must not compose in_pose twice)
    std::vector<float> new_pose((size_t)n * 16);
    std::vector<int> new_epoch((size_t)n);
    int max_count = 0, max_cand = 0, max_elems = 0;
    for (int q = 0; q < n; q++) {
        sf_map *m = maps[q];
        const int stream = streams[q];
        const float *in_pose = in_poses ? in_poses + (size_t)q * 16 : nullptr;
        const float *depth_metric = h->in_depth_metric + (size_t)stream * npx;
        const float *depth_filtered = h->k.pyr_new[0] + (size_t)stream * h->k.n_tot;
        const uint8_t *color = h->in_color + (size_t)stream * npx * 3;
        const float *b_img = h->k.b_img + (size_t)stream * npx;
        float *pose_q = new_pose.data() + (size_t)q * 16;
        std::memcpy(pose_q, m->pose, sizeof m->pose);
        new_epoch[q] = m->epoch;
        if (m->tick == 1) {  // Reconstruction.cpp:255-262
            if (in_pose) pose_compose(m->pose, in_pose, pose_q);
            InitModelArgs a;
            a.depth_metric = depth_metric; a.depth_filtered = depth_filtered; a.color = color; a.b_img = b_img;
            a.rows = h->k.rows; a.cols = h->k.cols; a.time = m->tick;
            for (int k = 0; k < 16; k++) a.pose[k] = pose_q[k];
            a.cx = p->cx; a.cy = p->cy; a.fx = p->fx; a.fy = p->fy; a.max_depth = p->max_depth;
            a.out = m->buf[0];
            a.count = h->res_dev + (size_t)q * 8;
            init.push_back(a);
            init_of.push_back(q);
            continue;
        }
        float last_pose[16];
        std::memcpy(last_pose, m->pose, sizeof last_pose);
        pose_compose(m->pose, in_pose, pose_q);                                         // :268
        FuseArgs a;
        a.depth_metric = depth_metric; a.depth_filtered = depth_filtered; a.color = color; a.b_img = b_img;
        a.rows = h->k.rows; a.cols = h->k.cols;
        for (int k = 0; k < 16; k++) a.pose[k] = pose_q[k];
        invert_pose(pose_q, a.t_inv);
        a.cx = p->cx; a.cy = p->cy; a.fx = p->fx; a.fy = p->fy;
        a.camz = float(1.0 / double(p->fx)); a.camw = float(1.0 / double(p->fy));       // GlobalModel.cpp:365-368
        a.max_depth = p->max_depth; a.conf_threshold = p->conf_high;
        a.weighting = sf_fusion_weighting(last_pose, pose_q, weight_multiplier);         // :270-282
        a.time = m->tick; a.time_delta = p->time_delta;
        a.src = m->buf[0]; a.dst = m->buf[1]; a.out = m->buf[0];
        a.count = m->count; a.capacity = m->capacity;
        a.keys = m->keys; a.winner = m->winner;
        a.occ = m->occ; a.occ_words = (a.rows * 4 + 63) / 64;
        if (new_epoch[q] + 2 > 255) {  // the 8-bit tag is used up: one real clear, then count again
            HIP_TRY(hipMemsetAsync(m->keys, 0xff, npx * 16 * sizeof(unsigned long long), h->stream));
            m->epoch = new_epoch[q] = 0;  // the key image IS cleared from here on, whatever happens next
        }
        a.tag_first = 255u - (unsigned)(new_epoch[q] + 1); a.tag_merged = 255u - (unsigned)(new_epoch[q] + 2);
        new_epoch[q] += 2;
        a.par = m->tick % 2;
        a.cand_rows = (a.rows - a.par + 1) / 2; a.cand_cols = (a.cols - a.par + 1) / 2;
        a.n_cand = a.cand_rows * a.cand_cols;
        a.rec = m->rec; a.meta = m->meta; a.flags = m->flags; a.block_counts = m->block_counts;
        a.result = h->res_dev + (size_t)q * 8;
        max_count = std::max(max_count, a.count);
        max_cand = std::max(max_cand, a.n_cand);
        max_elems = std::max(max_elems, a.count + a.n_cand);
        fuse.push_back(a);
        fuse_of.push_back(q);
    }
    // one upload: [init table | fuse table]
    const size_t init_bytes = (init.size() * sizeof(InitModelArgs) + 255) / 256 * 256;
    std::vector<unsigned char> blob(init_bytes + fuse.size() * sizeof(FuseArgs));
    if (!init.empty()) std::memcpy(blob.data(), init.data(), init.size() * sizeof(InitModelArgs));
    if (!fuse.empty()) std::memcpy(blob.data() + init_bytes, fuse.data(), fuse.size() * sizeof(FuseArgs));
    void *dev = nullptr;
    if (int e = upload_table(h, blob.data(), blob.size(), &dev)) return e;
    if (!init.empty()) init_model_launch(h, (const InitModelArgs *)dev, (int)init.size());
    if (!fuse.empty()) {
        const FuseArgs *tab = (const FuseArgs *)((const unsigned char *)dev + init_bytes);
        const unsigned nm = (unsigned)fuse.size();
        const unsigned surfel_blocks = (unsigned)((max_count + 255) / 256);
        const unsigned occ_blocks = (unsigned)(((size_t)h->k.cols * 4 * ((h->k.rows * 4 + 63) / 64) + 255) / 256);
        const unsigned begin_blocks = std::max(occ_blocks, surfel_blocks);
        const unsigned clean_blocks = (unsigned)((max_elems + SF_CLEAN_BLOCK - 1) / SF_CLEAN_BLOCK);
        hipLaunchKernelGGL(sf_fuse_begin_kernel, dim3(begin_blocks, nm), dim3(256), 0, h->stream, tab);                        // :284
        if (max_count) hipLaunchKernelGGL(sf_index_splat_kernel, dim3(surfel_blocks, nm), dim3(256), 0, h->stream, tab);
        if (max_cand) hipLaunchKernelGGL(sf_fuse_data_kernel, dim3((max_cand + 63) / 64, nm), dim3(64), 0, h->stream, tab);   // :286-298
        hipLaunchKernelGGL(sf_index_clear_kernel, dim3(occ_blocks, nm), dim3(256), 0, h->stream, tab);                        // :300
        if (max_count) hipLaunchKernelGGL(sf_fuse_update_kernel, dim3(surfel_blocks, nm), dim3(256), 0, h->stream, tab);       // merge + index image of the result
        if (clean_blocks) {                                                                                                    // :302-311
            hipLaunchKernelGGL(sf_clean_flag_kernel, dim3(clean_blocks, nm), dim3(SF_CLEAN_BLOCK), 0, h->stream, tab);
            hipLaunchKernelGGL(sf_clean_scan_kernel, dim3(nm), dim3(1024), 0, h->stream, tab);
            hipLaunchKernelGGL(sf_clean_write_kernel, dim3(clean_blocks, nm), dim3(SF_CLEAN_BLOCK), 0, h->stream, tab);
        }
    }
    HIP_TRY(hipGetLastError());
    std::vector<int> res((size_t)n * 8);
    if (int e = d2h(h, res.data(), h->res_dev, res.size() * sizeof(int))) return e;
    int overflow = -1;
    for (int q = 0; q < n; q++) {  // commit
        std::memcpy(maps[q]->pose, new_pose.data() + (size_t)q * 16, sizeof maps[q]->pose);
        maps[q]->epoch = new_epoch[q];
    }
    for (int q : init_of) {
        sf_map *m = maps[q];
        m->count = res[(size_t)q * 8];
        m->stats[0] = m->stats[1] = m->stats[2] = 0;
        m->stats[3] = m->count;
        m->tick++;
    }
    for (int q : fuse_of) {
        sf_map *m = maps[q];
        const int *r = res.data() + (size_t)q * 8;
        m->count = r[0];
        m->stats[0] = r[2]; m->stats[1] = r[3]; m->stats[2] = r[4]; m->stats[3] = r[0];
        m->have_index = true;
        m->tick++;
        if (r[1] > m->capacity && overflow < 0) overflow = q;
    }
    if (overflow >= 0) return fail(SF_ERR_STATE, "surfel map capacity exceeded (truncated): batch entry " + std::to_string(overflow));
    return SF_OK;
}
int sf_map_fuse_frame(sf_handle *h, int stream, sf_map *m, const float *in_pose, float weight_multiplier, const sf_model_params *p) {
    return sf_map_fuse_frames(h, 1, &stream, &m, in_pose, weight_multiplier, p);
}
// Reconstruction::getPredictedImages for n (stream, map) pairs at each map's currPose and tick, in four launches
int sf_map_predict_frames(sf_handle *h, int n, const int *streams, sf_map *const *maps, const sf_model_params *p) {
    if (!h || n < 0 || (n && (!streams || !maps)) || !p) return fail(SF_ERR_ARG, "bad argument");
    std::vector<PredictJob> jobs((size_t)n);
    for (int q = 0; q < n; q++) {
        if (int e = check_stream(h, streams[q])) return e;
        if (!maps[q] || maps[q]->h != h) return fail(SF_ERR_ARG, "a map belongs to the handle it was created from");
        for (int r = 0; r < q; r++)
            if (streams[r] == streams[q]) return fail(SF_ERR_ARG, "the same stream twice in one batch (its prediction would be written twice)");
        jobs[(size_t)q] = PredictJob{streams[q], maps[q]->buf[0], maps[q]->count, maps[q]->pose, maps[q]->tick, maps[q]->tick};
    }
    HIP_TRY(hipSetDevice(h->device));
    if (int e = input_alloc(h)) return e;
    return predict_batch(h, jobs, p);
}
int sf_map_predict(sf_handle *h, int stream, sf_map *m, const sf_model_params *p) {
    return sf_map_predict_frames(h, 1, &stream, &m, p);
}
int sf_map_info(sf_map *m, int *count, int *tick, float pose[16], int stats[4]) {
    if (!m) return fail(SF_ERR_ARG, "null");
    if (count) *count = m->count;
    if (tick) *tick = m->tick;
    if (pose) std::memcpy(pose, m->pose, sizeof m->pose);
    if (stats) std::memcpy(stats, m->stats, sizeof m->stats);
    return SF_OK;
}
int sf_map_download(sf_map *m, float *surfels, int max_count) {
    if (!m || (!surfels && max_count > 0) || max_count < 0) return fail(SF_ERR_ARG, "bad argument");
    if (!m->h) return fail(SF_ERR_STATE, "the handle this map was created from has been destroyed");
    const size_t k = (size_t)std::min(max_count, m->count);
    if (k) return d2h(m->h, surfels, m->buf[0], k * 12 * sizeof(float));
    return SF_OK;
}
int sf_map_upload(sf_map *m, const float *surfels, int count, const float pose[16], int tick) {
    if (!m || (!surfels && count > 0) || count < 0 || !pose || tick < 1) return fail(SF_ERR_ARG, "bad argument");
    if (count > m->capacity) return fail(SF_ERR_ARG, "count exceeds the map's capacity");
    if (!m->h) return fail(SF_ERR_STATE, "the handle this map was created from has been destroyed");
    HIP_TRY(hipSetDevice(m->h->device));
    if (count) {
        HIP_TRY(hipMemcpyAsync(m->buf[0], surfels, (size_t)count * 12 * sizeof(float), hipMemcpyHostToDevice, m->h->stream));
        HIP_TRY(hipStreamSynchronize(m->h->stream));
    }
    m->count = count;
    std::memcpy(m->pose, pose, sizeof m->pose);
    m->tick = tick;
    return SF_OK;
}
int sf_map_get_index_map(sf_map *m, uint32_t *out) {
    if (!m || !out) return fail(SF_ERR_ARG, "null");
    if (!m->h) return fail(SF_ERR_STATE, "the handle this map was created from has been destroyed");
    if (!m->have_index) return fail(SF_ERR_STATE, "no index map yet (sf_map_fuse_frame with tick > 1 renders it)");
    HIP_TRY(hipSetDevice(m->h->device));
    const size_t n = m->h->k.n0 * 16;
    hipLaunchKernelGGL(sf_index_export_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->h->stream, m->keys, m->occ,
                       (m->h->k.rows * 4 + 63) / 64, m->index_export, m->h->k.cols * 4, m->h->k.rows * 4);
    HIP_TRY(hipGetLastError());
    return d2h(m->h, out, m->index_export, n * sizeof(uint32_t));
}

int sf_level_rows(const sf_handle *h, int level) { return (h && level >= 0 && level < h->k.levels) ? h->k.lrows[level] : 0; }
int sf_level_cols(const sf_handle *h, int level) { return (h && level >= 0 && level < h->k.levels) ? h->k.lcols[level] : 0; }
int sf_batch(const sf_handle *h) { return h ? h->k.batch : 0; }

int sf_timed_process_frames(sf_handle *h, int im_count, int calls, float *elapsed_ms) {
    if (!h || calls < 1) return fail(SF_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    if (std::getenv("SF_TIMED_LAUNCH_PER_FRAME")) {  // A/B: one launch per frame, as before multi-frame launches existed
        for (int c = 0; c < calls; c++)
            if (int e = sf_process_frame(h, im_count + c)) return e;
    } else if (int e = sf_process_frames(h, im_count, calls, nullptr)) {
        return e;
    }
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return SF_OK;
}
int sf_get_counters(sf_handle *h, int64_t *frames, int64_t *n_irls, int64_t *n_outer, int64_t *pixel_iters) {
    if (!h) return fail(SF_ERR_ARG, "null");
    std::vector<StreamState> st(h->k.batch);
    if (int e = d2h(h, st.data(), h->k.state, st.size() * sizeof(StreamState))) return e;
    long long f = 0, i = 0, o = 0, p = 0;
    for (auto &s : st) {
        f += s.cum_frames;
        i += s.cum_irls;
        o += s.cum_outer;
        p += s.cum_pixel_iters;
    }
    if (frames) *frames = f;
    if (n_irls) *n_irls = i;
    if (n_outer) *n_outer = o;
    if (pixel_iters) *pixel_iters = p;
    return SF_OK;
}
int sf_get_stage_profile(sf_handle *h, int64_t ticks[32]) {
    if (!h || !ticks) return fail(SF_ERR_ARG, "null");
    std::vector<StreamState> st(h->k.batch);
    if (int e = d2h(h, st.data(), h->k.state, st.size() * sizeof(StreamState))) return e;
    for (int q = 0; q < SF_PROF_SLOTS; q++) ticks[q] = 0;
    for (auto &s : st)
        for (int q = 0; q < SF_PROF_SLOTS; q++) ticks[q] += s.prof[q];
    return SF_OK;
}
int sf_microbench_pass(sf_handle *h, int which, int variant, int reps, float *elapsed_ms) {
    const int slices = (variant >> 8) ? (variant >> 8) : 1;  // bits 8.. of `variant`: workgroups per stream (experiment)
    variant &= 255;
    if (!h || (which != 1 && which != 2) || variant < 0 || variant > 2 || reps < 1 || slices > 64) return fail(SF_ERR_ARG, "bad argument");
    if (h->fv->id == SF_VARIANT_CLUSTER) return fail(SF_ERR_STATE, "the isolated passes are not built for the cluster variant");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemsetAsync(h->k.queue, 0, sizeof(int), h->stream));
    const int grid = std::min(h->k.batch * slices, h->max_blocks);
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    h->fv->launch_irls_pass(grid, h->stream, (const KArgs *)h->d_args, which, variant, reps, slices);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return SF_OK;
}
int sf_clear_sync_timeout(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (!h->k.cluster_g) return SF_OK;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));  // no launch of this handle is polling the granules any more
    HIP_TRY(hipMemsetAsync(h->k.sync, 0, sizeof(unsigned long long) * (size_t)h->k.batch * 2 * h->k.cluster_g * SF_SYNC_WORDS, h->stream));
    hipLaunchKernelGGL(sf_clear_sync_kernel, dim3((h->k.batch + 255) / 256), dim3(256), 0, h->stream, h->k.state, h->k.batch);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SF_OK;
}
int sf_debug_stall_rank(sf_handle *h, int rank, float stall_ms, unsigned spin_limit) {
    if (!h) return fail(SF_ERR_ARG, "null");
    if (!h->k.cluster_g) return fail(SF_ERR_STATE, "only the cluster build has rendezvous");
    if (rank >= h->k.cluster_g || stall_ms < 0.f || stall_ms > 10000.f) return fail(SF_ERR_ARG, "rank / stall out of range");
    h->k.debug_stall_rank = rank;
    h->k.debug_stall_ticks = rank < 0 ? 0u : (unsigned)(stall_ms * 1.0e5f);  // 100 MHz wall clock
    h->k.sync_spin_limit = spin_limit;
    h->args_dirty = true;
    return SF_OK;
}
int sf_last_solver_kernel_ms(sf_handle *h, float *ms) {
    if (!h || !ms) return fail(SF_ERR_ARG, "null");
    if (!h->solver_timed) return fail(SF_ERR_STATE, "no solver launch yet");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventSynchronize(h->evk1));
    HIP_TRY(hipEventElapsedTime(ms, h->evk0, h->evk1));
    return SF_OK;
}

}  // extern "C"
