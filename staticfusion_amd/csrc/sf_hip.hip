// sf_hip.hip — the product: libsf_hip.so, the MI355X (gfx950) implementation of include/sf.h.
//
// One persistent kernel, `sf_frame_kernel`, runs any subset of the frame stages for every
// stream of the batch; workgroups pull stream indices from an atomic queue.  Each C-ABI entry
// point that the reference exposes as a StaticFusion method is one launch of that kernel with
// the corresponding stage mask; sf_process_frame is ONE launch with the whole per-frame
// sequence of the reference drivers (StaticFusion-datasets.cpp:171-184).
//
// There is no CPU fallback and no dependence on the test oracle.
#include "sf_host.h"

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
extern "C" __attribute__((visibility("hidden"))) int sf_variant_flags_nt256(void);
static const FrameVariant VARIANTS[3] = {
    {SF_VARIANT_THROUGHPUT, "throughput", sf_variant_geometry_nt256, sf_launch_frame_nt256, sf_launch_irls_pass_nt256, sf_launch_debug_rows_nt256},
    {SF_VARIANT_LATENCY, "latency", sf_variant_geometry_nt1024, sf_launch_frame_nt1024, sf_launch_irls_pass_nt1024, sf_launch_debug_rows_nt1024},
    {SF_VARIANT_CLUSTER, "cluster", sf_variant_geometry_ntcluster, sf_launch_frame_ntcluster, sf_launch_irls_pass_ntcluster, sf_launch_debug_rows_ntcluster},
};

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

// sf_clear_sync_timeout: epochs and the sticky timeout flag of every stream (the granules are zeroed by a memset)
__global__ __launch_bounds__(256) void sf_clear_sync_kernel(StreamState *state, int batch) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < batch) {
        state[b].sync_epoch = 0;
        state[b].sync_failed = 0;
        // a multi-frame launch that gave up on a stream (sf_frame_kernels.hip: skip) may have left its pyramid buffers swapped and
        // level 0 referring to the caller's pool: back to the layout the host assumes (the images are to be set again)
        state[b].flip = 0;
        for (int q = 0; q < 4; q++) ((const float **)state[b].lvl0)[q] = nullptr;
    }
}

// the nearest K-means seed of every level-1 pixel (KArgs::km_seed_lab): once per handle, with the device arithmetic
__global__ __launch_bounds__(256) void sf_seed_label_kernel(uint8_t *out, int rows_km, int cols_km) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows_km * cols_km) return;
    const int u = idx / rows_km, v = idx - u * rows_km;
    out[idx] = (uint8_t)km_nearest_seed(rows_km, cols_km, (unsigned)u, (unsigned)v);
}

// =============================================================================================
//  host side
// =============================================================================================
static thread_local std::string g_err;
int sf_fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

// Cluster launches need every one of their workgroups resident at once (sf_cluster.h): two of them must not share the
// GPU. All cluster launches of the process on one device are therefore chained through an event -- a launch waits (on the
// device) for the previous cluster launch of ANY handle. Kernels of other kinds on other streams are the caller's business:
// if they keep workgroups of a cluster launch from being scheduled, the frame reports SF_STATUS_SYNC_TIMEOUT.
#include <mutex>
static std::mutex g_cluster_mu;
static hipEvent_t g_cluster_done[64] = {};

// throughput build: the 5-workgroups-per-CU compilation of the frame kernel serves the full solver (Makefile: frame_nt256o5.o)
bool use_five_per_cu(const sf_handle *h) {
    if (!h->max_blocks_o5) return false;
    if (const char *v = std::getenv("SF_THROUGHPUT_WG_PER_CU")) return v[0] == '5';  // pins one of the two (A/B tooling)
    return h->k.p.segmentation_enabled != 0;
}

// One launch of the frame kernel: `n_frames` consecutive frames of every stream (1: the per-call API). ml: the per-launch
// pointers of a multi-frame launch (frame counters, index table, pools, trajectory), or null.
int launch(sf_handle *h, int mask, int im_count, int n_frames, const FrameLaunch *ml) {
    // The labelling at full resolution starts every pixel's search at labels_lowres(v/2, u/2) (KMeans.cpp:267), a matrix of
    // rows/2 x cols/2 entries: with an odd image size the reference reads past its last row / column, and whatever lies there
    // decides labels. There is nothing to be identical to: no launch runs K-means on such a handle (pure odometry, the input
    // stage, prediction and the map have no such read and take any size sf_create accepts).
    if ((mask & ST_KMEANS) && ((h->k.rows | h->k.cols) & 1)) return fail(SF_ERR_ARG, "K-means with segmentation_enabled needs even rows and cols (the reference reads labels_lowres(v/2, u/2) outside its matrix otherwise, KMeans.cpp:267)");
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
    // test support (tests/test_multi_frame.py): frame k of every third stream is given up once its previous frame is done -- the
    // path of a wait that ran into its bound, which nothing else can provoke
    if (const char *v = std::getenv("SF_DEBUG_GIVE_UP_AT_FRAME")) fl.debug_give_up = (int)std::strtol(v, nullptr, 10);
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

int solve_mask(const sf_handle *h, int create_image_pyr) {
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
int sf_abi_version(int *sizeof_params, int *sizeof_frame_stats, int *stage_profile_slots) {
    if (sizeof_params) *sizeof_params = (int)sizeof(sf_params);
    if (sizeof_frame_stats) *sizeof_frame_stats = (int)sizeof(sf_frame_stats);
    if (stage_profile_slots) *stage_profile_slots = SF_PROF_SLOTS;
    return SF_ABI_VERSION;
}
const char *sf_backend(void) { return (sf_variant_flags_nt256() & 1) ? "hip:gfx950:reference-order" : "hip:gfx950"; }

static int validate_params(const sf_params *p, int levels) {
    // K-means clusters image level 1 (KMeans.cpp:145): a one-level pyramid is only meaningful without segmentation
    if (levels < (p->segmentation_enabled ? 2 : 1) || levels > SF_MAX_LEVELS)
        return fail(SF_ERR_ARG, "ctf_levels must be in [2, 8] (1 is accepted with segmentation_enabled = 0)");
    if (p->max_iter_per_level < 1 || p->max_iter_irls < 1 || levels * p->max_iter_per_level > SF_MAX_OUTER)
        return fail(SF_ERR_ARG, "iteration counts out of range");
    return SF_OK;
}

// (orphan_maps: sf_hip_model.hip)
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
    h->reforder = (sf_variant_flags_nt256() & 1) != 0;
    if (h->reforder && h->fv->id == SF_VARIANT_CLUSTER) {
        sf_destroy(h);
        return fail(SF_ERR_ARG, "the reference-order build (libsf_hip_reforder.so) runs one workgroup per stream: no SF_VARIANT_CLUSTER");
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
    if (h->reforder) {
        TRY_OR_FREE(dev_alloc(h, &k.ro_list, slots * N0 * RO_LIST_K));  // source indices per cell of every record slot (sf_reforder.h)
    } else {
        // the ordered float splat of the coarse levels (sf_reforder.h): source lists per RESIDENT WORKGROUP, 256 KB each at the
        // product's 2048 pixels (one per stream while that is fewer: sf_reforder.h, ro_list_of). The block size is the FRAME
        // OBJECT's SF_ORDERED_SPLAT_MAX_PIXELS (sf_variant_flags), which indexes them -- never smaller than this file's own
        const size_t wgs = h->cluster_grid ? (size_t)h->cluster_grid : std::min<size_t>(B, (size_t)std::max(h->max_blocks, h->max_blocks_o5));
        const size_t block_px = std::max<size_t>((size_t)(sf_variant_flags_nt256() >> 8), (size_t)SF_ORDERED_SPLAT_MAX_PIXELS);
        TRY_OR_FREE(dev_alloc(h, &k.ro_list, wgs * block_px * RO_LIST_K));
        k.ro_blocks = (int)wgs;
    }
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

int check_stream(const sf_handle *h, int stream) {
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

int sf_clear_sync_timeout(sf_handle *h) {
    if (!h) return fail(SF_ERR_ARG, "null");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));  // no launch of this handle is polling the granules any more
    if (h->k.cluster_g)
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
}  // extern "C"
