// sf_host.h — what the host-side translation units of libsf_hip.so share: the handle, the error helpers, device
// allocation, and the few functions one part calls in another. Nothing here is exported (include/sf.h is the ABI).
//   sf_hip.hip         handle lifetime, parameters, the builds of the frame kernel, launch()
//   sf_hip_solver.hip  images in, frames (one or several per launch), results and debug planes out, measurement support
//   sf_hip_input.hip   the input stage: loader decimation / RGB -> intensity, bilateral depth filter (sf_input.h)
//   sf_hip_model.hip   frame-to-model prediction and the surfel map without OpenGL (sf_predict.h, sf_fusion.h)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sf_cluster.h"
#include "sf_device_common.h"

#define SF_INTERNAL __attribute__((visibility("hidden")))

// the two builds of the frame kernels (sf_frame_kernels.hip, -DSF_NT=256 / -DSF_NT=1024)
struct FrameVariant {
    int id;  // SF_VARIANT_*
    const char *name;
    void (*geometry)(int *threads, int *blocks_per_cu);
    void (*launch_frame)(int grid, hipStream_t st, const KArgs *ka, const FrameLaunch *fl);
    void (*launch_irls_pass)(int grid, hipStream_t st, const KArgs *ka, int which, int variant, int reps, int slices);
    void (*launch_debug_rows)(int grid, hipStream_t st, const KArgs *ka, int b, float *out);
};

struct sf_handle {
    KArgs k{};
    int device = 0;
    int max_blocks = 0;
    int wg_per_cu = 0;
    int *d_order = nullptr;   // KArgs::order storage (more streams than resident workgroups: a launch has a tail)
    int max_blocks_o5 = 0;  // throughput build: resident workgroups of the 5-per-CU kernel (0: not used)
    const FrameVariant *fv = nullptr;  // set by sf_create_ex
    std::vector<struct sf_map *> maps;  // live maps created from this handle: sf_destroy releases their memory and orphans them
    bool reforder = false;  // the library's frame kernels are the reference-order build (sf_reforder.h)
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
    int multi_capacity = 0;          // frames the two index buffers hold
    int traj_capacity = 0;           // frames d_traj holds
    int solver_timed_frames = 1;     // frames of the launch evk0 / evk1 bracket
};

// the thread's last error text (sf_last_error); returns `code`
SF_INTERNAL int sf_fail(int code, const std::string &msg);
#define fail sf_fail
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

// shared between the parts (defined in the file named)
SF_INTERNAL int launch(sf_handle *h, int mask, int im_count, int n_frames = 1, const FrameLaunch *ml = nullptr);  // sf_hip.hip
SF_INTERNAL int solve_mask(const sf_handle *h, int create_image_pyr);                                            // sf_hip.hip
SF_INTERNAL bool use_five_per_cu(const sf_handle *h);                                                           // sf_hip.hip
extern "C" {  // (defined inside the extern "C" blocks of their files)
SF_INTERNAL int check_stream(const sf_handle *h, int stream);                 // sf_hip.hip
SF_INTERNAL int d2h(sf_handle *h, void *dst, const void *src, size_t bytes);  // sf_hip_solver.hip: drain the handle's stream, then copy
SF_INTERNAL int input_alloc(sf_handle *h);                                    // sf_hip_input.hip: buffers of the input stage, on first use
SF_INTERNAL void orphan_maps(sf_handle *h);                                   // sf_hip_model.hip: sf_destroy releases the handle's maps
}
