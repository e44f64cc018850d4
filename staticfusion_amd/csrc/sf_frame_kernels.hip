// sf_frame_kernels.hip — the persistent frame kernel (and the isolated IRLS pass kernel) of libsf_hip.so.
// Compiled twice into the library: -DSF_NT=256 (throughput variant) and -DSF_NT=1024 (latency variant); every
// workgroup-size dependent constant (waves per workgroup, tile and window sizes, LDS layout) derives from SF_NT.
// The host side (sf_hip.hip) reaches the kernels through the two extern "C" launchers at the end.
#include <hip/hip_runtime.h>

#include "sf_cluster.h"
#include "sf_device_common.h"
#include "sf_kmeans.h"
#ifdef SF_CLUSTER
#include "sf_kmeans_cluster.h"
#endif
#include "sf_pyramid.h"
#include "sf_residuals.h"
#include "sf_smallmath.h"
#include "sf_solver.h"

#ifndef SF_OCC
#define SF_OCC 4
#endif
#define SF_BLOCKS_PER_CU (SF_OCC * 256 / SF_NT)  // 16 waves per CU at <= 128 VGPRs (5 x 256 per CU measured 3 % slower: DESIGN.md §9)
// __launch_bounds__(threads, 4): the second HIP parameter is the minimum number of WAVES PER SIMD, not workgroups per
// CU: 4 waves per SIMD = 16 waves per CU = <= 128 VGPRs for every workgroup size
#define SF_PASTE2(a, b) a##b
#define SF_PASTE(a, b) SF_PASTE2(a, b)
#ifndef SF_VARIANT_TAG
#define SF_VARIANT_TAG SF_NT
#endif
#define SF_VARIANT_FN(name) SF_PASTE(SF_PASTE(name, _nt), SF_VARIANT_TAG)
#define sf_frame_kernel SF_VARIANT_FN(sf_frame_kernel)
#define sf_irls_pass_kernel SF_VARIANT_FN(sf_irls_pass_kernel)
#define sf_debug_rows_kernel SF_VARIANT_FN(sf_debug_rows_kernel)

union FrameShared {
    KmShared km;
#ifdef SF_CLUSTER
    KmClusterShared kmc;
#endif
    SolveShared sv;
    ResShared rs;
};

// the stage sequence of one stream; cs says whether this workgroup works alone or as one of a cluster (sf_cluster.h)
__device__ __forceinline__ void run_stages(const KArgs &a, int b, int stage_mask, int im_count, LDS FrameShared &sh, LDS ClusterShared &cs, int tid) {
    long long t0 = 0, t1 = 0;
    long long *prof = a.state[b].prof;
    const bool writer = cl_writer(cs);
#ifdef SF_NO_STAGE_TIMED
#define STAGE_TIMED(slot, call) call
#else
#define STAGE_TIMED(slot, call)               \
    do {                                      \
        call;                                 \
        if (tid == 0 && writer) {             \
            t1 = wall_clock64();              \
            prof[slot] += t1 - t0;            \
            t0 = t1;                          \
        }                                     \
    } while (0)
#endif
    const long long t_begin = wall_clock64();
    t0 = t_begin;
    if ((stage_mask & (ST_PYR_OLD | ST_PYR_NEW)) == (ST_PYR_OLD | ST_PYR_NEW)) {
        STAGE_TIMED(PF_PYR_NEW, stage_pyramid(a, b, 3, tid, cs));  // both pyramids behind one barrier / rendezvous per level
    } else {
        if (stage_mask & ST_PYR_OLD) STAGE_TIMED(PF_PYR_OLD, stage_pyramid(a, b, 1, tid, cs));
        if (stage_mask & ST_PYR_NEW) STAGE_TIMED(PF_PYR_NEW, stage_pyramid(a, b, 2, tid, cs));
    }
    if (stage_mask & ST_KMEANS) {
#ifdef SF_CLUSTER
        if (cl_G(cs) > 1) {
            STAGE_TIMED(PF_KMEANS, stage_kmeans_cluster(a, b, *(LDS KmClusterShared *)&sh.kmc, cs, tid));
        } else
#endif
        {
            STAGE_TIMED(PF_KMEANS, stage_kmeans(a, b, *(LDS KmShared *)&sh.km, tid));
        }
        cluster_barrier(cs, tid);  // the labels of every level are visible to every workgroup of the cluster
    }
    if (stage_mask & ST_SOLVE) {
        stage_solve(a, b, *(LDS SolveShared *)&sh.sv, cs, tid);
        t0 = wall_clock64();
    }
    // the frame loop pushes the current images into the ring slot the residual stage has just read (im_count % 5 both):
    // the residual pass then stores them as it goes, having loaded them anyway
    const bool push_in_residuals = (stage_mask & ST_RESIDUALS) && (stage_mask & ST_PUSH_HISTORY);
    if (stage_mask & ST_RESIDUALS)
        STAGE_TIMED(PF_RESIDUALS, stage_residuals(a, b, im_count, push_in_residuals, *(LDS ResShared *)&sh.rs, cs, tid));
    __syncthreads();
    if (stage_mask & ST_SEGM_IMAGE) stage_segm_image(a, b, tid, cs);
    if (stage_mask & ST_PUSH_HISTORY) stage_push_history(a, b, im_count, !push_in_residuals, tid, cs);
    if (tid == 0 && writer) {
        t1 = wall_clock64();
        prof[PF_SEGM_HIST] += t1 - t0;
        prof[PF_TOTAL] += t1 - t_begin;
    }
}

#ifndef SF_CLUSTER
__global__ __launch_bounds__(SF_NT, SF_OCC) void sf_frame_kernel(const KArgs *__restrict__ ka, int stage_mask, int im_count) {
    __shared__ FrameShared sh;
    __shared__ ClusterShared cs;
    __shared__ int s_next;
    const KArgs &a = *ka;
    const int tid = threadIdx.x;
    for (;;) {
        if (tid == 0) {
            const int ticket = atomicAdd(a.queue, 1);
            // longest-expected-first when the host supplied an order (sf_hip.hip: streams sorted by the IRLS iterations of their
            // previous frame): the heavy streams of a launch do not end up alone in its tail
            s_next = (ticket < a.batch && a.order) ? a.order[ticket] : ticket;
        }
        __syncthreads();
        const int b = __builtin_amdgcn_readfirstlane(s_next);  // provably wave-uniform: scalar branches around the barriers
        __syncthreads();
        if (b >= a.batch) break;
        cluster_init(*(LDS ClusterShared *)&cs, tid, 1, 0, b, b, nullptr, 0);  // this workgroup alone, on the stream's own slot
        run_stages(a, b, stage_mask, im_count, *(LDS FrameShared *)&sh, *(LDS ClusterShared *)&cs, tid);
        // NOTE: the loop body must END with a barrier. With a lane-0-only block here the compiler
        // fuses it with the lane-0-only queue pop at the loop top into an outer loop, and the other
        // lanes of wave 0 then spin on the barrier of the inner loop forever (observed hang).
        __syncthreads();
    }
}
#else
// The cluster build: G workgroups per stream, fixed assignment (no queue: all workgroups of a stream must run at the same
// time, so the grid never exceeds the CUs). Workgroup j of XCD x -- blocks are dealt to the XCDs round robin, an
// observation the mapping only uses for speed -- serves stream (j / G) * 8 + x as rank j % G: the workgroups of a stream
// share an L2.
__global__ __launch_bounds__(SF_NT, SF_OCC) void sf_frame_kernel(const KArgs *__restrict__ ka, int stage_mask, int im_count) {
    __shared__ FrameShared sh;
    __shared__ ClusterShared cs;
    const KArgs &a = *ka;
    const int tid = threadIdx.x;
    const int G = a.cluster_g;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int b = (j / G) * 8 + xcd, rank = j % G;
    if (b >= a.batch) return;
    StreamState &st = a.state[b];
    cluster_init(*(LDS ClusterShared *)&cs, tid, G, rank, b, a.batch + b * G + rank, (gu64 *)a.sync + (size_t)b * 2 * G * SF_SYNC_WORDS,
                 st.sync_epoch);
    run_stages(a, b, stage_mask, im_count, *(LDS FrameShared *)&sh, *(LDS ClusterShared *)&cs, tid);
    // the epoch carries over to the next launch (every workgroup counted the same rendezvous)
    __syncthreads();
    if (tid == 0 && rank == 0) {
        st.sync_epoch = cs.epoch;
        if (cs.failed) a.stats[b].status |= SF_STATUS_SYNC_TIMEOUT;
    }
}
#endif

// the IRLS passes alone (measurement support; never part of a solve)
#ifndef SF_CLUSTER
__global__ __launch_bounds__(SF_NT, 4) void sf_irls_pass_kernel(const KArgs *__restrict__ ka, int which, int variant, int reps, int slices) {
    __shared__ FrameShared sh;
    __shared__ int s_next;
    const KArgs &a = *ka;
    const int tid = threadIdx.x;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (;;) {
        if (tid == 0) s_next = atomicAdd(a.queue, 1);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(s_next);
        __syncthreads();
        const int b = item / slices, slice = item - b * slices;
        if (b >= a.batch) {
            if (tid == 0 && blockIdx.x == 0) {  // shader clock estimate: s_memtime ticks per 100 MHz tick
                a.state[0].prof[22] = clock64() - c0;
                a.state[0].prof[23] = wall_clock64() - w0;
            }
            break;
        }
        if (which == 1) {
            if (variant == 0) microbench_pass<1, 0>(a, b, slice, slices, reps, *(LDS SolveShared *)&sh.sv, tid);
            if (variant == 1) microbench_pass<1, 1>(a, b, slice, slices, reps, *(LDS SolveShared *)&sh.sv, tid);
            if (variant == 2) microbench_pass<1, 2>(a, b, slice, slices, reps, *(LDS SolveShared *)&sh.sv, tid);
        } else {
            if (variant == 0) microbench_pass<2, 0>(a, b, slice, slices, reps, *(LDS SolveShared *)&sh.sv, tid);
            if (variant == 1) microbench_pass<2, 1>(a, b, slice, slices, reps, *(LDS SolveShared *)&sh.sv, tid);
            if (variant == 2) microbench_pass<2, 2>(a, b, slice, slices, reps, *(LDS SolveShared *)&sh.sv, tid);
        }
        __syncthreads();
    }
}


#endif  // !SF_CLUSTER

// test support: the Jacobian rows of one stream's last outer iteration (sf_get_jacobian_rows); never part of a solve
__global__ __launch_bounds__(SF_NT) void sf_debug_rows_kernel(const KArgs *__restrict__ ka, int b, float *out) {
    debug_rows(*ka, b, out, blockIdx.x * SF_NT + threadIdx.x, gridDim.x * SF_NT);
}
extern "C" __attribute__((visibility("hidden"))) void SF_VARIANT_FN(sf_launch_debug_rows)(int grid, hipStream_t st, const KArgs *ka, int b, float *out) {
    hipLaunchKernelGGL(sf_debug_rows_kernel, dim3(grid), dim3(SF_NT), 0, st, ka, b, out);
}

extern "C" __attribute__((visibility("hidden"))) void SF_VARIANT_FN(sf_launch_frame)(int grid, hipStream_t st, const KArgs *ka, int stage_mask, int im_count) {
    hipLaunchKernelGGL(sf_frame_kernel, dim3(grid), dim3(SF_NT), 0, st, ka, stage_mask, im_count);
}
extern "C" __attribute__((visibility("hidden"))) void SF_VARIANT_FN(sf_launch_irls_pass)(int grid, hipStream_t st, const KArgs *ka, int which, int variant, int reps, int slices) {
#ifndef SF_CLUSTER
    hipLaunchKernelGGL(sf_irls_pass_kernel, dim3(grid), dim3(SF_NT), 0, st, ka, which, variant, reps, slices);
#endif
}
extern "C" __attribute__((visibility("hidden"))) void SF_VARIANT_FN(sf_variant_geometry)(int *threads, int *blocks_per_cu) {
    *threads = SF_NT;
    *blocks_per_cu = SF_BLOCKS_PER_CU;
#ifndef SF_CLUSTER
    // what the runtime will really keep resident (registers, LDS granularity): never plan for more than that
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sf_frame_kernel, SF_NT, 0) == hipSuccess && n > 0 && n < *blocks_per_cu) *blocks_per_cu = n;
#endif
}
