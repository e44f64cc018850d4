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
    const long long t_begin = wall_clock64(), c_begin = clock64();
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
        prof[PF_SHADER_CYCLES] += clock64() - c_begin;
    }
}

#ifndef SF_CLUSTER
// prediction := current, current := pool frame f of stream b (what sf_advance_kernel does between two launches), by the
// workgroup that is about to solve the frame: the copying form -- the first frame of a launch, or a frame that does not swap
// its pyramid buffers. Level 0 of the current image is read where it lies (pyr_level: an earlier frame of this launch may have
// left it in the pool); both images end in the pyramid buffers, the stream's pool references are dropped.
__device__ __forceinline__ void advance_stream(const KArgs &a, int b, const float *pool_d, const float *pool_i, int f, int tid) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    typedef __attribute__((address_space(1))) const f4 gcf4;
    typedef __attribute__((address_space(1))) f4 gf4;
    const size_t po = (size_t)f * a.n0;
    const auto src_d = as_global(pyr_level(a, b, 0, 0, 0)), src_i = as_global(pyr_level(a, b, 0, 1, 0));
    const auto cur_d = as_global(pyr_plane(a, b, 0, 0)), cur_i = as_global(pyr_plane(a, b, 0, 1));
    const auto pred_d = as_global(pyr_plane(a, b, 1, 0)), pred_i = as_global(pyr_plane(a, b, 1, 1));
    const auto nd = as_global(pool_d + po), ni = as_global(pool_i + po);
    for (int q = tid * 4; q < a.n0; q += SF_NT * 4) {
        const f4 d = *(gcf4 *)(nd + q), i = *(gcf4 *)(ni + q);
        const f4 cd = *(gcf4 *)(src_d + q), ci = *(gcf4 *)(src_i + q);
        *(gf4 *)(pred_d + q) = cd;
        *(gf4 *)(pred_i + q) = ci;
        *(gf4 *)(cur_d + q) = d;
        *(gf4 *)(cur_i + q) = i;
    }
    __syncthreads();
    if (tid < 4) ((const float **)a.state[b].lvl0)[tid] = nullptr;
}
// The swapping form (from the second frame of a launch on): the pyramid the previous frame built for ITS current image is this
// frame's prediction pyramid -- the two buffers swap roles -- and level 0 of both images is READ IN PLACE: the previous frame's
// level 0 becomes the prediction's (wherever it lies), the new frame's is its pool frame. Nothing is copied (round 5; the launch
// used to copy the new frame into the buffer: 16 B per pixel and frame).
__device__ __forceinline__ void advance_stream_in_place(const KArgs &a, int b, const float *pool_d, const float *pool_i, int f, int flip, int tid) {
    __syncthreads();
    if (tid == 0) {
        StreamState &st = a.state[b];
        st.flip = flip ^ 1;
        st.lvl0[1][0] = st.lvl0[0][0];
        st.lvl0[1][1] = st.lvl0[0][1];
        st.lvl0[0][0] = pool_d + (size_t)f * a.n0;
        st.lvl0[0][1] = pool_i + (size_t)f * a.n0;
    }
    __syncthreads();
}
// End of a launch: level 0 of an image that is still read from the pool goes into its pyramid buffer (the host, and the next
// launch, find every image where they always were), then the references are dropped.
__device__ __forceinline__ void materialise_level0(const KArgs &a, int b, int tid) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    typedef __attribute__((address_space(1))) const f4 gcf4;
    typedef __attribute__((address_space(1))) f4 gf4;
    __syncthreads();
    for (int set = 0; set < 2; set++)
        for (int ch = 0; ch < 2; ch++) {
            const float *src = pyr_level(a, b, set, ch, 0);
            float *dst = pyr_plane(a, b, set, ch);
            if (src == dst) continue;  // (uniform)
            const auto gs = as_global(src);
            const auto gd = as_global(dst);
            for (int q = tid * 4; q < a.n0; q += SF_NT * 4) *(gf4 *)(gd + q) = *(gcf4 *)(gs + q);
        }
    __syncthreads();
    if (tid < 4) ((const float **)a.state[b].lvl0)[tid] = nullptr;
    __syncthreads();
}
// both pyramids of stream b, every level: exchange the contents of the two buffers (a stream left with flip = 1 at the
// end of a launch -- a frame_index < 0 broke the alternation -- goes back to the layout the host expects)
__device__ __forceinline__ void unflip_stream(const KArgs &a, int b, int tid) {
    typedef float __attribute__((ext_vector_type(4))) f4;
    typedef __attribute__((address_space(1))) f4 gf4;
    for (int ch = 0; ch < 2; ch++) {
        const auto x = as_global(a.pyr_new[ch] + (size_t)b * a.n_tot), y = as_global(a.pyr_pred[ch] + (size_t)b * a.n_tot);
        for (int q = tid * 4; q < a.n_tot; q += SF_NT * 4) {
            const f4 vx = *(gf4 *)(x + q), vy = *(gf4 *)(y + q);
            *(gf4 *)(x + q) = vy;
            *(gf4 *)(y + q) = vx;
        }
    }
}

__global__ __launch_bounds__(SF_NT, SF_OCC) void sf_frame_kernel(const KArgs *__restrict__ ka, const FrameLaunch fl) {
    __shared__ FrameShared sh;
    __shared__ ClusterShared cs;
    __shared__ int s_next;
    __shared__ int s_skip;
    const KArgs &a = *ka;
    const int tid = threadIdx.x;
    const int total = a.batch * fl.n_frames;
    for (;;) {
        if (tid == 0) s_next = atomicAdd(a.queue, 1);
        __syncthreads();
        const int ticket = __builtin_amdgcn_readfirstlane(s_next);  // provably wave-uniform: scalar branches around the barriers
        __syncthreads();
        if (ticket >= total) break;
        // frame-major: every frame k of every stream has a smaller ticket than any frame k + 1, so the frame a workgroup
        // waits for below is always held by a workgroup that is running (or done) -- no deadlock
        const int k = (fl.n_frames > 1) ? ticket / a.batch : 0;
        const int j = ticket - k * a.batch;
        // longest-expected-first when the host supplied an order (sf_hip.hip: streams sorted by the IRLS iterations of their
        // previous frame): the heavy streams of a launch do not end up alone in its tail
        const int b = a.order ? __builtin_amdgcn_readfirstlane(a.order[j]) : j;
        int mask = fl.stage_mask;
        if ((mask & ST_AUTO_RESIDUALS) && fl.im_count + k >= SF_HISTORY) mask |= ST_RESIDUALS;
        if (k == 0) {
            // A launch starts with every image in the handle's own buffers: materialise_level0 / unflip_stream see to that at
            // the end of a launch -- unless the launch GAVE UP on the stream (skip below), whose state then still names frames in
            // the caller's pool of that launch, which may be gone by now. Such a stream goes back to the host's layout here, in
            // the first frame of whatever is launched next (the launch that gave up cannot do it: the frame it waited for may
            // still be running); its images are undefined until they are set again, as sf.h says, but nothing dangles.
            const StreamState &st0 = a.state[b];
            const unsigned long long stale = __hip_atomic_load((const unsigned long long *)&st0.lvl0[0][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) |
                                             __hip_atomic_load((const unsigned long long *)&st0.lvl0[1][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (uniform_i(stale != 0)) {
                __syncthreads();
                if (tid < 4) ((const float **)a.state[b].lvl0)[tid] = nullptr;
                if (tid == 4) a.state[b].flip = 0;
                __syncthreads();
            }
        }
        if (fl.frame_done) {
            if (k > 0) {
                // frame k - 1 of this stream may still be running on another CU (another XCD): poll its counter, then an
                // agent-scope acquire (this CU's L1 may hold lines of the stream from two frames ago)
                if (tid == 0) {
                    unsigned spins = 0;
                    int done, skip = 0;
                    const bool forced = fl.debug_give_up == k && b % 3 == 0;  // (test support: the path below cannot be provoked otherwise)
                    while ((done = __hip_atomic_load(fl.frame_done + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < k || forced) {
                        // cannot happen (see above); never hang the device on a bug -- and never run a frame on a stream whose
                        // previous frame may still be writing: the rest of the stream's frames of this launch are skipped (the
                        // counter goes negative, so later frames see it at once) and every one of them reports the status
                        if (done < 0 || ++spins > fl.spin_limit || (forced && done >= k)) {
                            __hip_atomic_store(fl.frame_done + b, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            a.stats[b].status |= SF_STATUS_SYNC_TIMEOUT;
                            skip = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    s_skip = skip;
                }
                __syncthreads();
                const int skip_frame = __builtin_amdgcn_readfirstlane(s_skip);
                __syncthreads();
                if (skip_frame) continue;  // (the loop body ends with a barrier on this path too: the two above)
            }
            if (fl.seq_index) {
                const int f = __builtin_amdgcn_readfirstlane(fl.seq_index[(size_t)k * a.batch + b]);
                if (f >= 0) {
                    // The prediction of this frame is the current image of the previous one, whose pyramid the previous frame
                    // of this launch has built: from the second frame on the two pyramid buffers swap roles instead (no copy,
                    // no createImagePyramid(true): the same bits). The swaps alternate 0 -> 1 -> 0; the last frame only swaps
                    // BACK, so that the launch ends in the layout the host expects.
                    const int flip = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.state[b].flip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const bool swap = fl.flip_ok && k > 0 && (flip == 1 || k < fl.n_frames - 1);
                    if (swap) {
                        mask &= ~ST_PYR_OLD;
                        if (fl.flip_ok > 1) {
                            advance_stream_in_place(a, b, fl.pool_d, fl.pool_i, f, flip, tid);
                        } else {  // (-> SF_NO_POOL_IN_PLACE: round 4's form, the new frame copied into the buffer; A/B and bisection)
                            __syncthreads();
                            if (tid == 0) a.state[b].flip = flip ^ 1;
                            __syncthreads();
                            const auto cur_d = as_global(pyr_plane(a, b, 0, 0)), cur_i = as_global(pyr_plane(a, b, 0, 1));
                            const auto nd = as_global(fl.pool_d + (size_t)f * a.n0), ni = as_global(fl.pool_i + (size_t)f * a.n0);
                            typedef float __attribute__((ext_vector_type(4))) f4;
                            for (int q = tid * 4; q < a.n0; q += SF_NT * 4) {
                                const f4 d = *(__attribute__((address_space(1))) const f4 *)(nd + q), i = *(__attribute__((address_space(1))) const f4 *)(ni + q);
                                *(__attribute__((address_space(1))) f4 *)(cur_d + q) = d;
                                *(__attribute__((address_space(1))) f4 *)(cur_i + q) = i;
                            }
                        }
                    } else {
                        advance_stream(a, b, fl.pool_d, fl.pool_i, f, tid);
                    }
                }
                __syncthreads();
            }
        }
        cluster_init(*(LDS ClusterShared *)&cs, tid, 1, 0, b, b, nullptr, 0);  // this workgroup alone, on the stream's own slot
        run_stages(a, b, mask, fl.im_count + k, *(LDS FrameShared *)&sh, *(LDS ClusterShared *)&cs, tid);
        if (fl.frame_done) {
            if (fl.traj && tid < 16) fl.traj[((size_t)k * a.batch + b) * 16 + tid] = a.state[b].T[tid];
            if (k == fl.n_frames - 1) {
                if (fl.seq_index) materialise_level0(a, b, tid);
                if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.state[b].flip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
                    __syncthreads();
                    unflip_stream(a, b, tid);
                    __syncthreads();
                    if (tid == 0) a.state[b].flip = 0;
                }
            }
            // everything this workgroup wrote for the stream is visible to whoever takes its next frame: every wave's
            // stores have left, then ONE agent-scope release, then the counter (MI355X_MICROARCH.md, inter-workgroup visibility)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int seen = k;  // k -> k + 1, unless a later frame has given up on this stream (-1 stays)
                if (!__hip_atomic_compare_exchange_strong(fl.frame_done + b, &seen, k + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                    a.stats[b].status |= SF_STATUS_SYNC_TIMEOUT;  // ... whose report this frame's own statistics have just overwritten
            }
        }
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
__global__ __launch_bounds__(SF_NT, SF_OCC) void sf_frame_kernel(const KArgs *__restrict__ ka, const FrameLaunch fl) {
    __shared__ FrameShared sh;
    __shared__ ClusterShared cs;
    const KArgs &a = *ka;
    const int tid = threadIdx.x;
    const int stage_mask = fl.stage_mask, im_count = fl.im_count;  // one frame per launch: all workgroups of a stream are resident together
    const int G = a.cluster_g;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int b = (j / G) * 8 + xcd, rank = j % G;
    if (b >= a.batch) return;
    StreamState &st = a.state[b];
    // sticky: after a timeout the stream's granules and epochs are in no defined state. Its frames do nothing but report
    // the status until the host has reset them (sf_clear_sync_timeout); the solver state stays that of the last good frame
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&st.sync_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (tid == 0 && rank == 0) a.stats[b].status |= SF_STATUS_SYNC_TIMEOUT;
        return;
    }
    if (G > 1 && a.debug_stall_rank == rank && a.debug_stall_ticks) {  // test support: this workgroup is late
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (long long)a.debug_stall_ticks) __builtin_amdgcn_s_sleep(127);
    }
    cluster_init(*(LDS ClusterShared *)&cs, tid, G, rank, b, a.batch + b * G + rank, (gu64 *)a.sync + (size_t)b * 2 * G * SF_SYNC_WORDS,
                 st.sync_epoch, G > 1 ? &st.sync_failed : nullptr, a.sync_spin_limit);
    run_stages(a, b, stage_mask, im_count, *(LDS FrameShared *)&sh, *(LDS ClusterShared *)&cs, tid);
    // the epoch carries over to the next launch (every workgroup counted the same rendezvous)
    __syncthreads();
    if (tid == 0 && rank == 0) {
        if (commit_ok(*(LDS ClusterShared *)&cs))
            st.sync_epoch = cs.epoch;
        else
            a.stats[b].status |= SF_STATUS_SYNC_TIMEOUT;
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

extern "C" __attribute__((visibility("hidden"))) void SF_VARIANT_FN(sf_launch_frame)(int grid, hipStream_t st, const KArgs *ka, const FrameLaunch *fl) {
    hipLaunchKernelGGL(sf_frame_kernel, dim3(grid), dim3(SF_NT), 0, st, ka, *fl);
}
extern "C" __attribute__((visibility("hidden"))) void SF_VARIANT_FN(sf_launch_irls_pass)(int grid, hipStream_t st, const KArgs *ka, int which, int variant, int reps, int slices) {
#ifndef SF_CLUSTER
    hipLaunchKernelGGL(sf_irls_pass_kernel, dim3(grid), dim3(SF_NT), 0, st, ka, which, variant, reps, slices);
#endif
}
// bit 0: this object is a reference-order build (sf_reforder.h): the host allocates its source lists and refuses the cluster variant;
// bits 8..: SF_ORDERED_SPLAT_MAX_PIXELS as THIS object was compiled -- the host sizes the scratch blocks of the ordered coarse splat
// from it (an experiment built with another threshold would otherwise overrun blocks sized from the host's own constant)
extern "C" __attribute__((visibility("hidden"))) int SF_VARIANT_FN(sf_variant_flags)(void) { return (SF_REFORDER ? 1 : 0) | (SF_ORDERED_SPLAT_MAX_PIXELS << 8); }
extern "C" __attribute__((visibility("hidden"))) void SF_VARIANT_FN(sf_variant_geometry)(int *threads, int *blocks_per_cu) {
    *threads = SF_NT;
    *blocks_per_cu = SF_BLOCKS_PER_CU;
#ifndef SF_CLUSTER
    // what the runtime will really keep resident (registers, LDS granularity): never plan for more than that
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sf_frame_kernel, SF_NT, 0) == hipSuccess && n > 0 && n < *blocks_per_cu) *blocks_per_cu = n;
#endif
}
