// sf_cluster.h — several workgroups (CUs) working on ONE stream (the "cluster" build of the frame kernel).
//
// The one-workgroup-per-stream kernels never communicate. For a single live camera (the reference's actual use:
// StaticFusion-imagesequenceassoc.cpp:140-191, one runSolver per frame) that leaves 255 CUs idle, so the cluster build
// gives a stream G workgroups which split every per-pixel loop and meet in two kinds of rendezvous:
//
//   cluster_gather   every workgroup contributes a few 32-bit words and receives everybody's words, in rank order.
//                    The words travel as 8-byte {epoch tag, value} granules written with ONE agent-scope store each and
//                    polled with relaxed agent-scope loads: the data is the flag, no fence on either side. All
//                    cross-pixel reductions of the solver (normal equations, residual sums, maxima, counts) go this way;
//                    every workgroup then adds the G partial results in the SAME fixed rank order, so all of them hold
//                    bit-identical sums and take identical branches without any broadcast.
//   cluster_barrier  the same rendezvous with an agent-scope release before and an agent-scope acquire after it: bulk data
//                    one workgroup wrote with plain stores or atomics (pyramid levels, warp accumulators, records) may
//                    then be read by the others (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup
//                    visibility": per-CU L1s are never refreshed by other CUs' stores, per-XCD L2s are not coherent).
//
// Nothing here depends on dispatch order or on which XCD a workgroup lands; the host only PREFERS to put the workgroups
// of a stream on one XCD (speed). All workgroups of a launch must be resident at once (grid <= CUs, one 1024-thread
// workgroup per CU); every spin is bounded, a timeout sets SF_STATUS_SYNC_TIMEOUT and lets the kernel finish.
//
// Granule slots are double buffered by epoch parity: a workgroup can be at most one rendezvous ahead of the slowest
// one (it needs that one's granules of the current epoch to leave), so epoch e + 2 never overwrites data still being read.
#pragma once

#include "sf_device_common.h"

#define SF_SYNC_WORDS 128      // 32-bit payload words per workgroup and rendezvous (the largest gather moves 120)
#define SF_MAX_CLUSTER 32      // workgroups per stream (one XCD has 32 CUs)
#define SF_SYNC_SPIN_LIMIT (1u << 21)

typedef __attribute__((address_space(1))) unsigned long long gu64;

// The cluster a workgroup belongs to and the mode it currently works in. Lives in LDS (uniform over the workgroup); the
// one-workgroup kernels run the very same code with G = 1 (cluster_gather degenerates to an LDS copy), so the arithmetic of
// a sum does not depend on the build. Only the cluster build pays for a G-fold gather buffer.
#ifdef SF_CLUSTER
#define SF_GATHER_RANKS SF_MAX_CLUSTER
#else
#define SF_GATHER_RANKS 1
#endif
struct ClusterShared {
    // the work at hand
    int G, rank;     // workgroups sharing it, this workgroup's index among them (1, 0: working alone)
    int slot;        // record / accumulator slot it uses: the stream's shared slot when the cluster shares a level, this
                     // workgroup's private slot when it runs a (coarse) level redundantly on its own
    int writer;      // 1: this workgroup writes the stream's results (state, traces, counters): rank 0 of the real cluster
    // the real cluster
    int full_G, full_rank, shared_slot, private_slot;
    gu64 *sync;      // [2][full_G][SF_SYNC_WORDS] granules of this stream
    unsigned epoch;  // tag of the last rendezvous (continues across launches through StreamState::sync_epoch)
    int failed;      // a spin ran into its bound: every later rendezvous returns at once, the frame reports SF_STATUS_SYNC_TIMEOUT
    int *fail_flag;  // StreamState::sync_failed of the stream: set (agent scope) by ANY workgroup of the cluster that times out.
                     // A late workgroup finds every granule it waits for and never times out itself, although the others
                     // went on with stale words long ago: whoever commits results asks commit_ok(), which reads this flag
    unsigned spin_limit;
    unsigned in[SF_SYNC_WORDS];
    unsigned all[SF_GATHER_RANKS * SF_SYNC_WORDS];
};

__device__ __forceinline__ void cluster_init(LDS ClusterShared &cs, int tid, int G, int rank, int shared_slot, int private_slot,
                                             gu64 *sync, unsigned epoch, int *fail_flag = nullptr, unsigned spin_limit = SF_SYNC_SPIN_LIMIT) {
    if (tid == 0) {
        cs.G = cs.full_G = G;
        cs.rank = cs.full_rank = rank;
        cs.slot = cs.shared_slot = shared_slot;
        cs.private_slot = private_slot;
        cs.writer = (rank == 0) ? 1 : 0;
        cs.sync = sync;
        cs.epoch = epoch;
        cs.failed = 0;
        cs.fail_flag = fail_flag;
        cs.spin_limit = spin_limit ? spin_limit : SF_SYNC_SPIN_LIMIT;
    }
    __syncthreads();
}
// a spin ran into its bound: this workgroup stops waiting for good, and the stream is marked for everybody (sticky: the
// host clears it with sf_clear_sync_timeout)
__device__ __forceinline__ void cluster_fail(LDS ClusterShared &cs) {
    cs.failed = 1;
    int *f = *(int *volatile LDS *)&cs.fail_flag;
    if (f) __hip_atomic_store(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// May this workgroup store results into the stream's persistent state? Not when ANY workgroup of the cluster has timed
// out in this launch (or an earlier one): the words it gathered since may be stale. A workgroup that times out sets the
// flag BEFORE the late one arrives, and nobody gets past a rendezvous before the late one has arrived: the flag is up by
// the time a writer that missed the timeout asks.
__device__ __forceinline__ bool commit_ok(LDS ClusterShared &cs) {
    int *f = uniform_ptr(cs.fail_flag);
    if (!f) return true;  // one workgroup per stream: no rendezvous, nothing to time out
    const int bad = uniform_i(cs.failed) | __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(bad) == 0;
}
// solo = true: the next stages are run by this workgroup alone, on its private slot (every workgroup of the cluster does
// the same work redundantly and arrives at bit-identical state); false: back to sharing the work
__device__ __forceinline__ void cluster_set_solo(LDS ClusterShared &cs, int tid, bool solo) {
    __syncthreads();
    if (tid == 0) {
        cs.G = solo ? 1 : cs.full_G;
        cs.rank = solo ? 0 : cs.full_rank;
        cs.slot = solo ? cs.private_slot : cs.shared_slot;
    }
    __syncthreads();
}
#ifdef SF_CLUSTER
__device__ __forceinline__ int cl_G(const LDS ClusterShared &cs) { return uniform_i(cs.G); }
__device__ __forceinline__ int cl_rank(const LDS ClusterShared &cs) { return uniform_i(cs.rank); }
#else  // one workgroup per stream: compile-time constants (the rendezvous code and the strided loops fold away)
__device__ __forceinline__ constexpr int cl_G(const LDS ClusterShared &) { return 1; }
__device__ __forceinline__ constexpr int cl_rank(const LDS ClusterShared &) { return 0; }
#endif
__device__ __forceinline__ int cl_slot(const LDS ClusterShared &cs) { return uniform_i(cs.slot); }
__device__ __forceinline__ bool cl_writer(const LDS ClusterShared &cs) { return uniform_i(cs.writer) != 0; }

__device__ __forceinline__ gu64 *sync_words(gu64 *sync, int G, unsigned epoch, int rank) {
    return sync + ((size_t)(epoch & 1u) * G + rank) * SF_SYNC_WORDS;
}
__device__ __forceinline__ int sync_failed(LDS ClusterShared &cs) { return *(volatile LDS int *)&cs.failed; }

// every workgroup: words cs.in[0 .. n) -> cs.all[p * n + t] for every rank p, identical everywhere. Ends with a barrier.
// with_data_fence: also a cluster_barrier (the global writes before it are visible to everybody after it).
__device__ __forceinline__ void cluster_gather(LDS ClusterShared &cs, int n, int tid, bool with_data_fence = false) {
    const int G = cl_G(cs);
    if (G == 1) {
        __syncthreads();  // cs.in complete; the previous rendezvous' cs.all consumed
        if (tid < n) cs.all[tid] = cs.in[tid];
        __syncthreads();
        return;
    }
    if (with_data_fence) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores / atomics have left
    __syncthreads();  // cs.in complete; the previous rendezvous' cs.all consumed
    if (tid == 0) {
        cs.epoch++;
        if (with_data_fence) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-back has completed before any granule leaves
        }
    }
    __syncthreads();
    const unsigned e = (unsigned)uniform_i((int)cs.epoch);
    gu64 *sync = uniform_ptr(cs.sync);
    const int rank = cl_rank(cs);
    if (tid < n) __hip_atomic_store(sync_words(sync, G, e, rank) + tid, ((unsigned long long)e << 32) | cs.in[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int total = G * n;
    for (int q = tid; q < total; q += SF_NT) {
        const int p = q / n, t = q - p * n;
        gu64 *g = sync_words(sync, G, e, p) + t;
        unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((unsigned)(x >> 32) != e) {
            if (++spins > *(volatile LDS unsigned *)&cs.spin_limit || sync_failed(cs)) {
                cluster_fail(cs);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cs.all[q] = (unsigned)x;
    }
    __syncthreads();
    if (with_data_fence) {
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }
}

// all global writes (plain stores, atomics) of every workgroup of the cluster before the barrier are visible to every
// workgroup after it
__device__ __forceinline__ void cluster_barrier(LDS ClusterShared &cs, int tid) {
    const int G = cl_G(cs);
    if (G == 1) {
        __syncthreads();
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores / atomics have left
    __syncthreads();
    if (tid == 0) {
        const unsigned e = ++cs.epoch;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-back has completed before the flag leaves
        __hip_atomic_store(sync_words(cs.sync, G, e, cs.rank), (unsigned long long)e << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned e = (unsigned)uniform_i((int)cs.epoch);
    if (tid < G) {
        gu64 *g = sync_words(uniform_ptr(cs.sync), G, e, tid);
        unsigned spins = 0;
        while ((unsigned)(__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != e) {
            if (++spins > *(volatile LDS unsigned *)&cs.spin_limit || sync_failed(cs)) {
                cluster_fail(cs);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
}

// Rendezvous without fences: everything this workgroup wrote with agent-scope (sc1, write-through) stores or atomics has
// left, and every workgroup of the cluster has got here. What was written THAT way may then be read by the others with
// agent-scope loads (they bypass the reader's L1) -- the cheap hand-over for data both sides access with sc1 anyway
// (warp accumulators: atomics + atomic loads; the K-means label words). Plain stores / plain loads need cluster_barrier.
__device__ __forceinline__ void cluster_rendezvous(LDS ClusterShared &cs, int tid) {
    if (cl_G(cs) == 1) {
        __syncthreads();
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) cs.in[0] = 0;
    cluster_gather(cs, 1, tid);
}

// helpers to move doubles / 64-bit integers through the 32-bit payload words
__device__ __forceinline__ void put_f64(LDS unsigned *w, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    w[0] = (unsigned)b;
    w[1] = (unsigned)(b >> 32);
}
__device__ __forceinline__ double get_f64(const LDS unsigned *w) {
    return __longlong_as_double((long long)(((unsigned long long)w[1] << 32) | w[0]));
}
__device__ __forceinline__ void put_i64(LDS unsigned *w, long long v) {
    w[0] = (unsigned)(unsigned long long)v;
    w[1] = (unsigned)((unsigned long long)v >> 32);
}
__device__ __forceinline__ long long get_i64(const LDS unsigned *w) { return (long long)(((unsigned long long)w[1] << 32) | w[0]); }

// the share of [0, n) a workgroup streams in the IRLS passes: contiguous, boundaries multiples of `quantum`
__device__ __forceinline__ void cluster_range(const LDS ClusterShared &cs, int n, int quantum, int &begin, int &end) {
    const int G = cl_G(cs), rank = cl_rank(cs);
    const int per = ((n + G - 1) / G + quantum - 1) / quantum * quantum;
    begin = min(n, rank * per);
    end = min(n, begin + per);
}
