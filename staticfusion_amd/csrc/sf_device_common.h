// sf_device_common.h — shared definitions of the gfx950 solver kernels.
//
// Execution model: ONE WORKGROUP PER STREAM (independent RGB-D sequence).  A
// workgroup of SF_NT threads (8 wave64) owns a stream for a whole stage sequence; every
// "grid-wide" dependency of the reference algorithm (global max of the pre-weights, the 21+6
// normal-equation reduction, per-label residual sums, IRLS / level convergence tests) becomes a
// workgroup-local reduction through wave shuffles + LDS, so there is no inter-workgroup
// communication, no host round trip and no launch inside the coarse-to-fine loop.  256 CUs x 2
// resident workgroups keep 512 streams in flight; the streamed records of one stream (3.5 MB at
// QVGA) do not fit on chip, so the IRLS passes are HBM-bound.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sf.h"

// The frame kernels are compiled twice (sf_frame_kernels.hip): SF_NT = 256, four workgroups per CU -- the throughput
// variant, thousands of streams -- and SF_NT = 1024, one 16-wave workgroup per CU and stream -- the latency variant,
// up to a few hundred streams. The host (sf_hip.hip) picks one per handle.
#ifndef SF_NT
#define SF_NT 256  // threads per workgroup
#endif
#define SF_NW (SF_NT / 64) // waves per workgroup
#define SF_NC SF_NUM_CLUSTERS
#define SF_INVALID_LABEL 255

// in-kernel stage timers (wall_clock64, 100 MHz), accumulated per stream
enum {
    PF_PYR_OLD = 0, PF_PYR_NEW, PF_KMEANS, PF_WARP, PF_LINEARISE, PF_IRLS_INIT, PF_PASS1, PF_SOLVE6, PF_PASS2,
    PF_TAIL, PF_FILTER, PF_RESIDUALS, PF_SEGM_HIST, PF_TOTAL,
    PF_KM_INIT, PF_KM_SORT, PF_KM_ASSIGN, PF_KM_PARTITION, PF_KM_SUM, PF_KM_LABEL0, PF_KM_CONN_PYR,
    PF_SPLAT_REPLAYS = 24,  // not a timer (a slot of its own: 21..23 are shared by the profiling builds and the isolated-pass kernel): warp tiles replayed for targets outside their window (tiled_splat, lazy mode)
    PF_ORDERED_FALLBACKS = 25,  // not a timer: levels whose ordered tile splat gave up (a tap outside its window) and took the per-cell lists
    PF_SHADER_CYCLES = 26,  // the shader clock's cycles (clock64) over the same intervals as PF_TOTAL: cycles / ticks x 100 MHz = the clock the stream-frames ran at
    SF_PROF_SLOTS = 32
};

// record planes written by the linearisation and streamed by the IRLS passes.  Only what cannot be
// recomputed per pixel is stored: the warped depth, the four stencil gradients and the intensity
// difference (24 B) + one label byte; the passes also read the NEW depth from the pyramid (4 B).
// Inter depth / x / y, ddt and both pre-weights are recomputed with the reference's float
// expressions (bit-identical), which cuts the streamed bytes from 45 to 29 per pixel and pass.
enum { R_DW = 0, R_DCU, R_DCV, R_DCT, R_DDU, R_DDV, R_COUNT };

// stage mask bits of the frame kernel
enum {
    ST_PYR_OLD = 1,      // createImagePyramid(true)
    ST_PYR_NEW = 2,      // createImagePyramid(false)
    ST_KMEANS = 4,       // kMeans3DCoord + createClustersPyramidUsingKMeans
    ST_SOLVE = 8,        // coarse-to-fine loop of runSolver
    ST_RESIDUALS = 16,   // computeResidualsAgainstPreviousImage(index)
    ST_SEGM_IMAGE = 32,  // buildSegmImage
    ST_PUSH_HISTORY = 64, // ring[im_count % 5] = current
    ST_AUTO_RESIDUALS = 128 // several frames per launch: ST_RESIDUALS from the frame with im_count >= SF_HISTORY on
};

// Several consecutive frames of every stream in ONE launch of the frame kernel (sf_process_frames /
// sf_process_sequence_frames_device): the queue then hands out (frame, stream) pairs, frame-major, and a workgroup starts
// frame k of a stream as soon as frame k - 1 of THAT stream is done -- no barrier over the batch between frames, so the
// workgroups never line up again (a launch per frame makes all of them start with the same HBM-bound stage and ends with
// a tail in which the streams that needed most iterations run alone).
struct FrameLaunch {
    int stage_mask, im_count, n_frames;
    int *frame_done;        // [batch] frames of THIS launch completed per stream (zeroed by the host); null for one frame
    const int *seq_index;   // [n_frames][batch] pool frame of (frame, stream), < 0: leave the stream's images alone; or null
    const float *pool_d, *pool_i;  // [pool frame][n0]
    float *traj;            // [n_frames][batch][16] T_odometry after every frame, or null
    unsigned spin_limit;    // polls a workgroup waits for the previous frame of its stream
    int flip_ok;            // sequences: swap the roles of the two pyramid buffers instead of copying prediction := current
    int debug_give_up;      // test support: frame k = debug_give_up of every third stream is given up as if its wait had run out (0: never)
};

// fixed-point scales of the order-independent accumulations
#define FIX_DEPTH 67108864.f        /* 2^26 : warp depth accumulators */
#define FIX_INTENS 268435456.f      /* 2^28 : warp intensity accumulators (44-bit field, see ACC_W_SHIFT) */
// An accumulator cell is two 64-bit words: sum(w * depth) in Q26, and sum(w) << 44 + sum(w * intensity) in Q28 --
// the integer weight sum (each weight <= 200) shares the intensity word, so a splat is two atomics and a cell is
// 16 bytes. The intensity field holds +-2^43: > 160 full-weight contributions of intensity 1 on ONE cell (a rigid
// warp between neighbouring frames produces < 10); sum(w) holds 2^20.
#define ACC_W_SHIFT 44
#define FIX_RES 4294967296.f        /* 2^32 : per-label residual / prior sums */

// Per-stream persistent state (global memory, one per stream).
struct StreamState {
    float T[16];            // T_odometry, column-major
    float twist[6];         // twist_odometry
    float twist_level[6];   // twist_level_odometry
    float twist_old[6];     // twist_odometry_old
    float est_cov[36];
    float b_segm[SF_NC];
    float b_prior[SF_NC];
    float lambda_t_w[SF_NC];
    float kmeans[3 * SF_NC];        // column-major 3 x 24
    uint32_t conn[SF_NC];           // connectivity bit rows
    float cluster_res[SF_NC];       // perClusterAverageResidual
    float hist_T[SF_HISTORY][16];   // odomBuffer
    float kb;
    int32_t last_level;             // image level of the last executed outer iteration
    int32_t last_first;             // 1 if that iteration ran on Warped := Pred (the first of a solve)
    uint32_t sync_epoch;            // cluster build: tag of the last rendezvous of this stream's workgroups (sf_cluster.h)
    int32_t last_slot;              // record slot of the last executed outer iteration (cluster build: may be a private one)
    int32_t flip;                   // 1: the stream's two pyramid buffers have swapped roles (pyr_plane): only INSIDE a launch of
                                    // several frames of a sequence (sf_frame_kernels.hip), 0 whenever the host looks
    // Only INSIDE a launch of several SEQUENCE frames (null whenever the host looks): level 0 of the [set][channel] image is the
    // pool frame at this address instead of the pyramid buffer's level 0 (pyr_level) -- the launch reads the new frame, and as the
    // next frame's prediction the previous one, where they lie in the caller's HBM pool instead of copying 16 B per pixel and frame
    const float *lvl0[2][2];
    int32_t sync_failed;            // cluster build, sticky: a rendezvous of this stream timed out (sf_cluster.h: cluster_fail). Its frames
                                    // report SF_STATUS_SYNC_TIMEOUT and leave the state untouched until sf_clear_sync_timeout
    float inv_max_c, inv_max_d;     // 1/max of the raw pre-weights of that iteration
    long long cum_frames, cum_irls, cum_outer, cum_pixel_iters;  // totals since sf_create
    long long prof[SF_PROF_SLOTS];  // cumulative 100 MHz ticks per stage (lane 0 of the workgroup)
};

// Geometry, parameters and buffer table of a handle: lives in device memory, read through
// scalar loads (uniform addresses).  Per-launch values (stage mask, frame index) are kernel arguments.
struct KArgs {
    // geometry
    int rows, cols, levels, batch;
    int lrows[SF_MAX_LEVELS], lcols[SF_MAX_LEVELS], loff[SF_MAX_LEVELS], ln[SF_MAX_LEVELS];
    int n_tot;  // sum of level sizes
    int n0;     // level-0 size
    // parameters
    sf_params p;
    float tan_half_fovh;  // tanf(0.5f*fovh), evaluated on the host like the reference does
    // buffers
    float *pyr_new[4];   // [ch][batch][n_tot]  depth, intensity (xx, yy are recomputed: level_coord(); entries 2, 3 are null)
    float *pyr_pred[4];
    float *dbg_warped[4];  // debug_planes only, else null
    float *dbg_inter[4];
    uint8_t *labels;     // [batch][n_tot]
    long long *acc_d;    // [batch][n0] warp accumulators
    long long *acc_i;    // packed: (sum w << ACC_W_SHIFT) + sum(w * intensity)
    float *rec[R_COUNT];  // [plane][batch][n0]
    uint8_t *rec_lab;     // [batch][n0]  label of a valid pixel, SF_INVALID_LABEL otherwise
    uint8_t *rec_null;    // [batch][n0]  Null mask of the last linearisation
    const uint8_t *km_seed_lab;  // [ln[1]] label of the nearest K-means seed of every level-1 pixel (image-size constant, built at sf_create)
    float *hist_d, *hist_i;  // [SF_HISTORY][batch][n0]
    float *b_img;            // [batch][n0]
    StreamState *state;      // [batch]
    sf_frame_stats *stats;   // [batch]
    int *queue;              // work-queue counter (zeroed before every launch)
    const int *order;        // null, or the order in which the queue hands out the streams (longest expected first)
    // cluster build (sf_cluster.h): workgroups per stream; granules [batch][2][cluster_g][SF_SYNC_WORDS]. The record /
    // accumulator arrays then hold batch * (1 + cluster_g) slots of n0 pixels: slot b is stream b's shared one, slot
    // batch + b * cluster_g + r the private one of its workgroup r (coarse levels run redundantly per workgroup)
    int cluster_g;
    unsigned long long *sync;
    // test support (sf_debug_stall_rank): the workgroup of this rank of every stream idles `debug_stall_ticks` of the
    // 100 MHz clock before its first stage -- a late workgroup, as a co-running kernel causes -- and the rendezvous give up
    // after `sync_spin_limit` polls (0: SF_SYNC_SPIN_LIMIT)
    int debug_stall_rank;
    unsigned debug_stall_ticks, sync_spin_limit;
    // reference-order build only (sf_reforder.h; null otherwise): per record slot, RO_LIST_K source-pixel indices per cell
    int *ro_list;
    int ro_blocks;  // product builds: blocks of SF_ORDERED_SPLAT_MAX_PIXELS * RO_LIST_K indices in ro_list (sf_reforder.h: ro_list_of)
};

// The two pyramid buffers of a stream: set 0 = new (depthCurrent and its levels), set 1 = Pred. When consecutive frames
// of a SEQUENCE are solved inside one launch, the prediction of frame k + 1 is the current image of frame k and its pyramid
// is the pyramid frame k built: the buffers swap roles (StreamState::flip) instead of 0.6 MB being copied and a pyramid
// rebuilt bit for bit. Every stage takes its base pointers here.
__device__ __forceinline__ float *pyr_plane(const KArgs &a, int b, int set, int ch) {
    // an atomic (vector-memory) load: never the scalar data cache, which this workgroup's own store would not update
    const int f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.state[b].flip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    float *const *tab = ((set ^ f) & 1) ? a.pyr_pred : a.pyr_new;
    return tab[ch] + (size_t)b * a.n_tot;
}
// Level L of the [set][ch] image for READING: the pyramid buffer's, or -- level 0 inside a launch of sequence frames -- the pool
// frame the stream's StreamState::lvl0 names (read like `flip`: a vector-memory load, uniform, moved to SGPRs by as_global)
__device__ __forceinline__ const float *pyr_level(const KArgs &a, int b, int set, int ch, int L) {
    const float *p = pyr_plane(a, b, set, ch) + a.loff[L];
    if (L == 0) {
        const unsigned long long o = __hip_atomic_load((const unsigned long long *)&a.state[b].lvl0[set][ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)o), hi = __builtin_amdgcn_readfirstlane((unsigned)(o >> 32));
        if (lo | hi) p = (const float *)(((unsigned long long)hi << 32) | lo);
    }
    return p;
}

// ---------------------------------------------------------------------------------------------
//  wave / workgroup reductions (deterministic: fixed shuffle tree, fixed wave order)
// ---------------------------------------------------------------------------------------------
// Wave reductions on the DPP network (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast15 /
// row_bcast31 across rows: an inclusive scan whose last lane holds the total) instead of ds_bpermute
// shuffles: VALU-speed cross-lane moves, no trip through the LDS crossbar.  The association order is
// fixed, so the sums stay deterministic.  Every lane returns the total (broadcast from lane 63).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i32<CTRL, ROW_MASK>((int)(b & 0xffffffffll)), hi = dpp_i32<CTRL, ROW_MASK>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_i64(long long b) {
    const int lo = dpp_i32<CTRL, ROW_MASK>((int)(b & 0xffffffffll)), hi = dpp_i32<CTRL, ROW_MASK>((int)(b >> 32));
    return ((long long)hi << 32) | (unsigned)lo;
}
#define SF_DPP_REDUCE(v, MOVE, OP)         \
    v = OP(v, (MOVE<0x111, 0xf>(v)));      \
    v = OP(v, (MOVE<0x112, 0xf>(v)));      \
    v = OP(v, (MOVE<0x114, 0xf>(v)));      \
    v = OP(v, (MOVE<0x118, 0xf>(v)));      \
    v = OP(v, (MOVE<0x142, 0xa>(v)));      \
    v = OP(v, (MOVE<0x143, 0xc>(v)));
template <class T>
__device__ __forceinline__ T sf_op_add(T a, T b) { return a + b; }
__device__ __forceinline__ int sf_op_maxi(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int sf_op_ori(int a, int b) { return a | b; }

__device__ __forceinline__ double wave_sum_f64(double v) {
    SF_DPP_REDUCE(v, dpp_f64, sf_op_add)
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    SF_DPP_REDUCE(v, dpp_i64, sf_op_add)
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    SF_DPP_REDUCE(v, dpp_i32, sf_op_add)
    return __builtin_amdgcn_readlane(v, 63);
}
// maximum of non-negative floats (identity 0 shifts in at the row edges)
__device__ __forceinline__ float wave_max_f32(float v) {
    int b = __float_as_int(v);  // non-negative floats order like their bit patterns
    SF_DPP_REDUCE(b, dpp_i32, sf_op_maxi)
    return __int_as_float(__builtin_amdgcn_readlane(b, 63));
}

// x86 cvttss2si semantics of the reference's int(float) (reference FrontEnd.cpp:819-820): NaN and
// out-of-range -> INT_MIN.  (v_cvt_i32_f32 would saturate and turn NaN into 0, which the
// reference's `uwarp >= 0` test would then ACCEPT.)
__device__ __forceinline__ int cvt_trunc_x86(float x) {
    if (!(x > -2147483648.f && x < 2147483648.f)) return (int)0x80000000;
    return (int)x;
}

// Buffer pointers come out of the KArgs table as generic pointers; every stage casts them to the
// GLOBAL address space so that loads / stores / atomics compile to global_* instructions (vmcnt only).
// With flat_* accesses every LDS wait also drains the outstanding global loads, which serialises
// prefetched loads behind the first LDS access.
// LDS: the workgroup's shared structs are passed to the (separately compiled) stage functions as
// references qualified with the LOCAL address space, so that their members are accessed with ds_*
// instructions (a plain reference is a generic pointer: every access becomes a flat_* instruction
// that waits on both memory counters).
#define LDS __attribute__((address_space(3)))
typedef float __attribute__((ext_vector_type(2))) vfloat2;  // plain vector types: usable in any address space
typedef float __attribute__((ext_vector_type(4))) vfloat4;
template <class T>
__device__ __forceinline__ void lds_add(LDS T *p, T v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <class T>
__device__ __forceinline__ void lds_min(LDS T *p, T v) {
    __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <class T>
__device__ __forceinline__ void lds_or(LDS T *p, T v) {
    __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// a pointer that is the same in every lane, moved to an SGPR pair (arguments of the __noinline__
// stage functions arrive in VGPRs, which would force per-lane 64-bit address arithmetic)
template <class P>
__device__ __forceinline__ P uniform_ptr(P p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (P)(((unsigned long long)hi << 32) | lo);
}
template <class T>
using gptr = __attribute__((address_space(1))) T *;
template <class T>
__device__ __forceinline__ gptr<T> as_global(T *p) {  // every base pointer of a stage is workgroup-uniform
    return uniform_ptr((gptr<T>)p);
}
// Element idx >= 0 of an array whose base is workgroup-uniform, addressed as SGPR base + 32-bit unsigned BYTE offset:
// the form the hardware takes directly (global_load ... v_off, s[base]). `p[idx]` with a signed int costs a sign extension
// and a 64-bit shift-add in VALU per access -- about 7 % of all VALU instructions of a frame before this was used.
// (The offset passes through an empty asm so that two accesses with the same index do not share one zero-extension node:
// instruction selection folds the extension into the addressing mode only when the access is its single user.)
__device__ __forceinline__ unsigned gbyte_off(int idx, unsigned elem) {
    unsigned off = (unsigned)idx * elem;
    asm("" : "+v"(off));
    return off;
}
template <class T>
__device__ __forceinline__ T gld(gptr<const T> p, int idx) {
    return *(gptr<const T>)((__attribute__((address_space(1))) const char *)p + gbyte_off(idx, (unsigned)sizeof(T)));
}
template <class T>
__device__ __forceinline__ T gld(gptr<T> p, int idx) {
    return *(gptr<const T>)((__attribute__((address_space(1))) const char *)p + gbyte_off(idx, (unsigned)sizeof(T)));
}
template <class T, class V>
__device__ __forceinline__ void gst(gptr<T> p, int idx, V v) {
    *(gptr<T>)((__attribute__((address_space(1))) char *)p + gbyte_off(idx, (unsigned)sizeof(T))) = (T)v;
}
// Scope of the warp / residual accumulator cells. In the one-workgroup-per-stream builds a stream's accumulators are
// touched by ONE workgroup between two kernel boundaries, so workgroup scope is all the coherence they need: the atomics
// then complete in this XCD's L2 and a cell is written back once, when its line is evicted. Agent scope (what a cluster
// of workgroups on different CUs needs) makes every atomic write through to the fabric: the zero store AND the sums both
// reached HBM, 34 bytes written per pixel and warp where 16 are needed (profiles/r02j_traffic_by_stage.txt).
#if defined(SF_CLUSTER) || defined(SF_ACC_AGENT)
#define SF_ACC_SCOPE __HIP_MEMORY_SCOPE_AGENT
#else
#define SF_ACC_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
__device__ __forceinline__ void gatomic_add_at(gptr<long long> p, int idx, long long v) {
    __hip_atomic_fetch_add((gptr<long long>)((__attribute__((address_space(1))) char *)p + gbyte_off(idx, 8u)), v, __ATOMIC_RELAXED, SF_ACC_SCOPE);
}
template <class P>
__device__ __forceinline__ long long gld_agent_i64(P p, int idx) {  // load of a 64-bit accumulator cell at the accumulators' scope
    return __hip_atomic_load((gptr<const long long>)((__attribute__((address_space(1))) const char *)p + gbyte_off(idx, 8u)), __ATOMIC_RELAXED,
                             SF_ACC_SCOPE);
}
__device__ __forceinline__ void gatomic_add(gptr<long long> p, long long v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gatomic_add(gptr<uint32_t> p, uint32_t v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a value every lane holds identically (read from LDS / memory): move it to an SGPR
__device__ __forceinline__ float uniform_f(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ int uniform_i(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ __forceinline__ float sqf(float x) { return x * x; }
// std::max / std::min of the reference (argument order matters for NaN)
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }

// (long long)(y * scale) without the emulated float -> int64 conversion: v = y * scale (|v| <= 2^52 for every use here, the
// split works up to 2^57) is cut exactly into trunc(v / 2^26) and the remainder (both exact in float: v has 24 significant
// bits), each converted with the native 32-bit instruction. Truncation toward zero like the C conversion, bit for bit.
__device__ __forceinline__ long long to_fix(float x, float scale, float lim) {
    float y = x;
    if (!(y < lim)) y = lim;  // also catches NaN
    if (!(y > -lim)) y = -lim;
    const float v = y * scale;
    const float vh = truncf(v * (1.f / 67108864.f));
    const float vl = v - vh * 67108864.f;
    return (long long)(int)vh * 67108864ll + (long long)(int)vl;
}
// the same value for |lim * scale| < 2^31: one conversion (truncation toward zero, like the two-step form above)
__device__ __forceinline__ int to_fix_i32(float x, float scale, float lim) {
    float y = x;
    if (!(y < lim)) y = lim;
    if (!(y > -lim)) y = -lim;
    return (int)(y * scale);
}

// aggregate a per-lane 64-bit value into bins[label], one LDS atomic per distinct label per wave
__device__ __forceinline__ void wave_label_add_i64(bool active, int lab, long long v, LDS long long *bins, int lane) {
    unsigned long long rem = __ballot(active);
    while (rem) {
        const int src = __ffsll((long long)rem) - 1;
        const int l = __builtin_amdgcn_readlane(lab, src);
        const bool mine = active && lab == l;
        const unsigned long long m = __ballot(mine);
        const long long sum = wave_sum_i64(mine ? v : 0ll);
        if (lane == 0) lds_add(bins + l, sum);
        rem &= ~m;
    }
}
__device__ __forceinline__ void wave_label_count(bool active, int lab, LDS int *bins, int lane) {
    unsigned long long rem = __ballot(active);
    while (rem) {
        const int src = __ffsll((long long)rem) - 1;
        const int l = __builtin_amdgcn_readlane(lab, src);
        const unsigned long long m = __ballot(active && lab == l);
        if (lane == 0) lds_add(bins + l, (int)__popcll(m));
        rem &= ~m;
    }
}


// ---------------------------------------------------------------------------------------------
//  xx / yy of the pyramids (reference FrontEnd.cpp:385-386: xx = (inv_f_i (u - disp_u_i)) depth) are not
//  stored: every consumer has the depth and the pixel position and repeats this expression (same
//  operations, same bits), which saves 8 B per pixel on the pyramid write and on every read.
// ---------------------------------------------------------------------------------------------
struct LevelCoord {
    float inv_f_i, disp_u_i, disp_v_i, inv_rows;
    int rows_i;
};
__device__ __forceinline__ LevelCoord level_coord(const KArgs &a, int L) {
    LevelCoord c;
    const int rows_i = a.lrows[L], cols_i = a.lcols[L];
    c.inv_f_i = 2.f * a.tan_half_fovh / float(cols_i);
    c.disp_u_i = 0.5f * (cols_i - 1);
    c.disp_v_i = 0.5f * (rows_i - 1);
    c.inv_rows = 1.f / float(rows_i);
    c.rows_i = rows_i;
    return c;
}
__device__ __forceinline__ float coord_x(const LevelCoord &c, int u, float d) { return (c.inv_f_i * (float(u) - c.disp_u_i)) * d; }
__device__ __forceinline__ float coord_y(const LevelCoord &c, int v, float d) { return (c.inv_f_i * (float(v) - c.disp_v_i)) * d; }
// flat column-major index -> (u, v), exact (float reciprocal estimate + integer correction)
__device__ __forceinline__ void split_uv(const LevelCoord &c, int idx, int &u, int &v) {
    int q = (int)((float)idx * c.inv_rows);
    if (q * c.rows_i > idx) q--;
    if ((q + 1) * c.rows_i <= idx) q++;
    u = q;
    v = idx - q * c.rows_i;
}

// ---------------------------------------------------------------------------------------------
//  forward splat of one source pixel (reference FrontEnd.cpp:808-868 / :960-1019): transform with T
//  (rows 0..2 of the inverse odometry, row-major 3x4), project to centi-pixels, distribute to the
//  1 or 4 neighbouring target pixels with integer weights.  Integer atomics: order independent.
// ---------------------------------------------------------------------------------------------
struct SplatGeom {
    float T[12];
    float f, disp_u_i, disp_v_i;
    int cols_lim, rows_lim, rows_i;
};

// a * w for a 64-bit fixed-point value |a| < 2^55 and a weight 0 <= w < 2^23: the low word times w as one 32 x 32 -> 64
// product, the (small, signed) high word through the 24-bit multiplier. Equal to the plain 64-bit product, in 3 instructions.
__device__ __forceinline__ long long mul_i64_w(long long a, int w) {
    const unsigned long long lo = (unsigned long long)(unsigned)a * (unsigned)w;
    const int hi = __mul24((int)(a >> 32), w) + (int)(lo >> 32);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// the packed increment of an intensity cell, (w << ACC_W_SHIFT) + w * jf, for a 32-bit fixed-point intensity: one signed
// 32 x 32 -> 64 product, the weight added into the high word
__device__ __forceinline__ long long mul_packed_w(int jf, int w) {
    const long long p = (long long)jf * (long long)w;
    const int hi = (int)(p >> 32) + (w << (ACC_W_SHIFT - 32));
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)p);
}

// depth / intensity of a target pixel from its fixed-point accumulators: sum(w * value) / sum(w).
// The integer sums are exact; one int64 -> float conversion and one IEEE float division round twice
// (<= 1 ulp from the exact quotient, the same order as the reference's own float accumulation).
// The two quotients share the divisor (an integer in [1, 2^20], exact in float): one hardware reciprocal (1 ulp), refined
// by a Newton step, and a residual correction per quotient -- the correctly rounded quotient in all but a vanishing share
// of the cases, within 1 ulp always, at a third of the instructions of two IEEE division sequences.
#ifndef SF_FAST_NORMALISE
#define SF_FAST_NORMALISE 1
#endif
__device__ __forceinline__ void normalise_acc(long long sd, long long packed, float &dw, float &iw) {
    const long long si = (long long)((unsigned long long)packed << (64 - ACC_W_SHIFT)) >> (64 - ACC_W_SHIFT);
    const float wf = (float)(unsigned)((packed - si) >> ACC_W_SHIFT);
    const float nd = (float)sd * (1.f / 67108864.f), ni = (float)si * (1.f / FIX_INTENS);
#if SF_FAST_NORMALISE
    float r = __builtin_amdgcn_rcpf(wf);
    r = fmaf(fmaf(-wf, r, 1.f), r, r);
    float q = nd * r;
    dw = fmaf(fmaf(-wf, q, nd), r, q);
    q = ni * r;
    iw = fmaf(fmaf(-wf, q, ni), r, q);
#else
    dw = nd / wf;
    iw = ni / wf;
#endif
}

// initializeKMeans (reference KMeans.cpp:63-101): the seed a level-1 pixel starts with is the nearest of 24 seed positions
// that depend on the image size only, in the reference's unsigned wrap-around arithmetic. One table per handle
// (KArgs::km_seed_lab, built by sf_seed_label_kernel at sf_create) instead of 24 squared distances per pixel and frame.
__device__ __forceinline__ unsigned km_seed_u(int cols_km, int l) { return (unsigned)roundf((unsigned)(l + 1) * (float(cols_km) / float(SF_NC + 1))); }
__device__ __forceinline__ unsigned km_seed_v(int rows_km, int l) { return (unsigned)roundf((unsigned)(l % 5 + 1) * (float(rows_km) / float(5 + 1))); }  // 5 = ceil(sqrt(24))
__device__ __forceinline__ unsigned km_nearest_seed(int rows_km, int cols_km, unsigned u, unsigned v) {
    unsigned lab = SF_NC, min_dist = 1000000u;
    for (unsigned l = 0; l < SF_NC; l++) {
        const unsigned dv = v - km_seed_v(rows_km, (int)l), du = u - km_seed_u(cols_km, (int)l);  // unsigned wrap-around as in the reference
        const unsigned q = dv * dv + du * du;
        if (q < min_dist) {
            lab = l;
            min_dist = q;
        }
    }
    return lab;
}

#define SF_LOAD_BATCH 4  // independent pixels whose loads are issued before any of them is consumed

// ---------------------------------------------------------------------------------------------
//  Tiled splat: the level is walked in source tiles of SPLAT_TV x SPLAT_TU pixels.  The targets of
//  a tile fall into a small window of the warped image (a rigid warp is locally a shift), which is
//  accumulated in LDS with integer ds_add atomics and then added to the global accumulators once
//  per touched cell: ~3x fewer, fully coalesced global atomics than one global atomic triple per
//  bilinear tap.  Targets outside the window (strong local stretch) take the global path directly.
//  Integer sums => the result is independent of both orders.
// ---------------------------------------------------------------------------------------------
#define SPLAT_TV 64
#ifndef SPLAT_TU
#define SPLAT_TU ((SF_NT == 256 ? 4 : 2) * SF_NT / 64)
#endif
#define SPLAT_PX ((SPLAT_TV * SPLAT_TU) / SF_NT)  // source pixels per lane and tile
#ifndef SPLAT_MARGIN
#define SPLAT_MARGIN 6  // window cells beyond the tile size in each direction (a rigid warp is locally a shift: rarely more)
#endif
#define WIN_V (SPLAT_TV + SPLAT_MARGIN)
#define WIN_U (SPLAT_TU + SPLAT_MARGIN)
#define WIN_CELLS (WIN_V * WIN_U)

#define SPLAT_MAX_LAZY_TILES 512
#define SPLAT_LAZY_COLS 640  // lazy mode keeps a row watermark per accumulator column (SplatMarks)
#ifndef SF_SPLAT_FRESH
#define SF_SPLAT_FRESH 1
#endif
struct SplatWin {
    long long d[WIN_CELLS];
    long long i[WIN_CELLS];  // packed like the global cell
    int vmin, umin;
    int vmax, umax;  // lazy mode: the last row / column any tap of the tile can reach (the flush walks the touched box, not the window)
    unsigned ovf[SPLAT_MAX_LAZY_TILES / 32];  // lazy mode: tiles with targets outside their window (replayed at the end)
};
// Lazy initialisation (one workgroup per stream only): no pass zeroes the accumulator image before the splat. A watermark
// per accumulator COLUMN (SplatMarks: rows [0, zrow[c]) of column c hold zero or sums) tells the flush of a window which
// of its cells nobody has written yet: those are STORED -- the window's value, zero included -- and only cells below the
// watermark, which an earlier window reached, take atomics; rows between the watermark and the window's first row are
// stored as zero, and what no window reached is zeroed after the last tile. A cell then crosses the fabric once on its way
// out (the L2 writes stores through and gives a line up after an atomic: zero + atomic cost two write-backs and a fetch)
// instead of being zeroed first and added to afterwards. The tiles walk down a strip of SPLAT_TU columns and then move
// right, so the watermark of a column only grows. Targets outside a tile's window (rare: strong local stretch) cannot go
// straight to the global cells -- their cell may not be initialised yet -- so the tile is flagged and replayed after the
// last tile. (-DSF_SPLAT_FRESH=0: the round-2 form -- zero the columns a window reaches first, atomics for every cell.)
#ifndef SF_SPLAT_BOX
#define SF_SPLAT_BOX 1  // 0: the flush walks the whole window (round 4)
#endif
struct SplatMarks {
    unsigned short zrow[SPLAT_LAZY_COLS];
};
__device__ __forceinline__ bool splat_lazy_ok(int rows_i, int cols_i, int G) {
    const int tiles = ((rows_i + SPLAT_TV - 1) / SPLAT_TV) * ((cols_i + SPLAT_TU - 1) / SPLAT_TU);
    return G == 1 && tiles <= SPLAT_MAX_LAZY_TILES && cols_i <= SPLAT_LAZY_COLS && rows_i < 65536;
}

// Src::load(v, u, idx, z, xr, yr, iw) -> bool valid
template <class Src>
__device__ __forceinline__ void tiled_splat(const SplatGeom &g, int rows_i, int cols_i, const Src &src, gptr<long long> acc_d,
                                            gptr<long long> acc_i, LDS SplatWin &win, LDS SplatMarks &marks, int tid, int tile_first = 0,
                                            int tile_step = 1, bool lazy = false, long long *replayed = nullptr) {
    const int lane = tid & 63;
    const int tiles_v = (rows_i + SPLAT_TV - 1) / SPLAT_TV, tiles_u = (cols_i + SPLAT_TU - 1) / SPLAT_TU;
    const int n_tiles = tiles_v * tiles_u;
    int zcol = 0;  // lazy: accumulator columns [0, zcol) are zero or hold sums already
    if (lazy) {
        if (tid < SPLAT_MAX_LAZY_TILES / 32) win.ovf[tid] = 0;  // ordered before its first use by the tile loop's barriers
#if SF_SPLAT_FRESH
        for (int c = tid; c < cols_i; c += SF_NT) marks.zrow[c] = 0;
#endif
    }
    // the rows [from, to) of column c, for every lane that raises `flag`, zeroed by the whole wave (call it wave-uniformly)
    auto zero_flagged = [&](bool flag, int c, int from, int to) {
        for (unsigned long long m = __ballot(flag); m; m &= m - 1) {
            const int l = __builtin_ctzll(m);
            const int cc = __builtin_amdgcn_readlane(c, l), tt = __builtin_amdgcn_readlane(to, l);
            for (int v = __builtin_amdgcn_readlane(from, l) + lane; v < tt; v += 64) {
                gst(acc_d, v + cc * rows_i, 0ll);
                gst(acc_i, v + cc * rows_i, 0ll);
            }
        }
    };
    int pend_u0 = -1, pend_nu = 0, pend_z = 0;  // the columns the last flush reached and their new watermark (set one trip later: no barrier of its own)
    // lazy: a second walk over the tiles (it >= n_tiles) replays the flagged ones for their out-of-window targets
    for (int it = tile_first; it < (lazy ? 2 * n_tiles : n_tiles); it += tile_step) {  // a cluster's workgroups take every G-th tile
        const bool replay = it >= n_tiles;
        if (SF_SPLAT_FRESH && pend_u0 >= 0) {  // (the flush that read the watermarks ended with a barrier)
            if (tid < pend_nu && pend_u0 + tid < cols_i && pend_z > (int)marks.zrow[pend_u0 + tid]) marks.zrow[pend_u0 + tid] = (unsigned short)pend_z;
            pend_u0 = -1;
        }
        const int tile = replay ? it - n_tiles : it;
        if (replay) {
            if (it == n_tiles) {  // every tile flushed: zero what no window reached; the flags are complete
#if SF_SPLAT_FRESH
                __syncthreads();  // the last window's watermarks (set at the top of this trip) are visible
                for (int c0 = 0; c0 < cols_i; c0 += SF_NT) {  // a lane per column; a column short of the last row is rare
                    const int c = c0 + tid;
                    const int z = c < cols_i ? (int)marks.zrow[c] : rows_i;
                    zero_flagged(z < rows_i, c, z, rows_i);
                }
#else
                for (int idx = zcol * rows_i + tid; idx < cols_i * rows_i; idx += SF_NT) {
                    gst(acc_d, idx, 0ll);
                    gst(acc_i, idx, 0ll);
                }
#endif
                __syncthreads();
            }
            if (!((uniform_i((int)win.ovf[tile >> 5]) >> (tile & 31)) & 1)) continue;
            if (tid == 0 && replayed) *replayed += 1;  // diagnostic counter (slot 24 of sf_get_stage_profile)
        }
        const int tv0 = (tile % tiles_v) * SPLAT_TV, tu0 = (tile / tiles_v) * SPLAT_TU;
        // ---- phase 1: clear the window, load + project this lane's source pixels, window origin
        for (int q = tid; q < WIN_CELLS; q += SF_NT) {
            win.d[q] = 0;
            win.i[q] = 0;
        }
        if (tid == 0) {
            win.vmin = 0x7fffffff;
            win.umin = 0x7fffffff;
            win.vmax = -1;
            win.umax = -1;
        }
        // target pixel (qu, qv) and the centi-pixel offsets (ru, rv) inside it: uwarp = 100 qu + ru (reference FrontEnd.cpp:819-853)
        int qu[SPLAT_PX], ru[SPLAT_PX], qv[SPLAT_PX], rv[SPLAT_PX];
        long long dfix[SPLAT_PX];
        int ifix[SPLAT_PX];
        bool ok[SPLAT_PX];
        float z[SPLAT_PX], xr[SPLAT_PX], yr[SPLAT_PX], iw[SPLAT_PX];
#pragma unroll
        for (int k = 0; k < SPLAT_PX; k++) {
            const int v = tv0 + lane, u = tu0 + (tid >> 6) + k * (SF_NT / 64);
            const bool inside = v < rows_i && u < cols_i;
            const int idx = inside ? v + u * rows_i : 0;
            ok[k] = src.load(v, u, idx, z[k], xr[k], yr[k], iw[k]) && inside;
        }
        int vtop = 0, utop = 0;  // max over the lane's valid pixels of INT_MAX - q (q >= 0): the wave maximum gives the minimum
        int vbot = 0, ubot = 0;  // ... and of q + 2 (0: no valid pixel): the last row / column a tap can reach, + 1
#pragma unroll
        for (int k = 0; k < SPLAT_PX; k++) {
            const float x_w = g.T[0] * xr[k] + g.T[1] * yr[k] + g.T[2] * z[k] + g.T[3];
            const float y_w = g.T[4] * xr[k] + g.T[5] * yr[k] + g.T[6] * z[k] + g.T[7];
            const float depth_w = g.T[8] * xr[k] + g.T[9] * yr[k] + g.T[10] * z[k] + g.T[11];
            const int uw = cvt_trunc_x86(100.f * (g.f * x_w / depth_w + g.disp_u_i));
            const int vw = cvt_trunc_x86(100.f * (g.f * y_w / depth_w + g.disp_v_i));
            ok[k] = ok[k] && (uw >= 0) && (uw < g.cols_lim) && (vw >= 0) && (vw < g.rows_lim);
            const unsigned uu = ok[k] ? (unsigned)uw : 0u, vv = ok[k] ? (unsigned)vw : 0u;  // non-negative: unsigned division
            qu[k] = (int)(uu / 100u);
            ru[k] = (int)(uu - 100u * (unsigned)qu[k]);
            qv[k] = (int)(vv / 100u);
            rv[k] = (int)(vv - 100u * (unsigned)qv[k]);
            dfix[k] = to_fix(depth_w, FIX_DEPTH, 1000.f);
            ifix[k] = to_fix_i32(iw[k], FIX_INTENS, 4.f);
            if (ok[k]) {
                vtop = max(vtop, 0x7fffffff - qv[k]);
                utop = max(utop, 0x7fffffff - qu[k]);
                vbot = max(vbot, qv[k] + 2);
                ubot = max(ubot, qu[k] + 2);
            }
        }
        SF_DPP_REDUCE(vtop, dpp_i32, sf_op_maxi)
        SF_DPP_REDUCE(utop, dpp_i32, sf_op_maxi)
        const int vmin = 0x7fffffff - __builtin_amdgcn_readlane(vtop, 63), umin = 0x7fffffff - __builtin_amdgcn_readlane(utop, 63);
        int vlast = 0, ulast = 0;
        if (SF_SPLAT_BOX && lazy) {  // (uniform)
            SF_DPP_REDUCE(vbot, dpp_i32, sf_op_maxi)
            SF_DPP_REDUCE(ubot, dpp_i32, sf_op_maxi)
            vlast = __builtin_amdgcn_readlane(vbot, 63) - 1;
            ulast = __builtin_amdgcn_readlane(ubot, 63) - 1;
        }
        __syncthreads();  // window cleared, origin initialised
        if (lane == 0) {
            lds_min(&win.vmin, vmin);
            lds_min(&win.umin, umin);
            if (SF_SPLAT_BOX && lazy) {
                __hip_atomic_fetch_max(&win.vmax, vlast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(&win.umax, ulast, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        const int wv0 = uniform_i(win.vmin), wu0 = uniform_i(win.umin);
        if (!SF_SPLAT_FRESH && lazy && !replay && wu0 != 0x7fffffff) {  // zero the columns this window reaches first (stores ordered
                                                                        // before the flush's atomics by the barrier in front of phase 3)
            const int need = min(cols_i, wu0 + WIN_U);
            if (need > zcol) {
                for (int idx = zcol * rows_i + tid; idx < need * rows_i; idx += SF_NT) {
                    gst(acc_d, idx, 0ll);
                    gst(acc_i, idx, 0ll);
                }
                zcol = need;
            }
        }
        bool outside = false;  // lazy: this lane had a target outside the window
        // ---- phase 2: splat into the window (LDS atomics), or straight to global if outside
        auto add = [&](int v, int u, int w, long long df, int jf) {
            const int dv = v - wv0, du = u - wu0;
            if (dv >= 0 && dv < WIN_V && du >= 0 && du < WIN_U) {
                if (!replay) {
                    const int c = dv + du * WIN_V;
                    lds_add(&win.d[c], mul_i64_w(df, w));
                    lds_add(&win.i[c], mul_packed_w(jf, w));
                }
            } else if (lazy && !replay) {
                outside = true;
            } else {
                const int t = v + u * g.rows_i;
                gatomic_add_at(acc_d, t, mul_i64_w(df, w));
                gatomic_add_at(acc_i, t, mul_packed_w(jf, w));
            }
        };
#pragma unroll
        for (int k = 0; k < SPLAT_PX; k++) {
            if (!ok[k]) continue;
            const int delta_l = ru[k], delta_r = 100 - ru[k], delta_d = rv[k], delta_u = 100 - rv[k];
            const long long df = dfix[k];
            const int jf = ifix[k];
            if (min(delta_r, delta_l) + min(delta_u, delta_d) < 5) {  // within 5 centi-pixels of a pixel centre
                add(delta_u > delta_d ? qv[k] : qv[k] + 1, delta_r > delta_l ? qu[k] : qu[k] + 1, 200, df, jf);
            } else {
                const int dv0 = qv[k] - wv0, du0 = qu[k] - wu0;  // >= 0: the window origin is the tile's minimum
                if (dv0 < WIN_V - 1 && du0 < WIN_U - 1) {        // the 2 x 2 block lies in the window: one test, four fixed offsets
                    if (replay) continue;
                    const int c = dv0 + du0 * WIN_V;
                    const int w11 = delta_l + delta_d, w10 = delta_r + delta_d, w01 = delta_l + delta_u, w00 = delta_r + delta_u;
                    lds_add(&win.d[c + WIN_V + 1], mul_i64_w(df, w11));
                    lds_add(&win.i[c + WIN_V + 1], mul_packed_w(jf, w11));
                    lds_add(&win.d[c + 1], mul_i64_w(df, w10));
                    lds_add(&win.i[c + 1], mul_packed_w(jf, w10));
                    lds_add(&win.d[c + WIN_V], mul_i64_w(df, w01));
                    lds_add(&win.i[c + WIN_V], mul_packed_w(jf, w01));
                    lds_add(&win.d[c], mul_i64_w(df, w00));
                    lds_add(&win.i[c], mul_packed_w(jf, w00));
                } else {
                    add(qv[k] + 1, qu[k] + 1, delta_l + delta_d, df, jf);
                    add(qv[k] + 1, qu[k], delta_r + delta_d, df, jf);
                    add(qv[k], qu[k] + 1, delta_l + delta_u, df, jf);
                    add(qv[k], qu[k], delta_r + delta_u, df, jf);
                }
            }
        }
        if (lazy && !replay && __any(outside)) {
            if (lane == 0) lds_or(&win.ovf[tile >> 5], 1u << (tile & 31));
        }
        __syncthreads();
        // ---- phase 3: add the touched cells to the global accumulators (consecutive lanes -> consecutive v)
        if (SF_SPLAT_FRESH && lazy) {
            if (!replay && wu0 != 0x7fffffff) {  // (no valid source pixel in the tile: nothing to flush, no column reached)
                // Groups of 16 lanes take 16 rows that start on a multiple of 16 (a 128-byte line of cells where the level's
                // rows are a multiple of 16, as at QVGA's level 0): the stores are whole lines, written once. Rows [r0, znew)
                // of every window column: the window's cells and the padding up to the next multiple of 16 either side.
                // (round 5) only the box the tile's taps can have reached -- a rigid warp moves a 64 x 16 tile into about 66 x 18 cells of
                // its 70 x 22 window --: what lies beyond it in the window holds zeros that nobody needs to write now (the
                // watermarks say what is initialised; a later window, or the sweep after the last tile, takes care of the rest)
                const int vreach = SF_SPLAT_BOX ? min(wv0 + WIN_V, uniform_i(win.vmax) + 1) : wv0 + WIN_V;
                const int ureach = SF_SPLAT_BOX ? min(WIN_U, uniform_i(win.umax) + 1 - wu0) : WIN_U;
                const int vend = min(vreach, rows_i);
                const int znew = min((vend + 15) & ~15, rows_i), r0 = wv0 & ~15;
                const int ng = (znew - r0 + 15) >> 4;  // groups per column
                // gi / ng == (gi * mdiv) >> 16 for gi * ng < 65536 (mdiv = floor(65536 / ng) + 1; the quotient is far from an
                // integer unless ng is a power of two, where the reciprocal is exact)
                const unsigned mdiv = (unsigned)(65536.f * __builtin_amdgcn_rcpf((float)ng)) + 1u;
                const int ncols = min(ureach, cols_i - wu0);
                const int total = ng * ncols;
                for (int g0 = 0; g0 < total; g0 += SF_NT / 16) {  // (wave-uniform trip count: zero_flagged wants the whole wave)
                    const int gi = g0 + (tid >> 4);
                    const int du = (int)(((unsigned)gi * mdiv) >> 16), gr = gi - du * ng;
                    const int c = wu0 + du, v = r0 + (gr << 4) + (tid & 15);
                    const bool live = gi < total && v < znew;
                    const int z = live ? (int)marks.zrow[c] : 0x7fff;
                    const int dv = v - wv0;
                    const bool in_win = live && dv >= 0 && v < vend;
                    const int q = in_win ? dv + du * WIN_V : 0;
                    const long long packed = in_win ? win.i[q] : 0ll, sd = in_win ? win.d[q] : 0ll;
                    const int t = v + c * g.rows_i;
                    if (live && v >= z) {  // nobody has written this cell: the window's value -- or zero -- is its value
                        gst(acc_d, t, sd);
                        gst(acc_i, t, packed);
                    } else if (packed != 0) {
                        gatomic_add_at(acc_d, t, sd);
                        gatomic_add_at(acc_i, t, packed);
                    }
                    // rows between the watermark and r0 (a column the window above did not reach): the first lane of the column's
                    // first group reports it
                    zero_flagged(live && gr == 0 && (tid & 15) == 0 && z < r0, c, z, r0);
                }
                pend_u0 = wu0;
                pend_nu = ncols;
                pend_z = znew;
            }
        } else if (!replay)
        for (int q = tid; q < WIN_CELLS; q += SF_NT) {
            const long long packed = win.i[q];
            if (packed == 0) continue;  // sum(w) >= 1 makes a touched cell non-zero
            const int du = q / WIN_V, dv = q - du * WIN_V;
            const int t = (wv0 + dv) + (wu0 + du) * g.rows_i;
            gatomic_add_at(acc_d, t, win.d[q]);
            gatomic_add_at(acc_i, t, packed);
        }
        __syncthreads();  // before the next tile clears the window
    }
}

// the reference-order build (-DSF_REFORDER=1 -> libsf_hip_reforder.so): a parity instrument, see the header
#ifndef SF_REFORDER
#define SF_REFORDER 0
#endif
#include "sf_reforder.h"
