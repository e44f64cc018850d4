// sf_solver.h — the coarse-to-fine coupled odometry + segmentation solve of one stream.
// Replaces the loop of StaticFusion::runSolver (reference FrontEnd.cpp:1091-1144) and everything
// it calls: warpImagesAccurateInverse (:775-892), calculateCoord (:393-430), calculateDerivatives
// (:432-479), computeWeights (:481-510), computeSegPrior (SegmentationBackground.cpp:53-103),
// solveOdometryAndSegmJoint (:513-692) with buildSystemSegm / solveSegmIteration
// (SegmentationBackground.cpp:105-174) and filterEstimateAndComputeT (:713-772).
//
// Data flow per outer iteration at level L (N_L pixels):
//   warp        scatter the Pred level into 3 order-independent fixed-point accumulators
//               (64-bit integer atomics: the result does not depend on the scatter order)
//   linearise   register strips (a wave sweeps the columns of 62 rows; LDS tiles in the cluster build): Inter images,
//               edge-aware gradients, temporal differences, raw pre-weights  ->  11 float planes + 1 label byte per pixel
//               ("records", 45 B/px), plus the global max of the pre-weights and the per-label prior
//   IRLS        <= max_iter_irls iterations of two streaming passes over the records:
//                 pass 1  rebuild the two Jacobian rows, Cauchy x b weights, accumulate the 21+6
//                         normal-equation sums in fp64 registers -> wave shuffle -> LDS -> 6x6 LDL^T
//                 pass 2  residuals with the new solution, per-label |res| sums (fixed point),
//                         ||res||^2 -> 24x24 LDL^T for b, convergence test
//   filter      covariance, eigen-space velocity filter, SE(3) update (one lane, fp64)
// The Jacobian matrix A (2N x 6) of the reference is never materialised.
#pragma once

#include "sf_cluster.h"
#include "sf_device_common.h"
#include "sf_smallmath.h"
// the linearisation walks register strips in the one-workgroup builds (solve_linearise_strips) and LDS tiles in a cluster,
// whose workgroups share a level tile by tile (solve_linearise); -DSF_LIN_STRIPS=0: tiles everywhere (A/B)
#ifndef SF_LIN_STRIPS
#ifdef SF_CLUSTER
#define SF_LIN_STRIPS 0
#else
#define SF_LIN_STRIPS 1
#endif
#endif
#if SF_REFORDER && SF_RO_BEHIND
#define LS_RO_BEHIND 1  // validPixels by the reference's rule (:415-427), carried by the label plane
#else
#define LS_RO_BEHIND 0
#endif
#ifndef LS_ROWS
#define LS_ROWS 62  // rows a wave owns in a strip (lanes 1 .. LS_ROWS; lane 0 and lane LS_ROWS + 1 hold the halo rows)
#endif
#define TILE_V 64
#define TILE_U (2 * SF_NT / TILE_V)  // two centre pixels per lane
#define TILE_LV (TILE_V + 2)
#define TILE_LU (TILE_U + 2)
#define TILE_N (TILE_LV * TILE_LU)

// Per-pixel IRLS weights use the hardware reciprocal / reciprocal-square-root (1 ulp) instead of the
// IEEE division + square root sequences (~10 VALU instructions each; pass 1 is VALU-bound). The
// linearisation (max weights, records) stays bit-identical to the oracle; the solver result moves by
// ~1e-7, three orders of magnitude inside the pose tolerance. -DSF_FAST_WEIGHTS=0 restores IEEE.
#ifndef SF_FAST_WEIGHTS
#define SF_FAST_WEIGHTS 1
#endif
#if SF_FAST_WEIGHTS
__device__ __forceinline__ float vrsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float vrcpw(float x) { return __builtin_amdgcn_rcpf(x); }
#else
__device__ __forceinline__ float vrsq(float x) { return sqrtf(1.f / x); }
__device__ __forceinline__ float vrcpw(float x) { return 1.f / x; }
#endif


struct LinTile {  // linearisation tile (with halo)
    float t_D[TILE_N], t_I[TILE_N];    // Inter depth / intensity
    float t_dn[TILE_N], t_in[TILE_N];  // new depth / intensity
    float t_dw[TILE_N], t_iw[TILE_N];  // warped depth / intensity
    uint8_t t_null[TILE_N];
};

// Pass 1 keeps the 27 normal-equation sums per lane in fp32 and, every SF_P1_FLUSH pixel pairs, adds them -- reduced over a
// group of P1_GROUP neighbouring lanes on the DPP network -- into fp64 sums in LDS (one set per lane group; entry-major, so
// the group leaders of a wave touch consecutive 8-byte words). A lane's fp32 partial sum then never holds more than
// 4 SF_P1_FLUSH terms: the rounding error of the accumulated AtA / AtB drops about tenfold against one fp32 sum over the
// lane's whole share (<= 600 terms at QVGA), which is what moved b by 4e-5 against the oracle's fp64 sums ([C1]).
#ifndef SF_P1_FLUSH
#define SF_P1_FLUSH 32
#endif
#define P1_GROUP (SF_NT == 256 ? 4 : 16)  // 1024-thread builds: a lane sums a quarter of the terms, rows of 16 lanes share a set
#define P1_SETS (SF_NT / P1_GROUP)
#define P1_SETS_PER_WAVE (64 / P1_GROUP)

struct SolveShared {
    union {            // the warp window and the linearisation tile are never live together; the fp64 scratch of the
        LinTile lt;    // one-lane algebra (4 x 4 inverse before a warp, motion filter after the IRLS, 3 x 3 inverse at the
        SplatWin win;  // end of the solve) is used while neither is, and so are the fp64 sums of pass 1
        double dwork[36 * 3 + 32];
        double p1[27][P1_SETS];
#if SF_REFORDER
        RoChunk ro;    // reference-order build: a chunk of the ordered per-cluster sums
        RoChunk2 ro2;  // ... with the two residuals of every pixel (`ro2.c` IS `ro`)
        RoRows rows;   // ... a chunk of weighted rows for the row-by-row fp64 sums of pass 1
#endif
    };
    // reductions
    double red[SF_NW][28];
    float redf[SF_NW][2];
    int redi[SF_NW];
    long long lab_sum[SF_NC];
    long long prior_sum[SF_NC];
    int prior_size[SF_NC], prior_nonnull[SF_NC], valid_cnt[SF_NC];
    // stream state
    float T[16], Tinv[16];
    float twist[6], twist_level[6], twist_old[6];
    float est_cov[36];
    float b_segm[SF_NC], b_prior[SF_NC], lambda_t_w[SF_NC];
    unsigned conn[SF_NC];
    float kb;
    // IRLS
    float AtA[36], AtB[6], Var[6], prev_sol[6];
    float aver_res, aver_res_old, inv_max_c, inv_max_d, res_sqnorm;
    float last_delta;  // |Var - prev_sol|_inf of the last IRLS iteration (the trace reports it)
    double sq_total;  // ||res||^2 of the last pass 2, summed over the workgroups of the cluster
    int px_begin, px_end;  // pixel range of the level the streaming passes walk: this workgroup's share of the level
    int rec_slot;          // record slot the passes stream (the stream's, or this workgroup's private one)
    double init_abs_c, init_abs_d;  // sum of wc |dct| and wd |ddt| over validPixels (raw pre-weights), from the linearisation
    int n_valid, ctrl, status, n_irls, n_outer, first;
    long long pixel_iters;
    // small solves
    float M6[6 * 7], tmp6[6], y6[6];
    int tr6[6];
    union {
        float M24[SF_NC * (SF_NC + 1)];  // factored and used inside solve_irls
        SplatMarks marks;                // the warp's column watermarks (solve_warp, between two solve_irls)
    };
    float tmp24[SF_NC], y24[SF_NC], seg_diag[SF_NC], aver_res_label[SF_NC];
    int tr24[SF_NC];
    int seg_allzero;
    long long prof[SF_PROF_SLOTS], t_last;
};

#ifdef SF_NO_PROF_MARK
#define PROF_MARK(s, tid, slot) do {} while (0)
#else
#define PROF_MARK(s, tid, slot)                         \
    do {                                                \
        if ((tid) == 0) {                               \
            const long long now_ = wall_clock64();      \
            (s).prof[slot] += now_ - (s).t_last;        \
            (s).t_last = now_;                          \
        }                                               \
    } while (0)
#endif

// ---------------------------------------------------------------------------------------------
//  Jacobian rows of one pixel (reference FrontEnd.cpp:544-585). Expressions keep the reference's
//  association; the build uses -ffp-contract=off.
// ---------------------------------------------------------------------------------------------
// Records are read through GLOBAL address-space pointers (global_load_*, not flat_load_*) and
// SF_VEC consecutive pixels per lane (8- or 16-byte loads: more bytes in flight per wave).
typedef __attribute__((address_space(1))) const float gcfloat;
typedef __attribute__((address_space(1))) const uint8_t gcu8;
typedef __attribute__((address_space(1))) const vfloat2 gcfloat2;
typedef __attribute__((address_space(1))) const vfloat4 gcfloat4;
typedef __attribute__((address_space(1))) const unsigned short gcu16;
typedef __attribute__((address_space(1))) const unsigned int gcu32;

struct RecPtrs {
    int with_labels;
    gcfloat *p[R_COUNT];
    gcfloat *dnew;  // NEW depth of the level (pyramid plane)
    gcu8 *lab;
};

// base (uniform, SGPR pair) + 32-bit unsigned byte offset (one VGPR shared by all planes): the
// saddr + voffset form of global_load, no 64-bit per-plane address arithmetic in the loop
typedef __attribute__((address_space(1))) const char gcchar;
template <int VEC>
__device__ __forceinline__ void load_plane(gcfloat *p, int idx0, float (&out)[VEC]) {
    const unsigned boff = (unsigned)idx0 * 4u;
    gcchar *q = (gcchar *)p + boff;
    if constexpr (VEC == 1) {
        out[0] = *(gcfloat *)q;
    } else if constexpr (VEC == 2) {
        const vfloat2 v = *(gcfloat2 *)q;
        out[0] = v.x;
        out[1] = v.y;
    } else {
        const vfloat4 v = *(gcfloat4 *)q;
        out[0] = v.x;
        out[1] = v.y;
        out[2] = v.z;
        out[3] = v.w;
    }
}
template <int VEC>
__device__ __forceinline__ void load_labels(gcu8 *p, int idx0, int (&out)[VEC]) {
    gcchar *q = (gcchar *)p + (unsigned)idx0;
    if constexpr (VEC == 1) {
        out[0] = *(gcu8 *)q;
    } else if constexpr (VEC == 2) {
        const unsigned v = *(gcu16 *)q;
        out[0] = v & 255u;
        out[1] = v >> 8;
    } else {
        const unsigned v = *(gcu32 *)q;
        out[0] = v & 255u;
        out[1] = (v >> 8) & 255u;
        out[2] = (v >> 16) & 255u;
        out[3] = v >> 24;
    }
}

template <int VEC>
struct RecVec {
    float v[R_COUNT][VEC];
    float dn[VEC];
    unsigned labraw;  // the VEC label bytes as loaded; unpacked at the point of use (rec_label)
    int lab[VEC];
};
template <int VEC>
__device__ __forceinline__ void load_rec(const RecPtrs &rp, int idx0, RecVec<VEC> &r) {
    static_assert(VEC == 2, "the passes walk pixel pairs");
    // uniform: without segmentation every valid pixel belongs to cluster 0 and the plane is not read. The bytes are kept
    // as loaded: unpacking them here, inside the branch, made the compiler wait for the load (s_waitcnt vmcnt(0)) BEFORE
    // the other seven loads of the record were issued -- two memory round trips per trip of the loop
    unsigned raw = 0;
    if (rp.with_labels) raw = *(gcu16 *)((gcchar *)rp.lab + (unsigned)idx0);
    r.labraw = raw;
    load_plane<VEC>(rp.dnew, idx0, r.dn);
#pragma unroll
    for (int q = 0; q < R_COUNT; q++) load_plane<VEC>(rp.p[q], idx0, r.v[q]);
}

// Per-level constants needed to rebuild a pixel's rows from its compact record.
struct LevelGeom {
    int rows_i;
    float inv_rows;   // 1/rows_i, to split a flat index into (v, u)
    float disp_u_i, disp_v_i;
    float inv_f_pyr;  // 2 tan(fovh/2) / cols_i        (pyramid xx/yy, reference FrontEnd.cpp:378)
    float inv_f_w;    // 1 / (cols_i / (2 tan(fovh/2)))  (warp xx/yy,    reference FrontEnd.cpp:874)
    float f_inv;      // cols_i / (2 tan(fovh/2))       (reference :537; it is f)
    float kph, inv_max_c, inv_max_d;
    int first;        // Warped := Pred iteration: xxWarped / yyWarped use the pyramid formula
};

// split a flat column-major index into (column u, row v)
__device__ __forceinline__ void split_index(const LevelGeom &g, int idx, float &fu, float &fv) {
    int u = (int)((float)idx * g.inv_rows);
    if (u * g.rows_i > idx) u--;
    if ((u + 1) * g.rows_i <= idx) u++;
    fu = float(u);
    fv = float(idx - u * g.rows_i);
}

// ---------------------------------------------------------------------------------------------
//  warp (reference FrontEnd.cpp:775-892), scatter part.  Normalisation happens when the
//  accumulators are read by the linearisation.
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ void solve_warp(const KArgs &a, int b, int L, LDS SolveShared &s, LDS ClusterShared &cs, int tid) {
    const int rows_i = a.lrows[L], cols_i = a.lcols[L], n = a.ln[L];
    const int G = cl_G(cs), rank = cl_rank(cs);
    const size_t rb = (size_t)cl_slot(cs) * a.n0;
    const auto dpred = as_global(pyr_level(a, b, 1, 0, L)), ipred = as_global(pyr_level(a, b, 1, 1, L));
    const auto acc_d = as_global(a.acc_d + rb);
    const auto acc_i = as_global(a.acc_i + rb);

    if (tid == 0) inverse4_cm(s.T, s.Tinv, s.dwork);  // T = T_odometry.inverse()  (:800)
    // a cluster's workgroups zero every G-th block. Agent-scope (write-through) stores: the cells are only ever touched by
    // agent-scope atomics and atomic loads after this, so the two hand-overs below need no fence (sf_cluster.h)
    // coarse levels (and every level of the reference-order build): the reference's float sums in the reference's order
    // (uniform by construction, made so for the compiler: branches around barriers must be scalar branches, see ordered_splat)
    const bool ordered = uniform_i(splat_ordered(L, n, G) ? 1 : 0) != 0;
    const bool lazy = ordered || uniform_i(splat_lazy_ok(rows_i, cols_i, G) ? 1 : 0) != 0;  // one workgroup: the splat zeroes / initialises the cells itself
    if (!lazy)
    for (int idx = tid + rank * SF_NT; idx < n; idx += SF_NT * G) {
        if (G > 1) {
            __hip_atomic_store(acc_d + idx, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(acc_i + idx, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            gst(acc_d, idx, 0ll);
            gst(acc_i, idx, 0ll);
        }
    }
    cluster_rendezvous(cs, tid);  // the accumulators are zero everywhere before anybody splats into them (and s.Tinv is set)

    SplatGeom g;
    g.f = float(cols_i) / (2.f * a.tan_half_fovh);
    g.disp_u_i = 0.5f * float(cols_i - 1);
    g.disp_v_i = 0.5f * float(rows_i - 1);
    g.cols_lim = 100 * (cols_i - 1);
    g.rows_lim = 100 * (rows_i - 1);
    g.rows_i = rows_i;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) g.T[r * 4 + c] = uniform_f(s.Tinv[r + 4 * c]);

    struct Src {
        gptr<const float> d, i;
        LevelCoord lc;
        __device__ __forceinline__ bool load(int v, int u, int idx, float &z, float &xr, float &yr, float &iw) const {
            z = gld(d, idx);
            iw = gld(i, idx);
            xr = coord_x(lc, u, z);  // xxPrediction / yyPrediction of the pyramid (:385-386)
            yr = coord_y(lc, v, z);
            return z != 0.f;
        }
    } src{dpred, ipred, level_coord(a, L)};
    if (ordered)
        ordered_splat(a, g, level_coord(a, L), rows_i, cols_i, src, acc_d, acc_i, ro_list_of(a, rb, b), s.win, tid, &a.state[b].prof[PF_ORDERED_FALLBACKS]);
    else
        tiled_splat(g, rows_i, cols_i, src, acc_d, acc_i, s.win, s.marks, tid, rank, G, lazy, &a.state[b].prof[PF_SPLAT_REPLAYS]);
    cluster_rendezvous(cs, tid);  // all atomics of the workgroup(s) performed: the linearisation reads the cells with atomic loads
}

// The end of a linearisation, whatever walked the pixels: the level's maxima of the raw pre-weights, its valid-pixel count and
// the two initial |res| sums, from every lane's share to the stream's state (through the cluster's gather when there is one).
__device__ __forceinline__ void lin_finish(LDS SolveShared &s, LDS ClusterShared &cs, int tid, float min_ec, float min_ed, int n_valid, double abs_c,
                                           double abs_d) {
    const int lane = tid & 63, wave = tid >> 6;
    const int G = cl_G(cs);
    // global max of the raw pre-weights (reference :505-509) and the valid-pixel count
    // min of non-negative floats through the max of (largest finite pattern - bits)
    const float max_c = wave_max_f32(__int_as_float(0x7f7fffff - __float_as_int(min_ec)));
    const float max_d = wave_max_f32(__int_as_float(0x7f7fffff - __float_as_int(min_ed)));
    n_valid = wave_sum_i32(n_valid);
    abs_c = wave_sum_f64(abs_c);
    abs_d = wave_sum_f64(abs_d);
    if (lane == 0) {
        s.redf[wave][0] = max_c;
        s.redf[wave][1] = max_d;
        s.redi[wave] = n_valid;
        s.red[wave][0] = abs_c;
        s.red[wave][1] = abs_d;
    }
    __syncthreads();
    // this workgroup's partial results -> payload words; every workgroup of the cluster then receives all of them and
    // combines them in rank order (maxima, counts and the fixed-point sums are order free; the two fp64 sums are added in
    // the same order everywhere). The gather also is the barrier behind which the records may be read by everybody.
    enum { W_TC = 0, W_TD, W_NV, W_AC, W_AD = W_AC + 2, W_LIN_WORDS = W_AD + 2 };
    if (tid == 0) {
        int tc = 0, td = 0, nv = 0;  // transformed minima, see above
        double ac = 0.0, ad = 0.0;
        for (int w = 0; w < SF_NW; w++) {
            tc = max(tc, __float_as_int(s.redf[w][0]));
            td = max(td, __float_as_int(s.redf[w][1]));
            nv += s.redi[w];
            ac += s.red[w][0];
            ad += s.red[w][1];
        }
        cs.in[W_TC] = (unsigned)tc;
        cs.in[W_TD] = (unsigned)td;
        cs.in[W_NV] = (unsigned)nv;
        put_f64(&cs.in[W_AC], ac);
        put_f64(&cs.in[W_AD], ad);
    }
    const int n_words = (int)W_LIN_WORDS;
    cluster_gather(cs, n_words, tid, true);
    if (tid == 0) {
        int tc = 0, td = 0, nv = 0;
        double ac = 0.0, ad = 0.0;
        for (int p = 0; p < G; p++) {
            const LDS unsigned *w = &cs.all[p * n_words];
            tc = max(tc, (int)w[W_TC]);
            td = max(td, (int)w[W_TD]);
            nv += (int)w[W_NV];
            ac += get_f64(&w[W_AC]);
            ad += get_f64(&w[W_AD]);
        }
        s.init_abs_c = ac;
        s.init_abs_d = ad;
        const float mc = sqrtf(1.f / (1.f + __int_as_float(0x7f7fffff - tc)));    // = max over validPixels of the raw weights_c
        const float md = sqrtf(1.f / (0.01f + __int_as_float(0x7f7fffff - td)));  //   "    weights_d (reference :494-509)
        s.n_valid = nv;
        s.inv_max_c = (nv > 0) ? 1.f / mc : 0.f;
        s.inv_max_d = (nv > 0) ? 1.f / md : 0.f;
        if (nv == 0) s.status |= SF_STATUS_EMPTY_LEVEL;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
//  linearise: calculateCoord + calculateDerivatives + computeWeights (raw) + computeSegPrior
// ---------------------------------------------------------------------------------------------
// Tile geometry: TILE_V x TILE_U centre pixels (TILE_CPX per lane) + a 1-pixel halo.  The loads of
// tile t+1 (halo elements + the centre pixels' coordinates / labels) are issued into registers before
// tile t is evaluated from LDS, so the global-memory latency overlaps the stencil arithmetic.
#if !SF_LIN_STRIPS
#define TILE_CPX 2
#define TILE_EPT ((TILE_N + SF_NT - 1) / SF_NT)  // halo-tile elements per lane

__device__ __noinline__ void solve_linearise(const KArgs &a, int b, int L, bool first, LDS SolveShared &s, LDS ClusterShared &cs, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int rows_i = a.lrows[L], cols_i = a.lcols[L], o = a.loff[L];
    const int G = cl_G(cs), rank = cl_rank(cs);  // a cluster's workgroups take every G-th tile
    const size_t sb = (size_t)b * a.n_tot, rb = (size_t)cl_slot(cs) * a.n0;
    const auto dnew = as_global(pyr_level(a, b, 0, 0, L)), inew = as_global(pyr_level(a, b, 0, 1, L));
    const auto dpred = as_global(pyr_level(a, b, 1, 0, L)), ipred = as_global(pyr_level(a, b, 1, 1, L));
    const auto acc_d = as_global((const long long *)a.acc_d + rb), acc_i = as_global((const long long *)a.acc_i + rb);
    const auto labels = as_global((const uint8_t *)a.labels + sb + o);
    gptr<float> rec[R_COUNT];
#pragma unroll
    for (int q = 0; q < R_COUNT; q++) rec[q] = as_global(a.rec[q] + rb);
    const auto rec_lab = as_global(a.rec_lab + rb);
    const bool seg = a.p.segmentation_enabled != 0;
    const bool dbg = a.p.debug_planes != 0;
    const bool ordered = splat_ordered(L, a.ln[L], G);  // what solve_warp left in the accumulator cells of this level
    if (tid == 0) s.first = first ? 1 : 0;

    const float f = float(cols_i) / (2.f * a.tan_half_fovh);
    const float inv_f_w = 1.f / f;  // the warp's 1/f (reference FrontEnd.cpp:874), not the pyramid's
    const float disp_u_i = 0.5f * float(cols_i - 1);
    const float disp_v_i = 0.5f * float(rows_i - 1);
    const float epsilon_intensity = 1e-6f, epsilon_depth = 0.005f;

    // The raw pre-weights w = sqrt(1 / (eps + e)) are needed only through their image maximum (:505-509), and w is a
    // monotonic (non-increasing) function of e in float arithmetic too -- every step of it is -- so max w = w(min e),
    // bit for bit: the pass tracks min e and evaluates the division and the square root once, at the end.
    float min_ec = 3.0e38f, min_ed = 3.0e38f;
    double abs_c = 0.0, abs_d = 0.0;  // initial |res| = |B| sums (reference :588-590), scaled by 1/max afterwards
    int n_valid = 0;

    const int tiles_v = (rows_i + TILE_V - 1) / TILE_V, tiles_u = (cols_i + TILE_U - 1) / TILE_U;
    const int n_tiles = tiles_v * tiles_u;

    // prefetch registers (plain local arrays + a macro: a lambda capturing a struct kept it in scratch memory)
    float pf_dn[TILE_EPT], pf_in[TILE_EPT];
    long long pf_ad[TILE_EPT], pf_ai[TILE_EPT];
    int pf_lab[TILE_CPX];
#define LIN_PREFETCH(TILE_IDX)                                                                                        \
    do {                                                                                                              \
        const int ptv0 = ((TILE_IDX) % tiles_v) * TILE_V, ptu0 = ((TILE_IDX) / tiles_v) * TILE_U;                     \
        _Pragma("unroll") for (int q = 0; q < TILE_EPT; q++) {                                                       \
            const int e = tid + q * SF_NT;                                                                            \
            const int lu = e / TILE_LV, lv = e - lu * TILE_LV;                                                        \
            const int v = ptv0 - 1 + lv, u = ptu0 - 1 + lu;                                                           \
            const bool inside = e < TILE_N && v >= 0 && v < rows_i && u >= 0 && u < cols_i;                           \
            const int idx = inside ? v + u * rows_i : 0;                                                              \
            pf_dn[q] = gld(dnew, idx);                                                                                    \
            pf_in[q] = gld(inew, idx);                                                                                    \
            if (first) { /* Warped := Pred (reference FrontEnd.cpp:1103-1110): carry the float bits in pf_ad */      \
                const unsigned lo = __float_as_uint(gld(dpred, idx)), hi = __float_as_uint(gld(ipred, idx));                    \
                pf_ad[q] = (long long)(((unsigned long long)hi << 32) | lo);                                          \
            } else {                                                                                                  \
                pf_ad[q] = gld_agent_i64(acc_d, idx);                \
                pf_ai[q] = gld_agent_i64(acc_i, idx);                \
            }                                                                                                         \
        }                                                                                                             \
        if (seg) {                                                                                                    \
            _Pragma("unroll") for (int k = 0; k < TILE_CPX; k++) {                                                   \
                const int v = ptv0 + lane, u = ptu0 + wave + k * SF_NW;                                               \
                pf_lab[k] = (int)gld(labels, (v < rows_i && u < cols_i) ? v + u * rows_i : 0);                             \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
    if (rank < n_tiles) LIN_PREFETCH(rank);

    for (int tile = rank; tile < n_tiles; tile += G) {
        const int tv0 = (tile % tiles_v) * TILE_V, tu0 = (tile / tiles_v) * TILE_U;
        __syncthreads();  // previous tile consumed (and the bin initialisation above)
#pragma unroll
        for (int q = 0; q < TILE_EPT; q++) {
            const int e = tid + q * SF_NT;
            if (e >= TILE_N) continue;
            const int lu = e / TILE_LV, lv = e - lu * TILE_LV;
            const int v = tv0 - 1 + lv, u = tu0 - 1 + lu;
            const bool inside = (v >= 0 && v < rows_i && u >= 0 && u < cols_i);
            float dn = 0.f, in_ = 0.f, dw = 0.f, iw = 0.f;
            if (inside) {
                dn = pf_dn[q];
                in_ = pf_in[q];
                if (first) {
                    dw = __uint_as_float((unsigned)((unsigned long long)pf_ad[q] & 0xffffffffu));
                    iw = __uint_as_float((unsigned)((unsigned long long)pf_ad[q] >> 32));
                } else if (pf_ai[q] != 0) {  // normalise the warp accumulators (reference :876-881); touched <=> sum(w) > 0
                    if (ordered)
                        ro_unpack_cell(pf_ad[q], dw, iw);  // already divided, in the reference's order (ro_splat)
                    else
                        normalise_acc(pf_ad[q], pf_ai[q], dw, iw);
                }
            }
            const bool nul = !(inside && (dn != 0.f) && (dw != 0.f));
            s.lt.t_null[e] = nul ? 1 : 0;
            s.lt.t_D[e] = nul ? 0.f : 0.5f * (dn + dw);
            s.lt.t_I[e] = 0.5f * (in_ + iw);
            s.lt.t_dn[e] = dn;
            s.lt.t_in[e] = in_;
            s.lt.t_dw[e] = dw;
            s.lt.t_iw[e] = iw;
        }
        static_assert(TILE_CPX == 2, "two centre pixels per lane");
        const int c_lab0 = seg ? pf_lab[0] : 0, c_lab1 = seg ? pf_lab[1] : 0;  // scalars: indexing by the loop
                                                                                 // counter below would go to scratch
        __syncthreads();
        if (tile + G < n_tiles) LIN_PREFETCH(tile + G);  // in flight while this tile is evaluated

#pragma unroll 1
        for (int k = 0; k < TILE_CPX; k++) {
            const int lv = lane + 1, lu = wave + k * SF_NW + 1;
            const int v = tv0 + lv - 1, u = tu0 + lu - 1;
            const bool inside = (v < rows_i && u < cols_i);
            const int e = lv + lu * TILE_LV;
            const int idx = v + u * rows_i;
            bool valid = false;
            int lab = SF_NC;
            float ddt_ = 0.f;
            if (inside) {
                const float dn = s.lt.t_dn[e], dw = s.lt.t_dw[e];
                const bool nul = s.lt.t_null[e] != 0;
                const float dct_ = s.lt.t_in[e] - s.lt.t_iw[e];
                ddt_ = dn - dw;
                lab = seg ? (k ? c_lab1 : c_lab0) : ((dn != 0.f) ? 0 : SF_NC);
                // validPixels (reference :415-427). Departure: a point warped BEHIND the camera that still projects into the
                // image gives a negative warped depth, which the reference keeps in validPixels (:816-823 has no depth test);
                // here such a pixel is left out everywhere (counts, sums, passes), because the sign of the stored warped
                // depth is what marks membership for the passes. It needs a diverged pose to happen at all.
#if SF_REFORDER && SF_RO_BEHIND
                valid = !nul && (u != 0) && (v != 0) && (u != cols_i - 1) && (v != rows_i - 1);  // the reference's rule, :415-427
#else
                valid = !nul && (dw > 0.f) && (u != 0) && (v != 0) && (u != cols_i - 1) && (v != rows_i - 1);
#endif
                float dcu_ = 0.f, dcv_ = 0.f, ddu_ = 0.f, ddv_ = 0.f;
                if (valid) {
                    const int eL = e - TILE_LV, eR = e + TILE_LV, eU = e - 1, eD = e + 1;  // (v,u-1) (v,u+1) (v-1,u) (v+1,u)
                    const float Dc = s.lt.t_D[e], Ic = s.lt.t_I[e];
                    // rx / ry weights of this pixel and of its left / upper neighbour (reference :448-462)
                    const float rx_c = (u < cols_i - 1) ? fabsf(s.lt.t_D[eR] - Dc) + epsilon_depth : 1.f;
                    const float rxi_c = (u < cols_i - 1) ? fabsf(s.lt.t_I[eR] - Ic) + epsilon_intensity : 1.f;
                    const float ry_c = (v < rows_i - 1) ? fabsf(s.lt.t_D[eD] - Dc) + epsilon_depth : 1.f;
                    const float ryi_c = (v < rows_i - 1) ? fabsf(s.lt.t_I[eD] - Ic) + epsilon_intensity : 1.f;
                    const bool nulL = s.lt.t_null[eL] != 0, nulU = s.lt.t_null[eU] != 0;
                    const float rx_l = nulL ? 1.f : fabsf(Dc - s.lt.t_D[eL]) + epsilon_depth;
                    const float rxi_l = nulL ? 1.f : fabsf(Ic - s.lt.t_I[eL]) + epsilon_intensity;
                    const float ry_u = nulU ? 1.f : fabsf(Dc - s.lt.t_D[eU]) + epsilon_depth;
                    const float ryi_u = nulU ? 1.f : fabsf(Ic - s.lt.t_I[eU]) + epsilon_intensity;
                    dcu_ = (rxi_l * (s.lt.t_I[eR] - Ic) + rxi_c * (Ic - s.lt.t_I[eL])) / (rxi_c + rxi_l);
                    ddu_ = (rx_l * (s.lt.t_D[eR] - Dc) + rx_c * (Dc - s.lt.t_D[eL])) / (rx_c + rx_l);
                    dcv_ = (ryi_u * (s.lt.t_I[eD] - Ic) + ryi_c * (Ic - s.lt.t_I[eU])) / (ryi_c + ryi_u);
                    ddv_ = (ry_u * (s.lt.t_D[eD] - Dc) + ry_c * (Dc - s.lt.t_D[eU])) / (ry_c + ry_u);
                    // raw pre-weights (reference :487-502): only their global maxima are needed here
                    const float error_l_c = 10.f * (fabsf(dct_) + fabsf(dcu_) + fabsf(dcv_));
                    const float error_l_d = 200.f * (fabsf(ddt_) + fabsf(ddu_) + fabsf(ddv_));
                    min_ec = (error_l_c < min_ec) ? error_l_c : min_ec;
                    min_ed = (error_l_d < min_ed) ? error_l_d : min_ed;
                    abs_c += (double)(vrsq(1.f + error_l_c) * fabsf(dct_));  // IRLS-side quantity: 1-ulp rsq like the passes
                    abs_d += (double)(vrsq(0.01f + error_l_d) * fabsf(ddt_));
                    n_valid++;
                }
                // the SIGN carries validPixels (valid => dw > 0): the passes need no label plane for it. A NEGATIVE warped depth
                // (a point behind the camera that still projects into the image: a diverged pose) stays negative = not valid;
                // the segmentation prior then sees its magnitude (solve_seg_prior), the one place where this differs from the
                // reference, which carries such a pixel through with its sign
#if SF_REFORDER && SF_RO_BEHIND
                gst(rec[R_DW], idx, dw);  // validPixels rides in the label plane of this build, the sign is the warp's
#else
                gst(rec[R_DW], idx, valid ? dw : -fabsf(dw));
#endif
                gst(rec[R_DCU], idx, dcu_);
                gst(rec[R_DCV], idx, dcv_);
                gst(rec[R_DCT], idx, (valid || dbg) ? dct_ : 0.f);  // 0 outside validPixels: the passes run branch-free over every pixel
                gst(rec[R_DDU], idx, ddu_);
                gst(rec[R_DDV], idx, ddv_);
                if (seg || dbg || SF_REFORDER) gst(rec_lab, idx, valid ? (uint8_t)(seg ? lab : 0) : (uint8_t)SF_INVALID_LABEL);
                if (dbg) {
                    float d_i = 0.f, x_i = 0.f, y_i = 0.f, xw = 0.f, yw = 0.f;
                    const LevelCoord lcd = level_coord(a, L);
                    if (first) {  // xxWarped := xxPrediction (:1107-1108)
                        xw = coord_x(lcd, u, dw);
                        yw = coord_y(lcd, v, dw);
                    } else if (dw != 0.f) {
                        xw = (float(u) - disp_u_i) * dw * inv_f_w;
                        yw = (float(v) - disp_v_i) * dw * inv_f_w;
                    }
                    if (!nul) {
                        d_i = s.lt.t_D[e];
                        x_i = 0.5f * (coord_x(lcd, u, dn) + xw);
                        y_i = 0.5f * (coord_y(lcd, v, dn) + yw);
                    }
                    a.rec_null[rb + idx] = nul ? 1 : 0;
                    const size_t q = sb + o + idx;
                    a.dbg_warped[0][q] = dw;
                    a.dbg_warped[1][q] = s.lt.t_iw[e];
                    a.dbg_warped[2][q] = xw;
                    a.dbg_warped[3][q] = yw;
                    a.dbg_inter[0][q] = d_i;
                    a.dbg_inter[1][q] = s.lt.t_I[e];
                    a.dbg_inter[2][q] = x_i;
                    a.dbg_inter[3][q] = y_i;
                }
            }
        }
    }

    lin_finish(s, cs, tid, min_ec, min_ed, n_valid, abs_c, abs_d);
}

#undef LIN_PREFETCH
#endif  // !SF_LIN_STRIPS

// ---------------------------------------------------------------------------------------------
//  linearise, one-workgroup builds: the same arithmetic on REGISTER STRIPS.
//  A wave owns LS_ROWS consecutive rows of the (column-major) level -- lane l holds row v0 - 1 + l, lanes 0 and 63 are the
//  halo rows -- and sweeps the columns: every lane keeps the Inter depth / intensity / Null of the columns u - 1, u, u + 1 in
//  registers, the upper and lower neighbours of column u come from the adjacent lanes over the DPP network (wave_shr /
//  wave_shl, as in the pyramid), and the loads of column u + 4 are issued while column u is evaluated. No LDS, no barrier,
//  every cell is normalised once by the lane that loads it (the tiles normalised 660 halo elements per 512 pixels, staged
//  seven LDS words each and paid two barriers per tile: 5.6 wave instructions per pixel, of which the stencil is 1.5).
//  Few rows (the coarse levels) leave waves over: the columns are then cut into as many segments as waves are free.
//  Bit for bit the records, maxima and counts of the tiled form; the two fp64 sums of the initial |res| add the same terms
//  in another order.
// ---------------------------------------------------------------------------------------------
// computeSegPrior rides in the sweep (the product builds: its sums are integers, whoever adds them): the pass of its own read
// 9 bytes per pixel again -- 2.7 % of the full solver's HBM traffic. The reference-order build keeps ro_seg_prior.
#ifndef SF_LIN_FUSED_PRIOR
#define SF_LIN_FUSED_PRIOR (SF_LIN_STRIPS && !SF_REFORDER)
#endif
__device__ __forceinline__ void seg_prior_begin(LDS SolveShared &s, int tid);
__device__ __forceinline__ void seg_prior_finish(LDS SolveShared &s, LDS ClusterShared &cs, int tid);
#if SF_LIN_STRIPS
// (The first linearisation of a frame, the debug planes' stores and segmentation -- the label load, the prior's sums -- are
// template parameters: a memory operation the sweep may or may not issue makes every wait for a load a full one, the compiler
// counts the operations that are certain to follow it.)
template <bool DBG, bool FIRST, bool SEG>
__device__ __noinline__ void solve_linearise_strips(const KArgs &a, int b, int L, LDS SolveShared &s, LDS ClusterShared &cs, int tid) {
    constexpr bool first = FIRST;
    const int lane = tid & 63, wave = uniform_i(tid >> 6);  // (scalar: the items, the column range and the sweep's loop control with it)
    const int rows_i = uniform_i(a.lrows[L]), cols_i = uniform_i(a.lcols[L]), o = uniform_i(a.loff[L]);
    const size_t sb = (size_t)b * a.n_tot, rb = (size_t)cl_slot(cs) * a.n0;
    const auto dnew = as_global(pyr_level(a, b, 0, 0, L)), inew = as_global(pyr_level(a, b, 0, 1, L));
    const auto dpred = as_global(pyr_level(a, b, 1, 0, L)), ipred = as_global(pyr_level(a, b, 1, 1, L));
    const auto acc_d = as_global((const long long *)a.acc_d + rb), acc_i = as_global((const long long *)a.acc_i + rb);
    const auto labels = as_global((const uint8_t *)a.labels + sb + o);
    gptr<float> rec[R_COUNT];
#pragma unroll
    for (int q = 0; q < R_COUNT; q++) rec[q] = as_global(a.rec[q] + rb);
    const auto rec_lab = as_global(a.rec_lab + rb);
    constexpr bool seg = SEG;  // (a template parameter like the two others: no label load, no prior sums without segmentation)
    constexpr bool dbg = DBG;
    const bool ordered = uniform_i(splat_ordered(L, a.ln[L], 1) ? 1 : 0) != 0;  // what solve_warp left in the accumulator cells of this level
    if (tid == 0) s.first = first ? 1 : 0;
    constexpr bool fuse_prior = SF_LIN_FUSED_PRIOR && seg;
    if (fuse_prior) seg_prior_begin(s, tid);  // (uniform; a barrier)
    const float kz = uniform_f(a.p.kz);
    int pr_cur = 0, pr_size = 0, pr_nn = 0, pr_valid = 0;  // running totals for the label of this lane's last pixel (solve_seg_prior)
    long long pr_sum = 0;

    const float f = float(cols_i) / (2.f * a.tan_half_fovh);
    const float inv_f_w = 1.f / f;  // the warp's 1/f (reference FrontEnd.cpp:874), not the pyramid's
    const float disp_u_i = 0.5f * float(cols_i - 1);
    const float disp_v_i = 0.5f * float(rows_i - 1);
    const float epsilon_intensity = 1e-6f, epsilon_depth = 0.005f;

    float min_ec = 3.0e38f, min_ed = 3.0e38f;  // max w = w(min e), see solve_linearise
    double abs_c = 0.0, abs_d = 0.0;
    int n_valid = 0;

    const int n_strips = (rows_i + LS_ROWS - 1) / LS_ROWS;
    // as many column segments as it takes for the items to go round the waves evenly: SF_NW / gcd(strips, SF_NW)
    int g_ = n_strips, h_ = SF_NW;
    while (h_) {
        const int t_ = g_ % h_;
        g_ = h_;
        h_ = t_;
    }
    const int n_seg = min(cols_i, SF_NW / g_);
    const int seg_w = (cols_i + n_seg - 1) / n_seg;
    const int n_items = n_strips * n_seg;

    for (int item = wave; item < n_items; item += SF_NW) {  // (wave-uniform)
        const int strip = item % n_strips, sg = item / n_strips;
        const int ub = sg * seg_w, ue = min(cols_i, ub + seg_w);
        if (ub >= ue) continue;
        const int v = strip * LS_ROWS - 1 + lane;  // this lane's row
        const bool row_in = v >= 0 && v < rows_i;
        const bool owner = lane >= 1 && lane <= LS_ROWS && v < rows_i;  // lanes 0 and 63 only lend their row to the neighbours
        const bool v_inner = owner && v != 0 && v != rows_i - 1;

        // loads in flight (a ring of three columns) and the three committed columns around the one being evaluated
        float r_dn[3], r_in[3];
        long long r_ad[3], r_ai[3];
        int r_lab[3];
        float wD[3], wI[3], w_dn[3], w_dw[3], w_in[3], w_iw[3];
        int wN[3];  // bit 0: Null; bits 8..: the pixel's label
#define LS_LOAD(S, COL)                                                                                             \
    do {                                                                                                            \
        const int cc_ = (COL);                                                                                      \
        const int idx_ = (row_in && cc_ >= 0 && cc_ < cols_i) ? v + cc_ * rows_i : 0;                               \
        r_dn[S] = gld(dnew, idx_);                                                                                  \
        r_in[S] = gld(inew, idx_);                                                                                  \
        if (first) { /* Warped := Pred (reference FrontEnd.cpp:1103-1110): carry the float bits in r_ad */          \
            const unsigned lo_ = __float_as_uint(gld(dpred, idx_)), hi_ = __float_as_uint(gld(ipred, idx_));        \
            r_ad[S] = (long long)(((unsigned long long)hi_ << 32) | lo_);                                           \
            r_ai[S] = 0;                                                                                            \
        } else {                                                                                                    \
            r_ad[S] = gld_agent_i64(acc_d, idx_);                                                                   \
            r_ai[S] = gld_agent_i64(acc_i, idx_);                                                                   \
        }                                                                                                           \
        r_lab[S] = seg ? (int)gld(labels, idx_) : 0;                                                                \
    } while (0)
#define LS_COMMIT(S, COL)                                                                                           \
    do { /* branch-free: a loaded register consumed on one side of a divergent branch only costs the waits their precision */ \
        const int cc_ = (COL);                                                                                      \
        const bool in_ = row_in && cc_ >= 0 && cc_ < cols_i;                                                        \
        float dw_, iw_;                                                                                             \
        if (first) {                                                                                                \
            dw_ = __uint_as_float((unsigned)((unsigned long long)r_ad[S] & 0xffffffffu));                           \
            iw_ = __uint_as_float((unsigned)((unsigned long long)r_ad[S] >> 32));                                   \
        } else { /* normalise the warp accumulators (reference :876-881); touched <=> sum(w) > 0 */                 \
            if (ordered)                                                                                            \
                ro_unpack_cell(r_ad[S], dw_, iw_);                                                                  \
            else                                                                                                    \
                normalise_acc(r_ad[S], r_ai[S], dw_, iw_);                                                          \
            const bool touched_ = r_ai[S] != 0;                                                                     \
            dw_ = touched_ ? dw_ : 0.f;                                                                             \
            iw_ = touched_ ? iw_ : 0.f;                                                                             \
        }                                                                                                           \
        const float dn_ = in_ ? r_dn[S] : 0.f, i_ = in_ ? r_in[S] : 0.f;                                            \
        dw_ = in_ ? dw_ : 0.f;                                                                                      \
        iw_ = in_ ? iw_ : 0.f;                                                                                      \
        const bool nul_ = !(in_ && (dn_ != 0.f) && (dw_ != 0.f));                                                   \
        int lab_ = r_lab[S]; /* pinned here: hoisted into the loop's latch (as the compiler did: the expression recurs behind */ \
        asm volatile("" : "+v"(lab_)); /* the loop) it waited there for the youngest load of the sweep */               \
        wN[S] = (nul_ ? 1 : 0) | (lab_ << 8);                                                                       \
        wD[S] = nul_ ? 0.f : 0.5f * (dn_ + dw_);                                                                    \
        wI[S] = 0.5f * (i_ + iw_);                                                                                  \
        w_dn[S] = dn_;                                                                                              \
        w_in[S] = i_;                                                                                               \
        w_dw[S] = dw_;                                                                                              \
        w_iw[S] = iw_;                                                                                              \
    } while (0)
        // column c of the item (counted from ub - 1) lives in slot c % 3 of both rings
        LS_LOAD(0, ub - 1);
        LS_LOAD(1, ub);
        LS_LOAD(2, ub + 1);
        LS_COMMIT(0, ub - 1);
        LS_LOAD(0, ub + 2);
        LS_COMMIT(1, ub);
        LS_LOAD(1, ub + 3);
#define LS_PRIOR_FLUSH()                                           \
    do {                                                           \
        if (pr_size) {                                             \
            lds_add(&s.prior_size[pr_cur], pr_size);               \
            if (pr_nn) {                                           \
                lds_add(&s.prior_nonnull[pr_cur], pr_nn);          \
                lds_add(&s.prior_sum[pr_cur], pr_sum);             \
            }                                                      \
            if (pr_valid) lds_add(&s.valid_cnt[pr_cur], pr_valid); \
        }                                                          \
    } while (0)
#define LS_COLUMN(U_, J_)                                                                                            \
    do {                                                                                                            \
        const int u = (U_);                                                                                         \
        const int sl = (J_) % 3, sc = ((J_) + 1) % 3, sr = ((J_) + 2) % 3; /* slots of the columns u - 1, u, u + 1 */ \
        LS_COMMIT(sr, u + 1);                                                                                       \
        LS_LOAD(sr, u + 4);  /* (unconditionally: a load the sweep may or may not issue would make every wait a full one) */\
        /* the rows above and below, from the neighbouring lanes: every lane of the wave takes part */              \
        const float Dc = wD[sc], Ic = wI[sc];                                                                       \
        const float D_up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Dc), 0x138, 0xf, 0xf, false));  /* wave_shr: row v - 1 */\
        const float D_dn = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Dc), 0x130, 0xf, 0xf, false));  /* wave_shl: row v + 1 */\
        const float I_up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Ic), 0x138, 0xf, 0xf, false));\
        const float I_dn = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(Ic), 0x130, 0xf, 0xf, false));\
        const int N_up = __builtin_amdgcn_update_dpp(1, wN[sc], 0x138, 0xf, 0xf, false);                            \
        if (owner) {                                                                                                \
            const int idx = v + u * rows_i;                                                                         \
            const float dn = w_dn[sc], dw = w_dw[sc];                                                               \
            const bool nul = (wN[sc] & 1) != 0;                                                                     \
            const float dct_ = w_in[sc] - w_iw[sc];                                                                 \
            const float ddt_ = dn - dw;                                                                             \
            const int lab = seg ? (wN[sc] >> 8) : ((dn != 0.f) ? 0 : SF_NC);                                        \
            /* validPixels (reference :415-427), with the product's rule for points behind the camera (solve_linearise) */\
            const bool valid = v_inner && !nul && (LS_RO_BEHIND || dw > 0.f) && (u != 0) && (u != cols_i - 1);      \
            float dcu_ = 0.f, dcv_ = 0.f, ddu_ = 0.f, ddv_ = 0.f;                                                   \
            if (valid) {  /* (an inner pixel: all four neighbours are inside the image) */                          \
                const float D_l = wD[sl], I_l = wI[sl], D_r = wD[sr], I_r = wI[sr];                                 \
                /* rx / ry weights of this pixel and of its left / upper neighbour (reference :448-462) */          \
                const float rx_c = fabsf(D_r - Dc) + epsilon_depth;                                                 \
                const float rxi_c = fabsf(I_r - Ic) + epsilon_intensity;                                            \
                const float ry_c = fabsf(D_dn - Dc) + epsilon_depth;                                                \
                const float ryi_c = fabsf(I_dn - Ic) + epsilon_intensity;                                           \
                const bool nulL = (wN[sl] & 1) != 0, nulU = (N_up & 1) != 0;                                        \
                const float rx_l = nulL ? 1.f : fabsf(Dc - D_l) + epsilon_depth;                                    \
                const float rxi_l = nulL ? 1.f : fabsf(Ic - I_l) + epsilon_intensity;                               \
                const float ry_u = nulU ? 1.f : fabsf(Dc - D_up) + epsilon_depth;                                   \
                const float ryi_u = nulU ? 1.f : fabsf(Ic - I_up) + epsilon_intensity;                              \
                dcu_ = (rxi_l * (I_r - Ic) + rxi_c * (Ic - I_l)) / (rxi_c + rxi_l);                                 \
                ddu_ = (rx_l * (D_r - Dc) + rx_c * (Dc - D_l)) / (rx_c + rx_l);                                     \
                dcv_ = (ryi_u * (I_dn - Ic) + ryi_c * (Ic - I_up)) / (ryi_c + ryi_u);                               \
                ddv_ = (ry_u * (D_dn - Dc) + ry_c * (Dc - D_up)) / (ry_c + ry_u);                                   \
                /* raw pre-weights (reference :487-502): only their global maxima are needed here */                \
                const float error_l_c = 10.f * (fabsf(dct_) + fabsf(dcu_) + fabsf(dcv_));                           \
                const float error_l_d = 200.f * (fabsf(ddt_) + fabsf(ddu_) + fabsf(ddv_));                          \
                min_ec = (error_l_c < min_ec) ? error_l_c : min_ec;                                                 \
                min_ed = (error_l_d < min_ed) ? error_l_d : min_ed;                                                 \
                abs_c += (double)(vrsq(1.f + error_l_c) * fabsf(dct_));  /* IRLS-side quantity: 1-ulp rsq like the passes */\
                abs_d += (double)(vrsq(0.01f + error_l_d) * fabsf(ddt_));                                           \
                n_valid++;                                                                                          \
            }                                                                                                       \
            if (fuse_prior && (wN[sc] >> 8) != SF_NC) { /* computeSegPrior's sums (solve_seg_prior: the same integers) */\
                const int pl_ = wN[sc] >> 8;                                                                        \
                if (pl_ != pr_cur) {                                                                                \
                    LS_PRIOR_FLUSH();                                                                               \
                    pr_cur = pl_;                                                                                   \
                    pr_size = pr_nn = pr_valid = 0;                                                                 \
                    pr_sum = 0;                                                                                     \
                }                                                                                                   \
                pr_size++;                                                                                          \
                const float dwa_ = fabsf(dw);                                                                       \
                if (dn != 0.f && dwa_ != 0.f) { /* not Null */                                                      \
                    pr_nn++;                                                                                        \
                    pr_sum += to_fix(1.f - kz * fabsf(dn - dwa_), FIX_RES, 1.0e6f);                                 \
                }                                                                                                   \
                pr_valid += valid ? 1 : 0;                                                                          \
            }                                                                                                       \
            /* the SIGN carries validPixels (solve_linearise); LS_RO_BEHIND: the label plane does, the sign is the warp's */\
            gst(rec[R_DW], idx, (LS_RO_BEHIND || valid) ? dw : -fabsf(dw));                                         \
            gst(rec[R_DCU], idx, dcu_);                                                                             \
            gst(rec[R_DCV], idx, dcv_);                                                                             \
            gst(rec[R_DCT], idx, (valid || dbg) ? dct_ : 0.f);  /* 0 outside validPixels: the passes run branch-free over every pixel */\
            gst(rec[R_DDU], idx, ddu_);                                                                             \
            gst(rec[R_DDV], idx, ddv_);                                                                             \
            if (seg || dbg || SF_REFORDER) gst(rec_lab, idx, valid ? (uint8_t)(seg ? lab : 0) : (uint8_t)SF_INVALID_LABEL);\
            if (dbg) {                                                                                              \
                float d_i = 0.f, x_i = 0.f, y_i = 0.f, xw = 0.f, yw = 0.f;                                          \
                const LevelCoord lcd = level_coord(a, L);                                                           \
                if (first) {  /* xxWarped := xxPrediction (:1107-1108) */                                           \
                    xw = coord_x(lcd, u, dw);                                                                       \
                    yw = coord_y(lcd, v, dw);                                                                       \
                } else if (dw != 0.f) {                                                                             \
                    xw = (float(u) - disp_u_i) * dw * inv_f_w;                                                      \
                    yw = (float(v) - disp_v_i) * dw * inv_f_w;                                                      \
                }                                                                                                   \
                if (!nul) {                                                                                         \
                    d_i = Dc;                                                                                       \
                    x_i = 0.5f * (coord_x(lcd, u, dn) + xw);                                                        \
                    y_i = 0.5f * (coord_y(lcd, v, dn) + yw);                                                        \
                }                                                                                                   \
                a.rec_null[rb + idx] = nul ? 1 : 0;                                                                 \
                const size_t q = sb + o + idx;                                                                      \
                a.dbg_warped[0][q] = dw;                                                                            \
                a.dbg_warped[1][q] = w_iw[sc];                                                                      \
                a.dbg_warped[2][q] = xw;                                                                            \
                a.dbg_warped[3][q] = yw;                                                                            \
                a.dbg_inter[0][q] = d_i;                                                                            \
                a.dbg_inter[1][q] = Ic;                                                                             \
                a.dbg_inter[2][q] = x_i;                                                                            \
                a.dbg_inter[3][q] = y_i;                                                                            \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
        // whole triples of columns in a loop without an exit in its body (the waits for the loads in flight stay exact), the
        // last one or two columns behind it
        int u0 = ub;
        for (; u0 + 3 <= ue; u0 += 3) {
            LS_COLUMN(u0, 0);
            LS_COLUMN(u0 + 1, 1);
            LS_COLUMN(u0 + 2, 2);
        }
        if (u0 < ue) LS_COLUMN(u0, 0);
        if (u0 + 1 < ue) LS_COLUMN(u0 + 1, 1);
#undef LS_COLUMN
#undef LS_LOAD
#undef LS_COMMIT
    }
    if (fuse_prior) LS_PRIOR_FLUSH();  // (the barriers of lin_finish stand between these atomics and seg_prior_finish)
#undef LS_PRIOR_FLUSH
    lin_finish(s, cs, tid, min_ec, min_ed, n_valid, abs_c, abs_d);
}
#endif  // SF_LIN_STRIPS

// ---------------------------------------------------------------------------------------------
//  computeSegPrior (reference SegmentationBackground.cpp:53-103): per cluster the pixel count, the count of non-Null
//  pixels, the sum of 1 - kz |ddt| over them -- and validPixels per cluster for the b-solve (:651). A streaming pass over the
//  level right after the linearisation: new depth, stored warped depth (its sign carries validPixels) and the label byte,
//  9 bytes per pixel. Each lane walks consecutive pixel pairs of a column band, where labels are coherent: it keeps running
//  totals for the label of its last pixel and flushes them with four LDS integer atomics when the label changes (the sums
//  are integers / Q32.32: exact, order free). The linearisation itself used to aggregate these per tile with wave ballots
//  and 64-bit DPP sums -- more instructions than the stencil.
// ---------------------------------------------------------------------------------------------
// the bins of computeSegPrior, zeroed (a barrier: nothing may flush into them before)
__device__ __forceinline__ void seg_prior_begin(LDS SolveShared &s, int tid) {
    if (tid < SF_NC) {
        s.prior_sum[tid] = 0;
        s.prior_size[tid] = 0;
        s.prior_nonnull[tid] = 0;
        s.valid_cnt[tid] = 0;
    }
    __syncthreads();
}

// ... and what follows their last flush (behind a barrier): the cluster's gather, b_prior and lambda_t_w per label
__device__ __forceinline__ void seg_prior_finish(LDS SolveShared &s, LDS ClusterShared &cs, int tid) {
    enum { W_PSUM = 0, W_PSIZE = 2 * SF_NC, W_PNN = W_PSIZE + SF_NC, W_VCNT = W_PNN + SF_NC, W_WORDS = W_VCNT + SF_NC };
    static_assert(W_WORDS <= SF_SYNC_WORDS, "payload of the prior rendezvous");
    if (tid < SF_NC) {
        put_i64(&cs.in[W_PSUM + 2 * tid], s.prior_sum[tid]);
        cs.in[W_PSIZE + tid] = (unsigned)s.prior_size[tid];
        cs.in[W_PNN + tid] = (unsigned)s.prior_nonnull[tid];
        cs.in[W_VCNT + tid] = (unsigned)s.valid_cnt[tid];
    }
    cluster_gather(cs, W_WORDS, tid);
    if (tid < SF_NC) {  // reference SegmentationBackground.cpp:84-102
        const int l = tid, G = cl_G(cs);
        long long psum = 0;
        int psize = 0, pnn = 0, vcnt = 0;
        for (int p = 0; p < G; p++) {
            const LDS unsigned *w = &cs.all[p * W_WORDS];
            psum += get_i64(&w[W_PSUM + 2 * l]);
            psize += (int)w[W_PSIZE + l];
            pnn += (int)w[W_PNN + l];
            vcnt += (int)w[W_VCNT + l];
        }
        s.valid_cnt[l] = vcnt;  // num_pix_label of the whole level (the b-solve's 1 / (2 (n + 1)))
        float bp = 0.f, lt = 0.f;
        if (psize != 0) {
            const float ratio = float(pnn) / float(psize);
            if (ratio < 0.1f) {
                lt = 0.1f;
                bp = -1.f;
            } else {
                lt = ratio;
                const float sum = (float)((double)psum * (1.0 / 4294967296.0));
                bp = std_max(-1.f, std_min(2.f, sum / pnn));
            }
        }
        s.b_prior[l] = bp;
        s.lambda_t_w[l] = lt;
    }
    __syncthreads();
}

__device__ __noinline__ void solve_seg_prior(const KArgs &a, int b, int L, LDS SolveShared &s, LDS ClusterShared &cs, int tid) {
    const int n = a.ln[L];
    const float kz = a.p.kz;
    const size_t sb = (size_t)b * a.n_tot + a.loff[L], rb = (size_t)cl_slot(cs) * a.n0;
    const auto dnew = uniform_ptr((gcfloat *)pyr_level(a, b, 0, 0, L));
    const auto dwp = uniform_ptr((gcfloat *)(a.rec[R_DW] + rb));
    const auto labp = uniform_ptr((gcu8 *)(a.labels + sb));
    seg_prior_begin(s, tid);
    int pb, pe;
    cluster_range(cs, n, 2, pb, pe);
    int cur = 0, c_size = 0, c_nn = 0, c_valid = 0;
    long long c_sum = 0;
    auto flush = [&]() {
        if (c_size) {
            lds_add(&s.prior_size[cur], c_size);
            if (c_nn) {
                lds_add(&s.prior_nonnull[cur], c_nn);
                lds_add(&s.prior_sum[cur], c_sum);
            }
            if (c_valid) lds_add(&s.valid_cnt[cur], c_valid);
        }
    };
    for (int i0 = pb + tid * 2; i0 < pe; i0 += SF_NT * 2) {
        float dn[2], dw[2];
        int lab[2];
        load_plane<2>(dnew, i0, dn);
        load_plane<2>(dwp, i0, dw);
        load_labels<2>(labp, i0, lab);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            if (lab[j] == SF_NC) continue;  // invalid new depth: in no cluster
            if (lab[j] != cur) {
                flush();
                cur = lab[j];
                c_size = c_nn = c_valid = 0;
                c_sum = 0;
            }
            c_size++;
            const float dwa = fabsf(dw[j]);
            if (dn[j] != 0.f && dwa != 0.f) {  // not Null
                c_nn++;
                c_sum += to_fix(1.f - kz * fabsf(dn[j] - dwa), FIX_RES, 1.0e6f);
            }
            c_valid += (dw[j] > 0.f) ? 1 : 0;
        }
    }
    flush();
    __syncthreads();
    seg_prior_finish(s, cs, tid);
}

// ---------------------------------------------------------------------------------------------
//  filterEstimateAndComputeT (reference FrontEnd.cpp:713-772) + est_cov (:689). One lane.
// ---------------------------------------------------------------------------------------------
// Called by the whole wave 0: the 6 x 6 inverse and the Jacobi sweeps use six lanes (same arithmetic per element as
// one lane would do), everything else runs on lane 0.
__device__ __noinline__ void solve_filter_and_update(const KArgs &a, LDS SolveShared &s, int level, int lane) {
    // est_cov = AtA.inverse() * res.squaredNorm()
    LDS double *Ad = s.dwork, *V = s.dwork + 72;
#ifdef SF_FILTER_PROFILE
    long long ft = wall_clock64();
#define FILTER_MARK(slot) do { if (lane == 0) { const long long n_ = wall_clock64(); s.prof[slot] += n_ - ft; ft = n_; } } while (0)
#else
#define FILTER_MARK(slot) do {} while (0)
#endif
    {
        double aa = (lane < 36) ? (double)s.AtA[lane] : 0.0, ainv;
        inverse6_lanes(aa, ainv, lane);
        if (lane < 36) s.est_cov[lane] = (float)ainv * s.res_sqnorm;
    }
    __builtin_amdgcn_wave_barrier();
    FILTER_MARK(21);

    float twist[6];
    for (int i = 0; i < 6; i++) twist[i] = s.Var[i];

    if (a.p.use_motion_filter) {
        LDS double *S = Ad;  // reuse
        bool finite = true;
        for (int i = 0; i < 6; i++)  // uniform: every lane looks at the same 21 values
            for (int j = 0; j <= i; j++)
                if (!isfinite((double)s.est_cov[i * 6 + j])) finite = false;
        if (!finite) {  // "Eigensolver couldn't find a solution. Pose is not updated"
            if (lane == 0) s.status |= SF_STATUS_EIG_SKIPPED;
            return;
        }
        {
            const int l = (lane < 36) ? lane : 0, i = l / 6, j = l - 6 * i;
            double sa = (double)s.est_cov[(i >= j) ? i * 6 + j : j * 6 + i], vv;  // the lower triangle, mirrored
#if SF_REFORDER && SF_RO_JACOBI
            if (lane < 36) S[lane] = sa;
            __builtin_amdgcn_wave_barrier();
            jacobi_eig6_wave(S, V, lane);  // the cyclic order of the oracle ([C5]), element for element
            (void)vv;
#else
            jacobi6_lanes(sa, vv, lane);
            if (lane < 36) {
                S[lane] = sa;  // the diagonal holds the eigenvalues
                V[lane] = vv;
            }
#endif
        }
        __builtin_amdgcn_wave_barrier();
        FILTER_MARK(22);
        if (lane != 0) return;
        float kai_loc_sub[6], lt[6];
        log_twist_cm(s.T, lt);
        for (int i = 0; i < 6; i++) kai_loc_sub[i] = s.twist_old[i] - lt[i];
        const float e_l = (float)exp(-(double)level);
        const float cf = a.p.previous_speed_eig_weight * e_l, df = a.p.previous_speed_const_weight * e_l;
        double kai_b_fil[6];
        for (int i = 0; i < 6; i++) {
            double kb_ = 0, kbo = 0;
            for (int r = 0; r < 6; r++) {
                kb_ += V[r * 6 + i] * (double)twist[r];
                kbo += V[r * 6 + i] * (double)kai_loc_sub[r];
            }
            const double wgt = (double)cf * S[i * 6 + i] + (double)df;
            kai_b_fil[i] = (kb_ + wgt * kbo) / (1.0 + wgt);
        }
        for (int r = 0; r < 6; r++) {
            double acc = 0;
            for (int i = 0; i < 6; i++) acc += V[r * 6 + i] * kai_b_fil[i];
            twist[r] = (float)acc;
        }
    }
    if (lane != 0) return;

    double xi[6], E[16];
    for (int i = 0; i < 6; i++) xi[i] = (double)twist[i];
    se3_exp_d(xi, E);
    float Ef[16], Tn[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) Ef[r + 4 * c] = (float)E[r * 4 + c];
    for (int i = 0; i < 6; i++) s.twist_level[i] = twist[i];
    mul4_cm(Ef, s.T, Tn);
    for (int i = 0; i < 16; i++) s.T[i] = Tn[i];
    float tw[6];
    log_twist_cm(s.T, tw);
    for (int i = 0; i < 6; i++) s.twist[i] = tw[i];
    FILTER_MARK(23);
}

// ---------------------------------------------------------------------------------------------
//  Factored form of the two Jacobian rows.  With
//     g1 = [-1, 0, x/d, xy/d, -(x^2/d + d),  y],  g2 = [0, -1, y/d, y^2/d + d, -xy/d, -x],  g3 = [0, 0, 1, y, -x, 0]
//  the reference's rows (FrontEnd.cpp:552-585) are  a_c = pc g1 + qc g2,  a_d = twd g3 + pd g1 + qd g2,
//  b_c = -bct, b_d = -bdt  with pc = twc dcu f/d, qc = twc dcv f/d, pd = twd ddu f/d, qd = twd ddv f/d,
//  bct = twc dct, bdt = twd ddt.  Residuals then need three 6-term dot products with the solution instead
//  of twelve row entries, and the weighted rows of pass 1 are built from (w pc, w qc, ...) directly.
//  Same mathematics, different rounding association than the reference's expression order (~1e-7
//  relative on a row entry).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float vabs(float x) { return fabsf(x); }
// The rows and residuals of the passes contract multiply-add pairs explicitly (the library is built -ffp-contract=off because
// the reference has no FMA; the linearisation, whose planes are bit-compared, has none). -DSF_ROWS_FMA=0 -- part of the
// `precise` build, libsf_hip_precise.so, together with IEEE weights -- evaluates the same expressions with separate,
// individually rounded multiplies and adds: what the contraction costs in parity is measured, not asserted.
#ifndef SF_ROWS_FMA
#define SF_ROWS_FMA 1
#endif
// pass 1 folds each row's pre-weight into its Cauchy weight (one reciprocal square root per row instead of two): part of the
// product build's arithmetic, off in the `precise` build, which keeps the reference's two-step association
#ifndef SF_P1_FOLD
#define SF_P1_FOLD (SF_FAST_WEIGHTS && SF_ROWS_FMA)
#endif
#if SF_ROWS_FMA
__device__ __forceinline__ float vfma(float a, float b, float c) { return fmaf(a, b, c); }
#else
__device__ __forceinline__ float vfma(float a, float b, float c) { return a * b + c; }
#endif

// T = float: one pixel per lane and step (packed pixel pairs buy nothing on gfx950, §5.1 of DESIGN.md)
template <class T>
struct PixFact {
    T x, y, xd, yd, xyd, xxd, yyd;  // geometry: x, y, x/d, y/d, xy/d, x^2/d + d, y^2/d + d
    T pc, qc, pd, qd, twd, bct, bdt;
    T ac, ad;                       // RAW form only: the arguments 1 + e_c, 0.01 + e_d of the two pre-weights
};

// RAW = true: the same record WITHOUT the pre-weights: pc = dcu f/d, ..., bct = dct, bdt = ddt, twd = 1, and the arguments of the
// two reciprocal square roots in o.ac / o.ad -- for pass 1, which folds each pre-weight into the Cauchy weight of its row (below)
template <class T, bool RAW = false>
__device__ __forceinline__ void fact_from_record(const LevelGeom &g, T fu, T fv, T dn, T dw, T dcu_, T dcv_, T dct_, T ddu_,
                                                 T ddv_, PixFact<T> &o) {
    const T xn = (g.inv_f_pyr * (fu - g.disp_u_i)) * dn;
    const T yn = (g.inv_f_pyr * (fv - g.disp_v_i)) * dn;
    T xw, yw;
    if (g.first) {
        xw = (g.inv_f_pyr * (fu - g.disp_u_i)) * dw;
        yw = (g.inv_f_pyr * (fv - g.disp_v_i)) * dw;
    } else {
        xw = (fu - g.disp_u_i) * dw * g.inv_f_w;
        yw = (fv - g.disp_v_i) * dw * g.inv_f_w;
    }
    const T d = 0.5f * (dn + dw);
    o.x = 0.5f * (xn + xw);
    o.y = 0.5f * (yn + yw);
    const T ddt_ = dn - dw;
    const T error_l_c = 10.f * (vabs(dct_) + vabs(dcu_) + vabs(dcv_));
    const T error_l_d = 200.f * (vabs(ddt_) + vabs(ddu_) + vabs(ddv_));
    const T inv_d = vrcpw(d);
    const T fd = g.f_inv * inv_d;
    if constexpr (RAW) {
        o.ac = 1.f + error_l_c;
        o.ad = 0.01f + error_l_d;
        o.twd = 1.f;
        o.pc = dcu_ * fd;
        o.qc = dcv_ * fd;
        o.pd = ddu_ * fd;
        o.qd = ddv_ * fd;
        o.bct = dct_;
        o.bdt = ddt_;
    } else {
        const T twc = (g.inv_max_c * vrsq(1.f + error_l_c)) * g.kph;
        o.twd = g.inv_max_d * vrsq(0.01f + error_l_d);
        o.pc = twc * (dcu_ * fd);
        o.qc = twc * (dcv_ * fd);
        o.pd = o.twd * (ddu_ * fd);
        o.qd = o.twd * (ddv_ * fd);
        o.bct = twc * dct_;
        o.bdt = o.twd * ddt_;
    }
    o.xd = o.x * inv_d;
    o.yd = o.y * inv_d;
    o.xyd = o.xd * o.y;
    o.xxd = vfma(o.xd, o.x, d);
    o.yyd = vfma(o.yd, o.y, d);
}

// residuals res = A Var - B of both rows through s1 = g1.Var, s2 = g2.Var, s3 = g3.Var
template <class T>
__device__ __forceinline__ void fact_residuals(const PixFact<T> &p, const float (&V)[6], T &res_c, T &res_d) {
    const T s1 = vfma(p.y, V[5], vfma(-p.xxd, V[4], vfma(p.xyd, V[3], vfma(p.xd, V[2], -V[0]))));
    const T s2 = vfma(-p.x, V[5], vfma(-p.xyd, V[4], vfma(p.yyd, V[3], vfma(p.yd, V[2], -V[1]))));
    const T s3 = vfma(-p.x, V[4], vfma(p.y, V[3], V[2]));
    res_c = vfma(p.pc, s1, vfma(p.qc, s2, p.bct));
    res_d = vfma(p.pd, s1, vfma(p.qd, s2, vfma(p.twd, s3, p.bdt)));
}

// A pixel that is not in validPixels gets a harmless stand-in record (finite rows) and weight 0,
// so the streaming loops are branch-free: no exec-mask juggling around the 27 accumulators.
template <int VEC>
__device__ __forceinline__ bool sanitize(RecVec<VEC> &r, int j) {
    const bool ok = r.v[R_DW][j] > 0.f;  // the linearisation stores -dw (or -0) outside validPixels
    r.dn[j] = ok ? r.dn[j] : 1.f;
    r.v[R_DW][j] = ok ? r.v[R_DW][j] : 1.f;
    // the four gradients and dct of such a pixel are stored as 0 by the linearisation (dct keeps its value in the debug-plane
    // mode only): nothing to do for them here
    r.lab[j] = ok ? (int)((j ? r.labraw >> 8 : r.labraw) & 255u) : 0;
    return ok;
}

// ---------------------------------------------------------------------------------------------
//  solveOdometryAndSegmJoint (reference FrontEnd.cpp:513-692), split into separately compiled
//  pieces so that each streaming pass gets its own register allocation.
// ---------------------------------------------------------------------------------------------
struct IrlsCtx {
    RecPtrs rp;
    LevelGeom g;
    int n;       // end of the pixel range of this workgroup (the level size in the product)
    int begin;   // start of the range (0 in the product; tools/pass_microbench.py --slices splits a level)
    int N;       // valid pixels
};

__device__ __forceinline__ IrlsCtx make_irls_ctx(const KArgs &a, int b, int L, const LDS SolveShared &s) {
    IrlsCtx c;
    const size_t rb = (size_t)uniform_i(s.rec_slot) * a.n0;
#pragma unroll
    for (int q = 0; q < R_COUNT; q++) c.rp.p[q] = uniform_ptr((gcfloat *)(a.rec[q] + rb));
    c.rp.dnew = uniform_ptr((gcfloat *)pyr_level(a, b, 0, 0, L));
    c.rp.lab = uniform_ptr((gcu8 *)(a.rec_lab + rb));
    c.rp.with_labels = uniform_i(a.p.segmentation_enabled);
    c.n = uniform_i(s.px_end);
    c.begin = uniform_i(s.px_begin);
    c.N = uniform_i(s.n_valid);
    const int rows_i = a.lrows[L], cols_i = a.lcols[L];
    const float f = float(cols_i) / (2.f * a.tan_half_fovh);
    c.g.rows_i = rows_i;
    c.g.inv_rows = 1.f / float(rows_i);
    c.g.disp_u_i = 0.5f * float(cols_i - 1);
    c.g.disp_v_i = 0.5f * float(rows_i - 1);
    c.g.inv_f_pyr = 2.f * a.tan_half_fovh / float(cols_i);
    c.g.inv_f_w = 1.f / f;
    c.g.f_inv = f;
    c.g.kph = a.p.k_photometric_res;
    c.g.inv_max_c = uniform_f(s.inv_max_c);
    c.g.inv_max_d = uniform_f(s.inv_max_d);
    c.g.first = uniform_i(s.first);
    return c;
}

// pass 1: Cauchy x b weights, 21+6 normal-equation sums (reference :615-641) -> s.red[wave][0..26]
// VAR: 0 = product code; 1 = loads only; 2 = rows + weights, no accumulation (ablation builds for
// tools/pass_microbench.py; the product always instantiates VAR 0)
//
// Scalar fp32 per pixel: on gfx950 a v_pk_*_f32 and a v_fma_f64 both cost two v_fma_f32 issue slots
// (tools/micro/valu_rate.hip), so packing buys nothing and costs registers. The 27 sums are kept
// per lane in fp32 (each lane sees <= 2 x 300 terms at QVGA level 0; the reference accumulates the
// whole sum in fp32, FrontEnd.cpp:640-641) and the 256 lanes are combined in fp64. The record of
// the next pixel pair is in flight while the current one is evaluated.
__device__ __forceinline__ void accum_row(float (&acc)[27], const float (&aw)[7]) {
    acc[0] = fmaf(aw[0], aw[0], acc[0]);    acc[1] = fmaf(aw[0], aw[1], acc[1]);
    acc[2] = fmaf(aw[0], aw[2], acc[2]);    acc[3] = fmaf(aw[0], aw[3], acc[3]);
    acc[4] = fmaf(aw[0], aw[4], acc[4]);    acc[5] = fmaf(aw[0], aw[5], acc[5]);
    acc[6] = fmaf(aw[1], aw[1], acc[6]);    acc[7] = fmaf(aw[1], aw[2], acc[7]);
    acc[8] = fmaf(aw[1], aw[3], acc[8]);    acc[9] = fmaf(aw[1], aw[4], acc[9]);
    acc[10] = fmaf(aw[1], aw[5], acc[10]);  acc[11] = fmaf(aw[2], aw[2], acc[11]);
    acc[12] = fmaf(aw[2], aw[3], acc[12]);  acc[13] = fmaf(aw[2], aw[4], acc[13]);
    acc[14] = fmaf(aw[2], aw[5], acc[14]);  acc[15] = fmaf(aw[3], aw[3], acc[15]);
    acc[16] = fmaf(aw[3], aw[4], acc[16]);  acc[17] = fmaf(aw[3], aw[5], acc[17]);
    acc[18] = fmaf(aw[4], aw[4], acc[18]);  acc[19] = fmaf(aw[4], aw[5], acc[19]);
    acc[20] = fmaf(aw[5], aw[5], acc[20]);
    acc[21] = fmaf(aw[0], aw[6], acc[21]);  acc[22] = fmaf(aw[1], aw[6], acc[22]);
    acc[23] = fmaf(aw[2], aw[6], acc[23]);  acc[24] = fmaf(aw[3], aw[6], acc[24]);
    acc[25] = fmaf(aw[4], aw[6], acc[25]);  acc[26] = fmaf(aw[5], aw[6], acc[26]);
}

// sum of v over the lane's group of P1_GROUP lanes (every lane of the group ends up with the same bits: the two / four
// exchange steps are symmetric). All 64 lanes must be active.
__device__ __forceinline__ float p1_group_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
    if constexpr (P1_GROUP == 16) {
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
    }
    return v;
}
__device__ __forceinline__ void p1_flush(float (&acc)[27], LDS SolveShared &s, int set, bool leader) {
#pragma unroll
    for (int q = 0; q < 27; q++) acc[q] = p1_group_sum(acc[q]);
    if (leader) {  // the set belongs to this lane group alone: plain read-modify-writes, one exec-mask change for all 27
#pragma unroll
        for (int q = 0; q < 27; q++) s.p1[q][set] += (double)acc[q];
    }
#pragma unroll
    for (int q = 0; q < 27; q++) acc[q] = 0.f;
}

template <int VAR>
__device__ __noinline__ void irls_pass1(const KArgs &a, int b, int L, LDS SolveShared &s, int tid) {
    const IrlsCtx c = make_irls_ctx(a, b, L, s);
    const int lane = tid & 63, wave = tid >> 6;
    const float inv_c_Cauchy = 1.f / (a.p.kc_Cauchy * uniform_f(s.aver_res));
#if SF_P1_FOLD
    const float fold_kc = c.g.inv_max_c * c.g.kph, fold_kd = c.g.inv_max_d;                    // pre-weight = k rsq(a)
    const float fold_gc = fold_kc * inv_c_Cauchy, fold_gd = fold_kd * inv_c_Cauchy;
#endif
    float acc[27];
#pragma unroll
    for (int q = 0; q < 27; q++) acc[q] = 0.f;
    const int set = tid / P1_GROUP;
    const bool leader = (tid % P1_GROUP) == 0;
    if (lane < P1_SETS_PER_WAVE) {  // this wave's sets (nobody else touches them: no barrier, LDS operations of a wave are ordered)
#pragma unroll
        for (int q = 0; q < 27; q++) s.p1[q][wave * P1_SETS_PER_WAVE + lane] = 0.0;
    }
    float Vr[6];
#pragma unroll
    for (int q = 0; q < 6; q++) Vr[q] = uniform_f(s.Var[q]);
    const int last = (c.n - 2) & ~1;  // the prefetch past the end re-reads the last pair instead of branching
    RecVec<2> rv, nx;
    // the trip count is the WAVE's (its first lane's): every lane stays active to the end, so that the group sums of a
    // flush see all their lanes; a lane past the end re-reads the last pair with weight 0
    load_rec<2>(c.rp, min(c.begin + tid * 2, last), rv);
    int since = 0;
    for (int i0 = c.begin + tid * 2, iw = uniform_i(c.begin + (tid - lane) * 2); iw < c.n; i0 += SF_NT * 2, iw += SF_NT * 2) {
        load_rec<2>(c.rp, min(i0 + SF_NT * 2, last), nx);
        const bool in = i0 < c.n;
        const bool ok0 = sanitize<2>(rv, 0) && in, ok1 = sanitize<2>(rv, 1) && in;
        if constexpr (VAR == 1) {
            float t = rv.dn[0] + rv.dn[1];
#pragma unroll
            for (int q = 0; q < R_COUNT; q++) t += rv.v[q][0] + rv.v[q][1];
            acc[0] += t;
            rv = nx;
            continue;
        }
        float bseg0 = s.b_segm[rv.lab[0]], bseg1 = s.b_segm[rv.lab[1]];  // invalid pixels carry label 0 after sanitize()
        float fu0, fv0;
        split_index(c.g, min(i0, last), fu0, fv0);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const bool ok = j ? ok1 : ok0;
            float fu = fu0, fv = fv0;
            if (j) {  // the pair may straddle a column at the odd-sized coarse levels
                const bool wrap = (fv0 + 1.f) >= (float)c.g.rows_i;
                fu = wrap ? fu0 + 1.f : fu0;
                fv = wrap ? 0.f : fv0 + 1.f;
            }
            PixFact<float> p;
#if SF_P1_FOLD
            // The weight a row finally carries is (pre-weight) x (Cauchy weight) = k rsq(a) b rsq(1 + (k rsq(a) R / c)^2) with R the
            // residual of the UNWEIGHTED row, a = 1 + e_c (0.01 + e_d) and k = kph / max (1 / max): that is b k rsq(a + (k R / c)^2) --
            // one reciprocal square root per row instead of two, and the rows are scaled once instead of twice. Same mathematics;
            // the rounding of a row entry moves by ~1e-7 relative like the factored rows themselves (pass 2 and the debug
            // expansion of the rows keep the two-step form).
            fact_from_record<float, true>(c.g, fu, fv, rv.dn[j], rv.v[R_DW][j], rv.v[R_DCU][j], rv.v[R_DCV][j], rv.v[R_DCT][j],
                                          rv.v[R_DDU][j], rv.v[R_DDV][j], p);
            if (j == 0) asm volatile("" : "+v"(bseg0), "+v"(bseg1));  // LDS reads stay unconditional, landed by now
            const float b_weight = ok ? std_max(0.f, std_min(1.f, j ? bseg1 : bseg0)) : 0.f;
            float raw_c, raw_d;
            fact_residuals<float>(p, Vr, raw_c, raw_d);
            const float uc = raw_c * fold_gc, ud = raw_d * fold_gd;
            const float w_c = (b_weight * fold_kc) * vrsq(fmaf(uc, uc, p.ac));
            const float w_d = (b_weight * fold_kd) * vrsq(fmaf(ud, ud, p.ad));
#else
            fact_from_record<float>(c.g, fu, fv, rv.dn[j], rv.v[R_DW][j], rv.v[R_DCU][j], rv.v[R_DCV][j], rv.v[R_DCT][j],
                                    rv.v[R_DDU][j], rv.v[R_DDV][j], p);
            if (j == 0) asm volatile("" : "+v"(bseg0), "+v"(bseg1));  // LDS reads stay unconditional, landed by now
            const float b_weight = ok ? std_max(0.f, std_min(1.f, j ? bseg1 : bseg0)) : 0.f;
            float res_c, res_d;
            fact_residuals<float>(p, Vr, res_c, res_d);
            const float tc = res_c * inv_c_Cauchy, td = res_d * inv_c_Cauchy;
#if SF_FAST_WEIGHTS
            const float w_c = b_weight * vrsq(vfma(tc, tc, 1.f));
            const float w_d = b_weight * vrsq(vfma(td, td, 1.f));
#else
            const float w_c = b_weight * vrsq(1.f + tc * tc);
            const float w_d = b_weight * vrsq(1.f + td * td);
#endif
#endif
            float aw[7];
            {
                const float P = w_c * p.pc, Q = w_c * p.qc;
                aw[0] = -P;
                aw[1] = -Q;
                aw[2] = vfma(P, p.xd, Q * p.yd);
                aw[3] = vfma(P, p.xyd, Q * p.yyd);
                aw[4] = -vfma(P, p.xxd, Q * p.xyd);
                aw[5] = vfma(P, p.y, -(Q * p.x));
                aw[6] = -(w_c * p.bct);
            }
            if constexpr (VAR == 2)
                acc[0] += ((aw[0] + aw[1]) + (aw[2] + aw[3])) + ((aw[4] + aw[5]) + aw[6]);
            else
                accum_row(acc, aw);
            {
                const float W = w_d * p.twd, Pd = w_d * p.pd, Qd = w_d * p.qd;
                aw[0] = -Pd;
                aw[1] = -Qd;
                aw[2] = vfma(Pd, p.xd, vfma(Qd, p.yd, W));
                aw[3] = vfma(Pd, p.xyd, vfma(Qd, p.yyd, W * p.y));
                aw[4] = -vfma(Pd, p.xxd, vfma(Qd, p.xyd, W * p.x));
                aw[5] = vfma(Pd, p.y, -(Qd * p.x));
                aw[6] = -(w_d * p.bdt);
            }
            if constexpr (VAR == 2)
                acc[0] += ((aw[0] + aw[1]) + (aw[2] + aw[3])) + ((aw[4] + aw[5]) + aw[6]);
            else
                accum_row(acc, aw);
        }
        rv = nx;
        if constexpr (VAR == 0) {
            if (++since == SF_P1_FLUSH) {  // uniform: every lane of the wave has made the same number of trips
                since = 0;
                p1_flush(acc, s, set, leader);
            }
        }
    }
    p1_flush(acc, s, set, leader);
    // this wave's sets, in order -> s.red[wave][0..26]
    __builtin_amdgcn_wave_barrier();
    if (lane < 27) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < P1_SETS_PER_WAVE; g++) t += s.p1[lane][wave * P1_SETS_PER_WAVE + g];
        s.red[wave][lane] = t;
    }
}

// all threads: the 27 sums of pass 1 over the waves of this workgroup, then over the workgroups of the cluster (fixed
// orders: every workgroup ends up with the same bits) -> s.red[0][0..26]
__device__ __forceinline__ void irls_reduce_normal(LDS SolveShared &s, LDS ClusterShared &cs, int tid) {
    if (tid < 27) {
        double t = 0.0;
        for (int w = 0; w < SF_NW; w++) t += s.red[w][tid];
        put_f64(&cs.in[2 * tid], t);
    }
    cluster_gather(cs, 54, tid);
    if (tid < 27) {
        const int G = cl_G(cs);
        double t = 0.0;
        for (int p = 0; p < G; p++) t += get_f64(&cs.all[p * 54 + 2 * tid]);
        s.red[0][tid] = t;
    }
    __syncthreads();
}

// wave 0: AtA / AtB from the reduced sums, Var = AtA.ldlt().solve(AtB) (reference :640-642)
__device__ __noinline__ void irls_solve_normal(LDS SolveShared &s, int lane) {
    if (lane < 36) {
        const int i = lane / 6, j = lane - 6 * i;
        const int lo = min(i, j), hi = max(i, j);
        const int q = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);  // upper-triangular packing of pass 1
        const float v = (float)s.red[0][q];
        s.AtA[lane] = v;
        s.M6[i * 7 + j] = v;
    }
    if (lane < 6) {
        const float v = (float)s.red[0][21 + lane];
        s.AtB[lane] = v;
        s.y6[lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    const bool az = ldlt_factor_wave<6>(s.M6, s.tmp6, s.tr6, lane);
    ldlt_solve_wave<6>(s.M6, s.tr6, az, s.y6, lane);
    if (lane < 6) s.Var[lane] = s.y6[lane];
    if (lane < SF_NC) s.lab_sum[lane] = 0;
}

// all threads, after pass 2: the per-label sums (exact integers) and ||res||^2 over the workgroups of the cluster
__device__ __forceinline__ void irls_reduce_residuals(LDS SolveShared &s, LDS ClusterShared &cs, int tid) {
    if (tid < SF_NC) put_i64(&cs.in[2 * tid], s.lab_sum[tid]);
    if (tid == SF_NC) {
        double q = 0.0;
        for (int w = 0; w < SF_NW; w++) q += s.red[w][27];
        put_f64(&cs.in[2 * SF_NC], q);
    }
    cluster_gather(cs, 2 * SF_NC + 2, tid);
    const int G = cl_G(cs);
    if (tid < SF_NC) {
        long long t = 0;
        for (int p = 0; p < G; p++) t += get_i64(&cs.all[p * (2 * SF_NC + 2) + 2 * tid]);
        s.lab_sum[tid] = t;
    }
    if (tid == SF_NC) {
        double q = 0.0;
        for (int p = 0; p < G; p++) q += get_f64(&cs.all[p * (2 * SF_NC + 2) + 2 * SF_NC]);
        s.sq_total = q;
    }
    __syncthreads();
}

// non-negative float (< 2^20) -> Q32.32 fixed point without the emulated float->int64 conversion
__device__ __forceinline__ unsigned long long to_fix32_pos(float x) {
    float y = x;
    if (!(y < 1.0e6f)) y = 1.0e6f;  // also catches NaN
    const unsigned hi = (unsigned)y;              // floor
    const float frac = y - (float)hi;              // exact
    const unsigned lo = (unsigned)(frac * 4294967296.f);
    return ((unsigned long long)hi << 32) | lo;
}

// pass 2: residuals with the new solution, per-label sums, ||res||^2 (reference :644-667).
// Per-label sums: each lane keeps a running fixed-point sum for the label of its last pixel and
// flushes it to the workgroup bins (LDS integer atomics: order-independent) only when the label
// changes -- labels are spatially coherent, so flushes are rare.
template <int VAR>
__device__ __noinline__ void irls_pass2(const KArgs &a, int b, int L, LDS SolveShared &s, int tid) {
    const IrlsCtx c = make_irls_ctx(a, b, L, s);
    const int lane = tid & 63, wave = tid >> 6;
    float Vr[6];
#pragma unroll
    for (int q = 0; q < 6; q++) Vr[q] = uniform_f(s.Var[q]);
    double sq = 0.0;
    int cur_lab = 0;
    unsigned long long cur_sum = 0;
    const int last = (c.n - 2) & ~1;
    RecVec<2> rv, nx;
    if (c.begin + tid * 2 < c.n) load_rec<2>(c.rp, c.begin + tid * 2, rv);
    for (int i0 = c.begin + tid * 2; i0 < c.n; i0 += SF_NT * 2) {
        load_rec<2>(c.rp, min(i0 + SF_NT * 2, last), nx);  // next pair in flight during this one
        const bool ok0 = sanitize<2>(rv, 0), ok1 = sanitize<2>(rv, 1);
        if constexpr (VAR == 1) {
            float t = rv.dn[0] + rv.dn[1];
#pragma unroll
            for (int q = 0; q < R_COUNT; q++) t += rv.v[q][0] + rv.v[q][1];
            sq += (double)t;
            rv = nx;
            continue;
        }
        float fu0, fv0;
        split_index(c.g, i0, fu0, fv0);
#pragma unroll
        for (int px = 0; px < 2; px++) {
            const bool ok = px ? ok1 : ok0;
            float fu = fu0, fv = fv0;
            if (px) {
                const bool wrap = (fv0 + 1.f) >= (float)c.g.rows_i;
                fu = wrap ? fu0 + 1.f : fu0;
                fv = wrap ? 0.f : fv0 + 1.f;
            }
            PixFact<float> p;
            fact_from_record<float>(c.g, fu, fv, rv.dn[px], rv.v[R_DW][px], rv.v[R_DCU][px], rv.v[R_DCV][px], rv.v[R_DCT][px],
                                    rv.v[R_DDU][px], rv.v[R_DDV][px], p);
            float rc, rd;
            fact_residuals<float>(p, Vr, rc, rd);
            const float rcs = ok ? rc : 0.f;
            const float rds = ok ? rd : 0.f;
            sq = fma((double)rcs, (double)rcs, sq);
            sq = fma((double)rds, (double)rds, sq);
            const unsigned long long fx = to_fix32_pos(fabsf(rcs) + fabsf(rds));
            if constexpr (VAR == 2) {
                sq += (double)(unsigned)(fx >> 32);
                continue;
            }
            const int lab = ok ? rv.lab[px] : cur_lab;
            if (lab != cur_lab) {
                if (cur_sum) lds_add(&s.lab_sum[cur_lab], (long long)cur_sum);
                cur_lab = lab;
                cur_sum = 0;
            }
            cur_sum += fx;
        }
        rv = nx;
    }
    if (cur_sum) lds_add(&s.lab_sum[cur_lab], (long long)cur_sum);
    sq = wave_sum_f64(sq);
    if (lane == 0) s.red[wave][27] = sq;
}

#include "sf_reforder_solver.h"  // (empty unless SF_REFORDER)

// wave 0: build and factorise A_seg^T A_seg once per outer iteration
// (reference SegmentationBackground.cpp:105-130,143-165)
__device__ __noinline__ void irls_seg_factor(const KArgs &a, LDS SolveShared &s, int lane) {
    const float lambda_prior = a.p.lambda_prior;
    const float weight_reg = 2.f * a.p.lambda_reg;
    const float w2 = weight_reg * weight_reg, nw2 = weight_reg * (-weight_reg);
    if (lane < SF_NC) {
        const int l = lane;
        const float lt = s.lambda_t_w[l];
        const float dg = (lt > 0.1f) ? 2.f * lt * lambda_prior : 2.f * lt;
        s.seg_diag[l] = dg;
        const unsigned cm = s.conn[l];
        double dd = (double)(dg * dg);
        for (int lc = 0; lc < SF_NC; lc++) {
            const bool con = (lc != l) && ((cm >> lc) & 1u);
            if (con) dd += (double)w2;
            if (lc != l) s.M24[l * (SF_NC + 1) + lc] = con ? nw2 : 0.f;
        }
        s.M24[l * (SF_NC + 1) + l] = (float)dd;
    }
    __builtin_amdgcn_wave_barrier();
    const bool az = ldlt_factor_wave<SF_NC>(s.M24, s.tmp24, s.tr24, lane);
    if (lane == 0) s.seg_allzero = az ? 1 : 0;
}

// wave 0, after pass 2: averages, solveSegmIteration, convergence test (reference :666-683)
__device__ __noinline__ void irls_iteration_tail(const KArgs &a, LDS SolveShared &s, int N, int k, int lane) {
    const bool seg = a.p.segmentation_enabled != 0;
#if !(SF_REFORDER && SF_RO_LABSUM)  // (the reference-order build's pass 2 leaves the sequential float sums there itself)
    if (lane < SF_NC) s.aver_res_label[lane] = (float)((double)s.lab_sum[lane] * (1.0 / 4294967296.0));
#endif
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        double t = 0.0;
        for (int l = 0; l < SF_NC; l++) t += (double)s.aver_res_label[l];
        s.aver_res_old = s.aver_res;
        s.aver_res = (float)t / float(2 * N);
        s.res_sqnorm = (float)s.sq_total;
    }
    __builtin_amdgcn_wave_barrier();
    if (seg) {
        // solveSegmIteration (reference SegmentationBackground.cpp:133-174)
        if (lane < SF_NC) {
            const int l = lane;
            const int npl = s.valid_cnt[l] + 1;  // num_pix_label starts at 1 (reference :651)
            const float arl = s.aver_res_label[l] / float(2 * npl);
            const float aro = s.aver_res_old;  // the PREVIOUS iteration's overall average (reference :652,672)
            const float kc = a.p.kc_Cauchy;
            const float repr_res = std_max(0.001f, aro);
            const float fixed_term = (float)log((double)(1.f + sqf(s.kb * repr_res / (kc * aro))));
            const float mult_res = 1.f / (kc * aro);
            const float lt = s.lambda_t_w[l];
            float Bseg;
            if (lt > 0.1f) {
                const float dataterm = fixed_term - (float)log((double)(1.f + sqf(arl * mult_res)));
                Bseg = dataterm + 2.f * a.p.lambda_prior * lt * s.b_prior[l];
            } else {
                Bseg = 2.f * lt * s.b_prior[l];
            }
            s.y24[l] = s.seg_diag[l] * Bseg;
        }
        __builtin_amdgcn_wave_barrier();
        ldlt_solve_wave<SF_NC>(s.M24, s.tr24, s.seg_allzero != 0, s.y24, lane);
        if (lane < SF_NC) s.b_segm[lane] = std_max(-1.f, std_min(2.f, s.y24[lane]));
    }
    if (lane == 0) {
        float delta = 0.f;
        for (int c = 0; c < 6; c++) delta = std_max(delta, fabsf(s.prev_sol[c] - s.Var[c]));
        for (int c = 0; c < 6; c++) s.prev_sol[c] = s.Var[c];
        s.last_delta = delta;
        s.ctrl = ((delta < a.p.irls_delta_threshold) || (k == a.p.max_iter_irls)) ? 1 : 0;
        s.n_irls++;
        s.pixel_iters += N;
    }
}

__device__ __noinline__ void solve_irls(const KArgs &a, int b, int L, int level, int kouter, LDS SolveShared &s, LDS ClusterShared &cs, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const bool seg = a.p.segmentation_enabled != 0;
    const int N = __builtin_amdgcn_readfirstlane(s.n_valid);
    const int n_outer_now = __builtin_amdgcn_readfirstlane(s.n_outer);
    // the trace is written by ONE workgroup of a cluster (all of them hold the same values)
    sf_outer_trace *tr = (n_outer_now < SF_MAX_OUTER && cl_writer(cs)) ? &a.stats[b].outer[n_outer_now] : nullptr;

    // b initialisation (reference :603-607)
    if (tid < SF_NC) {
        if (!seg)
            s.b_segm[tid] = 1.f;
        else if (level == 0)
            s.b_segm[tid] = s.b_prior[tid];
    }
    if (tid < 6) {
        s.Var[tid] = 0.f;
        s.prev_sol[tid] = 0.f;
    }
    if (tid == 0) {
        int pb, pe;
        cluster_range(cs, a.ln[L], 2, pb, pe);  // the passes walk pixel pairs
        s.px_begin = pb;
        s.px_end = pe;
        s.rec_slot = cs.slot;
    }
    __syncthreads();

    if (N == 0) {  // defined behaviour for an empty level: nothing moves
        if (tid < 6) s.twist_level[tid] = 0.f;
        if (tr) {
            if (tid == 0) {
                tr->level = level; tr->k = kouter; tr->n_valid = 0; tr->irls_iters = 0; tr->aver_res = 0.f;
                tr->delta_sol_max = 0.f;
            }
            if (tid < 6) tr->var[tid] = tr->twist_level[tid] = tr->AtB[tid] = 0.f;
            if (tid < 16) tr->T[tid] = s.T[tid];
            if (tid < 36) tr->AtA[tid] = 0.f;
            if (tid < SF_NC) {
                tr->b_segm[tid] = s.b_segm[tid];
                tr->b_prior[tid] = s.b_prior[tid];
                tr->lambda_t_w[tid] = s.lambda_t_w[tid];
            }
        }
        __syncthreads();
        return;
    }

    // initial aver_res = mean |res| with res = -B (reference :588-590). B = (pre-weight / max) * derivative:
    // the sums of raw pre-weight x |dct|, |ddt| come from the linearisation, so no extra pass over the records
#if SF_REFORDER && SF_RO_INIT_RES
    ro_initial_residual(a, b, L, s, tid);  // from the rows' B, as the reference does
#else
    if (tid == 0) {
        const double t = (double)(s.inv_max_c * a.p.k_photometric_res) * s.init_abs_c + (double)s.inv_max_d * s.init_abs_d;
        s.aver_res = (float)t / float(2 * N);
    }
#endif
    if (seg && wave == 0) irls_seg_factor(a, s, lane);
    __syncthreads();
    PROF_MARK(s, tid, PF_IRLS_INIT);

    int iters_done = 0;
    for (int k = 1; k <= a.p.max_iter_irls; k++) {
        iters_done = k;
#if SF_REFORDER
        ro_pass1(a, b, L, s, tid);
#else
        irls_pass1<0>(a, b, L, s, tid);
#endif
        __syncthreads();
        irls_reduce_normal(s, cs, tid);
        PROF_MARK(s, tid, PF_PASS1);
        if (wave == 0) irls_solve_normal(s, lane);
        __syncthreads();
        PROF_MARK(s, tid, PF_SOLVE6);
#if SF_REFORDER
        ro_pass2(a, b, L, s, tid);
#else
        irls_pass2<0>(a, b, L, s, tid);
#endif
        __syncthreads();
        irls_reduce_residuals(s, cs, tid);
        PROF_MARK(s, tid, PF_PASS2);
        if (wave == 0) irls_iteration_tail(a, s, N, k, lane);
        __syncthreads();
        PROF_MARK(s, tid, PF_TAIL);
        if (__builtin_amdgcn_readfirstlane(s.ctrl)) break;
    }

    if (tr && wave == SF_NW - 1) {  // the trace: a wave that is not busy with the filter (one wave: after it, in order)
        if (lane == 0) {
            tr->level = level; tr->k = kouter; tr->n_valid = N; tr->irls_iters = iters_done;
            tr->aver_res = s.aver_res;
            tr->delta_sol_max = s.last_delta;
        }
        if (lane < 6) {
            tr->var[lane] = s.Var[lane];
            tr->AtB[lane] = s.AtB[lane];
        }
        if (lane < 36) tr->AtA[lane] = s.AtA[lane];
        if (lane < SF_NC) {
            tr->b_prior[lane] = s.b_prior[lane];
            tr->lambda_t_w[lane] = s.lambda_t_w[lane];
        }
    }
    if (wave == 0) solve_filter_and_update(a, s, level, lane);
    __syncthreads();
    if (tr) {
        if (tid < 6) tr->twist_level[tid] = s.twist_level[tid];
        if (tid < SF_NC) tr->b_segm[tid] = s.b_segm[tid];
        if (tid < 16) tr->T[tid] = s.T[tid];
    }
    __syncthreads();
    PROF_MARK(s, tid, PF_FILTER);
}

// ---------------------------------------------------------------------------------------------
//  the coarse-to-fine loop (reference FrontEnd.cpp:1091-1144)
// ---------------------------------------------------------------------------------------------
// Levels of at most this many pixels are not worth a rendezvous per reduction: in the cluster build every workgroup runs
// them on its own (redundantly, on a private record slot) and all arrive at bit-identical state.
#define SF_CLUSTER_SOLO_PIXELS 8192

__device__ __noinline__ void stage_solve(const KArgs &a, int b, LDS SolveShared &s, LDS ClusterShared &cs, int tid) {
    StreamState &st = a.state[b];
    if (tid < 16) s.T[tid] = (tid % 5 == 0) ? 1.f : 0.f;  // T_odometry.setIdentity()  (:1091)
    if (tid < 6) {
        s.twist_old[tid] = st.twist_old[tid];
        s.twist[tid] = st.twist[tid];
        s.twist_level[tid] = st.twist_level[tid];
    }
    if (tid < SF_NC) {
        s.b_segm[tid] = st.b_segm[tid];
        s.conn[tid] = st.conn[tid];
        s.b_prior[tid] = st.b_prior[tid];
        s.lambda_t_w[tid] = st.lambda_t_w[tid];
    }
    if (tid < SF_PROF_SLOTS) s.prof[tid] = 0;
    if (tid == 0) {
        s.t_last = wall_clock64();
        s.kb = st.kb;
        s.status = 0;
        s.n_irls = 0;
        s.n_outer = 0;
        s.pixel_iters = 0;
    }
    __syncthreads();
    const bool clustered = uniform_i(cs.full_G) > 1;

    int last_L = 0;
    for (int i = 0; i < a.levels; i++) {
        const int L = a.levels - i - 1;  // image_level
        if (clustered) cluster_set_solo(cs, tid, a.ln[L] <= SF_CLUSTER_SOLO_PIXELS);
        for (int k = 0; k < a.p.max_iter_per_level; k++) {
            last_L = L;
            const bool first = (i == 0) && (k == 0);
            if (!first) solve_warp(a, b, L, s, cs, tid);
            PROF_MARK(s, tid, PF_WARP);
#if SF_LIN_STRIPS
            {
                const int which = (uniform_i(a.p.debug_planes) ? 4 : 0) | (first ? 2 : 0) | (uniform_i(a.p.segmentation_enabled) ? 1 : 0);
                switch (which) {  // (uniform)
                case 0: solve_linearise_strips<false, false, false>(a, b, L, s, cs, tid); break;
                case 1: solve_linearise_strips<false, false, true>(a, b, L, s, cs, tid); break;
                case 2: solve_linearise_strips<false, true, false>(a, b, L, s, cs, tid); break;
                case 3: solve_linearise_strips<false, true, true>(a, b, L, s, cs, tid); break;
                case 4: solve_linearise_strips<true, false, false>(a, b, L, s, cs, tid); break;
                case 5: solve_linearise_strips<true, false, true>(a, b, L, s, cs, tid); break;
                case 6: solve_linearise_strips<true, true, false>(a, b, L, s, cs, tid); break;
                default: solve_linearise_strips<true, true, true>(a, b, L, s, cs, tid); break;
                }
            }
#else
            solve_linearise(a, b, L, first, s, cs, tid);
#endif
#if SF_REFORDER
            if (a.p.segmentation_enabled) ro_seg_prior(a, b, L, s, cs, tid);
#else
#if SF_LIN_FUSED_PRIOR
            if (uniform_i(a.p.segmentation_enabled)) seg_prior_finish(s, cs, tid);  // (the sums: solve_linearise_strips)
#else
            if (a.p.segmentation_enabled) solve_seg_prior(a, b, L, s, cs, tid);
#endif
#endif
            PROF_MARK(s, tid, PF_LINEARISE);
            solve_irls(a, b, L, i, k, s, cs, tid);
            if (tid == 0) {
                s.n_outer++;
                double s2 = 0.0;
#if SF_REFORDER  // the oracle's reading of twist_level_odometry.norm(): fp64 sum of the FLOAT squares
                for (int c = 0; c < 6; c++) s2 += (double)(s.twist_level[c] * s.twist_level[c]);
                const float nrm = (float)sqrt((double)(float)s2);
#else
                for (int c = 0; c < 6; c++) s2 += (double)s.twist_level[c] * (double)s.twist_level[c];
                const float nrm = sqrtf((float)s2);
#endif
                s.ctrl = (nrm < 0.04f) ? 1 : 0;  // reference :1130
            }
            __syncthreads();
            const int brk = __builtin_amdgcn_readfirstlane(s.ctrl);
            __syncthreads();
            if (brk) break;
        }
    }
    const int last_slot = cl_slot(cs);
    if (clustered) cluster_set_solo(cs, tid, false);

    // twist_odometry_old = R_inc^-1 * twist_odometry (reference :1139-1144)
    if (tid == 0) {
        LDS double *R = s.dwork, *Ri = s.dwork + 16;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) R[r * 3 + c] = (double)s.T[r + 4 * c];
        inverse_double_lds(R, Ri, 3);
        float Rif[9];
        for (int q = 0; q < 9; q++) Rif[q] = (float)Ri[q];
        for (int half = 0; half < 2; half++)
            for (int r = 0; r < 3; r++) {
                float acc = Rif[r * 3 + 0] * s.twist[half * 3 + 0];
                acc += Rif[r * 3 + 1] * s.twist[half * 3 + 1];
                acc += Rif[r * 3 + 2] * s.twist[half * 3 + 2];
                s.twist_old[half * 3 + r] = acc;
            }
    }
    __syncthreads();
    const bool commit = commit_ok(cs);  // false: a rendezvous of this frame timed out somewhere in the cluster -- the values
                                        // in LDS may come from stale words; the stream keeps the state of its last good frame
    if (cl_writer(cs) && !commit && tid == 0) a.stats[b].status = s.status | SF_STATUS_SYNC_TIMEOUT;
    if (cl_writer(cs) && commit) {  // ONE workgroup of a cluster stores the stream's results (all of them hold the same values)
        if (tid == 0) {
            sf_frame_stats &fs = a.stats[b];
            fs.n_outer = s.n_outer;
            fs.n_irls = s.n_irls;
            fs.pixel_iters = s.pixel_iters;
            fs.status = s.status;
            st.last_level = last_L;
            st.last_first = s.first;
            st.last_slot = last_slot;
            st.cum_frames += 1;
            st.cum_irls += s.n_irls;
            st.cum_outer += s.n_outer;
            st.cum_pixel_iters += s.pixel_iters;
            st.inv_max_c = s.inv_max_c;
            st.inv_max_d = s.inv_max_d;
            if (!a.p.segmentation_enabled) fs.kmeans_iters = 0;
        }
        if (tid < 16) st.T[tid] = s.T[tid];
        if (tid < 6) {
            st.twist_old[tid] = s.twist_old[tid];
            st.twist[tid] = s.twist[tid];
            st.twist_level[tid] = s.twist_level[tid];
        }
        if (tid < 36) st.est_cov[tid] = s.est_cov[tid];
        if (tid >= PF_WARP && tid <= PF_FILTER) st.prof[tid] += s.prof[tid];
#ifdef SF_FILTER_PROFILE
        if (tid >= 21 && tid <= 23) st.prof[tid] += s.prof[tid];
#endif
        if (tid < SF_NC) {
            st.b_segm[tid] = s.b_segm[tid];
            st.b_prior[tid] = s.b_prior[tid];
            st.lambda_t_w[tid] = s.lambda_t_w[tid];
        }
    }
    cluster_barrier(cs, tid);  // the stream state is visible to the stages that follow (in every workgroup)
}


// ---------------------------------------------------------------------------------------------
//  test support (sf_get_jacobian_rows): the rows of A and B of the LAST outer iteration of stream b, expanded
//  from the factored per-pixel form the passes evaluate: a_c = pc g1 + qc g2, a_d = twd g3 + pd g1 + qd g2,
//  b_c = -bct, b_d = -bdt (see above). out = 14 planes of n pixels: a_c[0..5], b_c, a_d[0..5], b_d; pixels outside
//  validPixels get NaN in plane 0. Never part of a solve.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void debug_rows(const KArgs &a, int b, float *out, int gtid, int gstride) {
    const StreamState &st = a.state[b];
    const int L = st.last_level;
    const int n = a.ln[L], rows_i = a.lrows[L], cols_i = a.lcols[L];
    const size_t rb = (size_t)st.last_slot * a.n0;
    const float f = float(cols_i) / (2.f * a.tan_half_fovh);
    LevelGeom g;
    g.rows_i = rows_i;
    g.inv_rows = 1.f / float(rows_i);
    g.disp_u_i = 0.5f * float(cols_i - 1);
    g.disp_v_i = 0.5f * float(rows_i - 1);
    g.inv_f_pyr = 2.f * a.tan_half_fovh / float(cols_i);
    g.inv_f_w = 1.f / f;
    g.f_inv = f;
    g.kph = a.p.k_photometric_res;
    g.inv_max_c = st.inv_max_c;
    g.inv_max_d = st.inv_max_d;
    g.first = st.last_first;
    const float *dnew = pyr_level(a, b, 0, 0, L);
    for (int idx = gtid; idx < n; idx += gstride) {
        const float dw = a.rec[R_DW][rb + idx];
#if SF_REFORDER && SF_RO_BEHIND
        const bool in_valid = a.rec_lab[rb + idx] != SF_INVALID_LABEL;  // (this build's records keep the warp's own sign)
#else
        const bool in_valid = dw > 0.f;
#endif
        if (!in_valid) {
            out[idx] = __int_as_float(0x7fc00000);
            continue;
        }
        float fu, fv;
        split_index(g, idx, fu, fv);
        PixFact<float> p;
        fact_from_record<float>(g, fu, fv, dnew[idx], dw, a.rec[R_DCU][rb + idx], a.rec[R_DCV][rb + idx], a.rec[R_DCT][rb + idx],
                                a.rec[R_DDU][rb + idx], a.rec[R_DDV][rb + idx], p);
        const float g1[6] = {-1.f, 0.f, p.xd, p.xyd, -p.xxd, p.y};
        const float g2[6] = {0.f, -1.f, p.yd, p.yyd, -p.xyd, -p.x};
        const float g3[6] = {0.f, 0.f, 1.f, p.y, -p.x, 0.f};
        for (int c = 0; c < 6; c++) {
            out[(size_t)c * n + idx] = vfma(p.pc, g1[c], p.qc * g2[c]);
            out[(size_t)(7 + c) * n + idx] = vfma(p.twd, g3[c], vfma(p.pd, g1[c], p.qd * g2[c]));
        }
        out[(size_t)6 * n + idx] = -p.bct;
        out[(size_t)13 * n + idx] = -p.bdt;
    }
}

// ---------------------------------------------------------------------------------------------
//  measurement support: the two IRLS streaming passes in isolation, over the level-0 records the
//  last solve left behind (tools/pass_microbench.py, sf_microbench_pass)
// ---------------------------------------------------------------------------------------------
template <int WHICH, int VAR>
__device__ void microbench_pass(const KArgs &a, int b, int slice, int slices, int reps, LDS SolveShared &s, int tid) {
    const StreamState &st = a.state[b];
    if (tid < SF_NC) s.b_segm[tid] = a.p.segmentation_enabled ? st.b_segm[tid] : 1.f;
    if (tid < 6) s.Var[tid] = st.twist_level[tid];
    if (tid == 0) {
        s.inv_max_c = st.inv_max_c;
        s.inv_max_d = st.inv_max_d;
        s.rec_slot = b;
        s.aver_res = 0.002f;
        s.first = 0;
        s.n_valid = a.ln[0];
        const int per = ((a.ln[0] / slices) + 1) & ~1;  // even: the passes walk pixel pairs
        s.px_begin = slice * per;
        s.px_end = (slice == slices - 1) ? a.ln[0] : min(a.ln[0], (slice + 1) * per);
    }
    if (tid < SF_NC) s.lab_sum[tid] = 0;
    __syncthreads();
    for (int r = 0; r < reps; r++) {
        if (WHICH == 1)
            irls_pass1<VAR>(a, b, 0, s, tid);
        else
            irls_pass2<VAR>(a, b, 0, s, tid);
        __syncthreads();
    }
}
