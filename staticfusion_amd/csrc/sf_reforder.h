// sf_reforder.h — the REFERENCE-ORDER build of the frame kernel (-DSF_REFORDER=1, libsf_hip_reforder.so).
//
// A parity instrument, not a product: every place where the product build replaces one of the reference's
// deterministic float operation sequences by something faster (and differently rounded) is put back, whatever it
// costs, so that what remains between this build and the CPU restatement is only what the reference itself leaves
// open (the internal order of its Eigen GEMM, FrontEnd.cpp:640-641) or delegates to a library (libm, Eigen's
// eigensolver). profiles/PARITY.md has the table: this build against the oracle over the hunt seeds, then ONE product
// shortcut switched back on at a time (the SF_RO_* switches below).
//
//   what the reference does (its source fixes the order)                        product build          this build
//   warp / 5-frame residual splat: float `+=` of float(w) * depth_w per target  exact Q26 / Q28        the taps of a cell
//     cell, source pixels u outer / v inner (FrontEnd.cpp:804-871, :947-1034)     integer sums           gathered with their source
//                                                                                                       index, sorted, added in
//                                                                                                       that order; IEEE division
//   Jacobian rows a_c, a_d with the expression order of :552-585                 factored rows, fmaf    that expression order
//   pre-weights / Cauchy weights sqrtf(1 / x) (:494-502, :626-634)                v_rsq / v_rcp (1 ulp)  IEEE
//   AtA, AtB = Aw^T Aw, Aw^T Bw (:640-641; Eigen GEMM, order not in the source)   fp32 lane sums ->      per row: fp64 products
//                                                                                 fp64 every 128 terms   (exact) + fp64 sums = [C1]
//   res = -B + sum_k Var(k) A.col(k), k ascending (:644-646)                      three dot products     that order
//   per-cluster float sums, validPixels order (:650-664), level order             exact Q32.32 integers  sequential float sums in
//     (SegmentationBackground.cpp:61-81), (:1041-1062)                                                   that order
//   initial mean |res| from the rows' B (:588-590)                                 scaled sums of the     a pass over the rows
//                                                                                 linearisation
//   SelfAdjointEigenSolver (:719; oracle [C5]: cyclic Jacobi in fp64)              round-robin Jacobi     cyclic Jacobi
//   ||twist_level|| (:1130; oracle: fp64 sum of the float squares)                 fp64 squares           the oracle's form
//   validPixels keeps a point warped behind the camera (:415-427)                  dropped                kept
//
// Performance is irrelevant here (a QVGA frame takes tens of milliseconds on one workgroup); one workgroup per stream only.
#pragma once

#define RO_LIST_K 32     // source pixels remembered per target cell; a cell with more is summed by a scan over the level

#if SF_REFORDER

// ONE product shortcut back on at a time (attribution builds, tools/build_variant.sh):
#ifndef SF_RO_SPLAT
#define SF_RO_SPLAT 1   // 0: the product's exact integer splat sums (divided with IEEE division)
#endif
// ... or only at some levels of the pyramid: the ordered float splat runs at image levels [SF_RO_SPLAT_MIN_LEVEL, SF_RO_SPLAT_MAX_LEVEL]
// (0 = full resolution), the product's integer sums (IEEE division) at the others -- which levels carry the sensitivity
#ifndef SF_RO_SPLAT_MIN_LEVEL
#define SF_RO_SPLAT_MIN_LEVEL 0
#endif
#ifndef SF_RO_SPLAT_MAX_LEVEL
#define SF_RO_SPLAT_MAX_LEVEL 99
#endif
#define RO_SPLAT_AT(L) (SF_RO_SPLAT && (L) >= SF_RO_SPLAT_MIN_LEVEL && (L) <= SF_RO_SPLAT_MAX_LEVEL)
#ifndef SF_RO_ROWS
#define SF_RO_ROWS 1    // 0: the product's factored rows / three dot products (with SF_ROWS_FMA as given)
#endif
#ifndef SF_RO_P1_FP64
#define SF_RO_P1_FP64 1 // 0: the product's fp32 lane sums, flushed into fp64 every SF_P1_FLUSH pixel pairs
#endif
#ifndef SF_RO_LABSUM
#define SF_RO_LABSUM 1  // 0: the product's exact Q32.32 per-cluster sums
#endif
#ifndef SF_RO_JACOBI
#define SF_RO_JACOBI 1  // 0: the product's round-robin Jacobi
#endif
#ifndef SF_RO_INIT_RES
#define SF_RO_INIT_RES 1  // 0: the product's initial mean |res| from the linearisation's scaled sums
#endif
#ifndef SF_RO_BEHIND
#define SF_RO_BEHIND 1  // 0: the product's rule for points warped behind the camera
#endif

#define RO_CHUNK 1024    // pixels per trip of the ordered per-cluster sums

struct RoChunk {
    float val[RO_CHUNK];
    uint8_t lab[RO_CHUNK];   // cluster of the entry, SF_INVALID_LABEL: no entry
    uint8_t flag[RO_CHUNK];  // bit 0: counts as non-Null / contributes `val`; bit 1: in validPixels
};

// ---------------------------------------------------------------------------------------------
//  sequential per-cluster float sums in pixel order: 24 lanes, one per cluster, walk the chunk front to back
// ---------------------------------------------------------------------------------------------
struct RoLabelAcc {
    float sum;
    int n_all, n_val, n_valid;  // entries of the cluster, entries with bit 0, entries with bit 1
};
__device__ __forceinline__ void ro_label_walk(const LDS RoChunk &c, int m, int tid, RoLabelAcc &a) {
    if (tid < SF_NC) {
        for (int q = 0; q < m; q++) {
            if ((int)c.lab[q] != tid) continue;
            const int f = c.flag[q];
            a.n_all++;
            if (f & 1) {
                a.n_val++;
                a.sum += c.val[q];  // the reference's `+=` on a float, in the reference's pixel order
            }
            if (f & 2) a.n_valid++;
        }
    }
}

// ---------------------------------------------------------------------------------------------
//  the splat of warpImagesAccurateInverse / computeResidualsAgainstPreviousImage in the reference's order
// ---------------------------------------------------------------------------------------------
struct RoTaps {
    int n;          // 0: rejected, 1: within 5 centi-pixels of a pixel centre (weight 200), 4: bilinear
    int cell[4];    // flat index of the target cells, taps in the reference's order ur, ul, dr, dl (:846-867)
    int w[4];
    float depth_w, inten_w;
};

template <class Src>
__device__ __forceinline__ void ro_project(const SplatGeom &g, const Src &src, const LevelCoord &lc, int idx, RoTaps &t) {
    int u, v;
    split_uv(lc, idx, u, v);
    float z, xr, yr, iw;
    t.n = 0;
    if (!src.load(v, u, idx, z, xr, yr, iw)) return;
    const float x_w = g.T[0] * xr + g.T[1] * yr + g.T[2] * z + g.T[3];
    const float y_w = g.T[4] * xr + g.T[5] * yr + g.T[6] * z + g.T[7];
    const float depth_w = g.T[8] * xr + g.T[9] * yr + g.T[10] * z + g.T[11];
    const int uw = cvt_trunc_x86(100.f * (g.f * x_w / depth_w + g.disp_u_i));
    const int vw = cvt_trunc_x86(100.f * (g.f * y_w / depth_w + g.disp_v_i));
    if (!((uw >= 0) && (uw < g.cols_lim) && (vw >= 0) && (vw < g.rows_lim))) return;
    const int qu = uw / 100, qv = vw / 100;
    const int delta_l = uw - 100 * qu, delta_r = 100 - delta_l, delta_d = vw - 100 * qv, delta_u = 100 - delta_d;
    t.depth_w = depth_w;
    t.inten_w = iw;
    if (min(delta_r, delta_l) + min(delta_u, delta_d) < 5) {
        t.n = 1;
        t.cell[0] = (delta_u > delta_d ? qv : qv + 1) + (delta_r > delta_l ? qu : qu + 1) * g.rows_i;
        t.w[0] = 200;
    } else {
        t.n = 4;
        t.cell[0] = (qv + 1) + (qu + 1) * g.rows_i;  t.w[0] = delta_l + delta_d;
        t.cell[1] = (qv + 1) + qu * g.rows_i;        t.w[1] = delta_r + delta_d;
        t.cell[2] = qv + (qu + 1) * g.rows_i;        t.w[2] = delta_l + delta_u;
        t.cell[3] = qv + qu * g.rows_i;              t.w[3] = delta_r + delta_u;
    }
}

// After the call cell idx of the level holds: acc_i[idx] != 0 <=> some source pixel reached it (wacu != 0), and then
// acc_d[idx] = bits(depthWarped) | bits(intensityWarped) << 32, both already divided by the float weight sum.
template <class Src>
__device__ __forceinline__ void ro_splat(const SplatGeom &g, const LevelCoord &lc, int n, const Src &src, gptr<long long> acc_d,
                                         gptr<long long> acc_i, gptr<int> list, int tid) {
    for (int idx = tid; idx < n; idx += SF_NT) gst(acc_i, idx, 0ll);
    __syncthreads();
    // who reaches which cell: a counter per cell (the low word of acc_i) and up to RO_LIST_K source indices
    for (int idx = tid; idx < n; idx += SF_NT) {
        RoTaps t;
        ro_project(g, src, lc, idx, t);
        for (int k = 0; k < t.n; k++) {
            const int slot = __hip_atomic_fetch_add((gptr<int>)((__attribute__((address_space(1))) char *)acc_i + (size_t)t.cell[k] * 8u), 1,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (slot < RO_LIST_K) list[(size_t)t.cell[k] * RO_LIST_K + slot] = idx;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // (one workgroup: the lists and counters are read by other waves of it)
    __syncthreads();
    // per cell: its source pixels in ascending index = the order in which the reference's loops (j outer, i inner) add them
    for (int idx = tid; idx < n; idx += SF_NT) {
        const int cnt = (int)(gld_agent_i64(acc_i, idx) & 0xffffffffll);
        if (cnt == 0) continue;  // acc_i stays 0: never reached
        float dsum = 0.f, isum = 0.f, wsum = 0.f;
        auto add_source = [&](int s) {
            RoTaps t;
            ro_project(g, src, lc, s, t);
            for (int k = 0; k < t.n; k++)
                if (t.cell[k] == idx) {
                    const float wf = (float)t.w[k];  // `w_ur * depth_w`: int -> float, then a float product (:848-866; 200.f at :840)
                    dsum += wf * t.depth_w;
                    isum += wf * t.inten_w;
                    wsum += wf;
                }
        };
        if (cnt <= RO_LIST_K) {
            int e[RO_LIST_K];
            for (int k = 0; k < cnt; k++) e[k] = list[(size_t)idx * RO_LIST_K + k];
            for (int a = 1; a < cnt; a++) {  // insertion sort: the slots were handed out in whatever order the atomics ran
                const int x = e[a];
                int b2 = a - 1;
                while (b2 >= 0 && e[b2] > x) {
                    e[b2 + 1] = e[b2];
                    b2--;
                }
                e[b2 + 1] = x;
            }
            for (int k = 0; k < cnt; k++) add_source(e[k]);
        } else {
            for (int s = 0; s < n; s++) add_source(s);  // more contributors than the list holds: every source pixel, in order
        }
        const float iw = isum / wsum, dw = dsum / wsum;  // :876-880
        gst(acc_d, idx, (long long)(((unsigned long long)__float_as_uint(iw) << 32) | __float_as_uint(dw)));
        gst(acc_i, idx, 1ll);
    }
    __syncthreads();
}

__device__ __forceinline__ void ro_unpack_cell(long long sd, float &dw, float &iw) {
    dw = __uint_as_float((unsigned)((unsigned long long)sd & 0xffffffffu));
    iw = __uint_as_float((unsigned)((unsigned long long)sd >> 32));
}

#endif  // SF_REFORDER
