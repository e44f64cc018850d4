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

// ---------------------------------------------------------------------------------------------
//  ALL builds: the ordered float splat for the COARSE levels of the product (round 4). The attribution of
//  profiles/PARITY.md says where the product's pose excursions come from: with everything else as in the product, the
//  reference's ordered float sums at the coarse levels alone take 42 % of the frames past the pose bar and 44 % of the
//  iteration-count mismatches away (5000 sequences at 160 x 120: 38 -> 22 frames, 25 -> 14). Levels of at most
//  SF_ORDERED_SPLAT_MAX_PIXELS pixels (QVGA: image levels 3 and 4, 1.5 % of the pyramid's pixels; with 8192 -- level 2 as well --
//  the excursions were the same within their noise and the cost double) take the ordered splat below in every build of the
//  product; larger levels keep the exact integer sums (sf_device_common.h). The per-cell source lists of the fall-back are
//  scratch of the WORKGROUP that runs the warp (KArgs::ro_list: one block of 256 KB per stream or resident workgroup).
// ---------------------------------------------------------------------------------------------
#ifndef SF_ORDERED_COARSE_SPLAT
#define SF_ORDERED_COARSE_SPLAT 1
#endif
#ifndef SF_ORDERED_SPLAT_MAX_PIXELS
#define SF_ORDERED_SPLAT_MAX_PIXELS 2048  // (<= SF_CLUSTER_SOLO_PIXELS: a cluster's workgroups run such levels each on its own)
#endif

// ---------------------------------------------------------------------------------------------
//  the splat of warpImagesAccurateInverse / computeResidualsAgainstPreviousImage in the reference's order
// ---------------------------------------------------------------------------------------------
struct RoTaps {
    int n;          // 0: rejected, 1: within 5 centi-pixels of a pixel centre (weight 200), 4: bilinear
    int cell[4];    // flat index of the target cells, taps in the reference's order ur, ul, dr, dl (:846-867)
    int tv[4], tu[4];  // ... and their (row, column)
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
        t.tv[0] = delta_u > delta_d ? qv : qv + 1;
        t.tu[0] = delta_r > delta_l ? qu : qu + 1;
        t.w[0] = 200;
    } else {
        t.n = 4;
        t.tv[0] = qv + 1;  t.tu[0] = qu + 1;  t.w[0] = delta_l + delta_d;
        t.tv[1] = qv + 1;  t.tu[1] = qu;      t.w[1] = delta_r + delta_d;
        t.tv[2] = qv;      t.tu[2] = qu + 1;  t.w[2] = delta_l + delta_u;
        t.tv[3] = qv;      t.tu[3] = qu;      t.w[3] = delta_r + delta_u;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) t.cell[k] = t.tv[k] + t.tu[k] * g.rows_i;  // (taps beyond t.n are not read)
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

// The same sums for a level of at most SPLAT_TV rows, WITHOUT lists in memory: such a level is walked in tiles of SPLAT_TU
// whole columns, i.e. in the reference's source order; a tile's targets lie in a window of the warped image that is kept in
// LDS as three running float sums per cell. The window is LOADED from the cells (the running sums of earlier tiles), the
// tile's taps are added in ascending source index per cell, and the window is stored back. The order inside a tile comes
// from rounds: every pending tap posts its source index to its cell with ds_min, the tap whose index the cell then holds is
// the cell's next contribution in the reference's order -- it is added (a plain read-modify-write: a cell has one owner per
// round) and the cell is released; as many rounds as the most contested cell of the tile has taps (4 - 8). A tap outside its
// tile's window (the window is the tile's size + SPLAT_MARGIN: strong local stretch) ends the attempt: the caller then takes
// ro_splat above for the whole level. Returns (uniformly) whether the cells hold the result, in ro_splat's format.
template <class Src>
__device__ __forceinline__ bool ordered_tile_splat(const SplatGeom &g, const LevelCoord &lc, int rows_i, int cols_i, const Src &src,
                                                   gptr<long long> acc_d, gptr<long long> acc_i, LDS SplatWin &win, int tid) {
    const int lane = tid & 63, n = rows_i * cols_i;
    LDS vfloat2 *sums = (LDS vfloat2 *)win.d;  // {sum w depth, sum w intensity} of a window cell
    LDS float *wsum = (LDS float *)win.i;      // [2 c]: sum w; [2 c + 1] (as unsigned): the smallest pending source index
    LDS unsigned *owner = (LDS unsigned *)win.i;
    for (int idx = tid; idx < n; idx += SF_NT) {
        gst(acc_d, idx, 0ll);
        gst(acc_i, idx, 0ll);
    }
    __syncthreads();
    const int tiles = (cols_i + SPLAT_TU - 1) / SPLAT_TU;
    for (int t = 0; t < tiles; t++) {
        const int tu0 = t * SPLAT_TU;
        RoTaps tp[SPLAT_PX];
        unsigned sidx[SPLAT_PX];
        int vtop = 0, utop = 0;
        if (tid == 0) {
            win.vmin = 0x7fffffff;
            win.umin = 0x7fffffff;
            win.ovf[0] = win.ovf[1] = win.ovf[2] = 0;  // workgroup-wide flags of this tile (set with ds_or, read behind a barrier)
        }
#pragma unroll
        for (int k = 0; k < SPLAT_PX; k++) {
            const int v = lane, u = tu0 + (tid >> 6) + k * (SF_NT / 64);
            tp[k].n = 0;
            sidx[k] = (unsigned)(v + u * rows_i);
            if (v < rows_i && u < cols_i) ro_project(g, src, lc, (int)sidx[k], tp[k]);
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (q < tp[k].n) {
                    vtop = max(vtop, 0x7fffffff - tp[k].tv[q]);
                    utop = max(utop, 0x7fffffff - tp[k].tu[q]);
                }
        }
        SF_DPP_REDUCE(vtop, dpp_i32, sf_op_maxi)
        SF_DPP_REDUCE(utop, dpp_i32, sf_op_maxi)
        // (read lane 63 HERE, by every lane: inside the lane-0 block below the compiler sinks the last v_max of the
        // reduction into that block, where lane 63 does not execute it -- umin came out as "no tap" for levels of < 64 rows)
        const int vmin = 0x7fffffff - __builtin_amdgcn_readlane(vtop, 63), umin = 0x7fffffff - __builtin_amdgcn_readlane(utop, 63);
        __syncthreads();  // origin initialised; the previous tile's window stored
        if (lane == 0) {
            lds_min(&win.vmin, vmin);
            lds_min(&win.umin, umin);
        }
        __syncthreads();
        const int wv0 = uniform_i(win.vmin), wu0 = uniform_i(win.umin);
        if (wu0 == 0x7fffffff) continue;  // no source pixel of the tile reaches the image
        // the running sums of the window's cells
        for (int q = tid; q < WIN_CELLS; q += SF_NT) {
            const int du = q / WIN_V, dv = q - du * WIN_V;
            const int v = wv0 + dv, u = wu0 + du;
            vfloat2 s2 = {0.f, 0.f};
            float w = 0.f;
            if (v < rows_i && u < cols_i) {
                const long long sd = gld_agent_i64(acc_d, v + u * rows_i), si = gld_agent_i64(acc_i, v + u * rows_i);
                float sx, sy;
                ro_unpack_cell(sd, sx, sy);
                s2.x = sx;
                s2.y = sy;
                w = __uint_as_float((unsigned)((unsigned long long)si & 0xffffffffu));
            }
            sums[q] = s2;
            wsum[2 * q] = w;
            owner[2 * q + 1] = 0xffffffffu;
        }
        // window-local cell of every tap
        int cell[SPLAT_PX][4];
        unsigned pend = 0;
        bool outside = false;
#pragma unroll
        for (int k = 0; k < SPLAT_PX; k++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                cell[k][q] = 0;
                if (q < tp[k].n) {
                    const int dv = tp[k].tv[q] - wv0, du = tp[k].tu[q] - wu0;  // >= 0: the origin is the tile's minimum
                    if (dv < WIN_V && du < WIN_U) {
                        cell[k][q] = dv + du * WIN_V;
                        pend |= 1u << (4 * k + q);
                    } else {
                        outside = true;
                    }
                }
            }
        if (outside) lds_or(&win.ovf[2], 1u);
        __syncthreads();  // the window is loaded, the flag complete
        if (uniform_i((int)win.ovf[2]) != 0) return false;
        for (int round = 0;; round++) {
#pragma unroll
            for (int k = 0; k < SPLAT_PX; k++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (pend & (1u << (4 * k + q))) lds_min(&owner[2 * cell[k][q] + 1], sidx[k]);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < SPLAT_PX; k++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if ((pend & (1u << (4 * k + q))) && owner[2 * cell[k][q] + 1] == sidx[k]) {
                        const int c = cell[k][q];
                        const float wf = (float)tp[k].w[q];
                        vfloat2 s2 = sums[c];
                        s2.x += wf * tp[k].depth_w;
                        s2.y += wf * tp[k].inten_w;
                        sums[c] = s2;
                        wsum[2 * c] += wf;
                        owner[2 * c + 1] = 0xffffffffu;  // (a lane that still reads it sees the winner's index or this: not its own)
                        pend &= ~(1u << (4 * k + q));
                    }
            // anybody left? Two flags take turns: the one of the next round is cleared while nobody reads or sets it
            if (pend != 0) lds_or(&win.ovf[round & 1], 1u);
            if (tid == 0) win.ovf[(round + 1) & 1] = 0;
            __syncthreads();
            if (uniform_i((int)win.ovf[round & 1]) == 0) break;
            if (round >= 4 * SPLAT_TV * SPLAT_TU) return false;  // (cannot happen: a round retires a tap of every contested cell; never spin on a bug)
        }
        for (int q = tid; q < WIN_CELLS; q += SF_NT) {
            const int du = q / WIN_V, dv = q - du * WIN_V;
            const int v = wv0 + dv, u = wu0 + du;
            if (v < rows_i && u < cols_i) {
                const vfloat2 s2 = sums[q];
                gst(acc_d, v + u * rows_i, (long long)(((unsigned long long)__float_as_uint(s2.y) << 32) | __float_as_uint(s2.x)));
                gst(acc_i, v + u * rows_i, (long long)(unsigned long long)__float_as_uint(wsum[2 * q]));
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < n; idx += SF_NT) {  // the quotients (:876-880); untouched cells keep acc_i == 0
        const long long si = gld_agent_i64(acc_i, idx);
        const float w = __uint_as_float((unsigned)((unsigned long long)si & 0xffffffffu));
        if (w != 0.f) {
            float ds, is;
            ro_unpack_cell(gld_agent_i64(acc_d, idx), ds, is);
            const float iw = is / w, dw = ds / w;
            gst(acc_d, idx, (long long)(((unsigned long long)__float_as_uint(iw) << 32) | __float_as_uint(dw)));
            gst(acc_i, idx, 1ll);
        }
    }
    __syncthreads();
    return true;
}

// the ordered float splat of a level: in LDS tiles when the level is at most one tile high, else (or when a tile's targets
// do not fit its window) through the per-cell lists
template <class Src>
__device__ __forceinline__ void ordered_splat(const KArgs &a, const SplatGeom &g, const LevelCoord &lc, int rows_i, int cols_i,
                                              const Src &src, gptr<long long> acc_d, gptr<long long> acc_i, gptr<int> list,
                                              LDS SplatWin &win, int tid, long long *fallbacks = nullptr) {
#ifndef SF_ORDERED_TILE_SPLAT
#define SF_ORDERED_TILE_SPLAT 1  // 0: always the lists (A/B and bisection builds)
#endif
    // Both conditions go through readfirstlane: `rows_i` arrives in a VGPR (a member of KArgs read through a pointer the
    // compiler cannot prove uniform) and a branch on it is compiled as a DIVERGENT one -- EXEC masks around code that holds
    // barriers. That is how round 4's "address 0" fault came about (profiles/HISTORY.md, round 5; tools/diag/exec_lint.py): the
    // exit block of ro_splat's cell loop got a register copy IN FRONT of the `s_or_b64 exec` that re-enables its lanes.
    if (SF_ORDERED_TILE_SPLAT && uniform_i(rows_i) <= SPLAT_TV) {
        if (uniform_i(ordered_tile_splat(g, lc, rows_i, cols_i, src, acc_d, acc_i, win, tid) ? 1 : 0)) return;
        if (tid == 0 && fallbacks) *fallbacks += 1;  // (a counter of the stream's profile: tests want to know that this path ran)
    }
    ro_splat(g, lc, rows_i * cols_i, src, acc_d, acc_i, list, tid);
}

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
#ifndef SF_RO_SEQ64
#define SF_RO_SEQ64 1   // 0: the fp64 sums ([C1]: AtA / AtB, sum |res|, ||res||^2) as per-lane partial sums + a tree, not row by row
#endif

#define RO_CHUNK 1024    // pixels per trip of the ordered per-cluster sums

struct RoChunk {
    float val[RO_CHUNK];
    uint8_t lab[RO_CHUNK];   // cluster of the entry, SF_INVALID_LABEL: no entry
    uint8_t flag[RO_CHUNK];  // bit 0: counts as non-Null / contributes `val`; bit 1: in validPixels
};

// the [C1] sums of the oracle are fp64 sums of float terms, row after row. Partial sums per lane round differently in the
// 16th digit, and once in some ten million sums that moves the float the sum is converted to (hunt seed 61826, frame 6):
// the reference-order build walks them row by row too -- a chunk's terms go to LDS, one lane per sum adds them front to back
struct RoChunk2 {  // sum |res| and ||res||^2: two float terms per pixel next to the per-cluster chunk
    RoChunk c;
    float rc[RO_CHUNK], rd[RO_CHUNK];
};
#define RO_ROWS_CHUNK 256
struct RoRows {  // pass 1: the two weighted rows of a pixel, [entry][pixel]; entries 0..5 + 6 (Bw): colour row, 7..12 + 13: depth row
    float aw[14][RO_ROWS_CHUNK];
    uint8_t ok[RO_ROWS_CHUNK];
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

#endif  // SF_REFORDER

#if SF_REFORDER
#define RO_SPLAT_AT_(L) RO_SPLAT_AT(L)
#else
#define RO_SPLAT_AT_(L) false
#endif
// does level L (n pixels) of this launch take the ordered float splat? G: workgroups that share the level right now
__device__ __forceinline__ bool splat_ordered(int L, int n, int G) {
#if SF_REFORDER
    (void)n; (void)G;
    return RO_SPLAT_AT_(L);
#else
    (void)L;
    return SF_ORDERED_COARSE_SPLAT && G == 1 && n <= SF_ORDERED_SPLAT_MAX_PIXELS;
#endif
}
// the source lists this workgroup uses: the record slot's (reference-order build: every level, any size) or the workgroup's own block
__device__ __forceinline__ gptr<int> ro_list_of(const KArgs &a, size_t rb, int b) {
#if SF_REFORDER
    (void)b;
    return as_global(a.ro_list + rb * RO_LIST_K);
#else
    // a block per stream when the handle has one for each (two frames of a stream never run at the same time), else a block
    // per workgroup of the launch (the grid never exceeds KArgs::ro_blocks then); a cluster's workgroups each take their own
    (void)rb;
    const size_t blk = (a.cluster_g == 0 && a.batch <= a.ro_blocks) ? (size_t)b : (size_t)blockIdx.x;
    return as_global(a.ro_list + blk * SF_ORDERED_SPLAT_MAX_PIXELS * RO_LIST_K);
#endif
}

