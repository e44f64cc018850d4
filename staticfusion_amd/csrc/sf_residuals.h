// sf_residuals.h — the stages that follow runSolver in the drivers' frame loop:
//   computeResidualsAgainstPreviousImage  (reference FrontEnd.cpp:896-1069)
//   buildSegmImage                        (reference SegmentationBackground.cpp:176-197)
//   ring-buffer push                      (reference StaticFusion-datasets.cpp:182-184)
#pragma once

#include "sf_cluster.h"
#include "sf_device_common.h"
#include "sf_smallmath.h"

struct ResShared {
    float Tinv[16];
    long long lab_sum[SF_NC];
    int lab_cnt[SF_NC];
    double dwork[32];
    union {
        struct {
            SplatWin win;
            SplatMarks marks;
        };
#if SF_REFORDER
        RoChunk ro;  // reference-order build: a chunk of the ordered per-cluster sums
#endif
    };
};

__device__ __forceinline__ int level0_label(const KArgs &a, gptr<const uint8_t> labels0, int idx) {
    // without segmentation the reference's clusterAllocation[0] stays at its constructor value 0
    return a.p.segmentation_enabled ? (int)gld(labels0, idx) : 0;
}

__device__ __noinline__ void stage_residuals(const KArgs &a, int b, int index, bool push, LDS ResShared &s, LDS ClusterShared &cs, int tid) {
    StreamState &st = a.state[b];
    const int rows = a.lrows[0], cols = a.lcols[0], n = a.ln[0];
    const int G = cl_G(cs), rank = cl_rank(cs);
    const size_t sb = (size_t)b * a.n_tot, rb = (size_t)cl_slot(cs) * a.n0;
    const auto dcur = as_global(pyr_level(a, b, 0, 0, 0)), icur = as_global(pyr_level(a, b, 0, 1, 0));  // depthCurrent / intensityCurrent
    const auto labels0 = as_global((const uint8_t *)a.labels + sb);
    const int idx_to_warp = (index - SF_HISTORY) % SF_HISTORY;
    const auto dbuf = as_global((const float *)a.hist_d + ((size_t)idx_to_warp * a.batch + b) * a.n0);
    const auto ibuf = as_global((const float *)a.hist_i + ((size_t)idx_to_warp * a.batch + b) * a.n0);
    // push: the slot warped from is the slot depthBuffer[index % 5] = depthCurrent goes to (datasets.cpp:182-184); the pass
    // below stores the current images there once it has read the old depth of a pixel (everything else that reads the
    // slot -- the splat -- is behind the rendezvous in front of the pass)
    const auto dpush = as_global(a.hist_d + ((size_t)idx_to_warp * a.batch + b) * a.n0);
    const auto ipush = as_global(a.hist_i + ((size_t)idx_to_warp * a.batch + b) * a.n0);
    const auto acc_d = as_global(a.acc_d + rb), acc_i = as_global(a.acc_i + rb);

    if (tid == 0) {
        // T = prod odomBuffer[(index-4 .. index-1) % 5] * T_odometry, then inverse (:901-909)
        float T[16], Tn[16];
        for (int q = 0; q < 16; q++) T[q] = (q % 5 == 0) ? 1.f : 0.f;
        for (int i = index - SF_HISTORY + 1; i < index; i++) {
            mul4_cm(T, st.hist_T[i % SF_HISTORY], Tn);
            for (int q = 0; q < 16; q++) T[q] = Tn[q];
        }
        mul4_cm(T, st.T, Tn);
        inverse4_cm(Tn, s.Tinv, s.dwork);
    }
    if (tid < SF_NC) {
        s.lab_sum[tid] = 0;
        s.lab_cnt[tid] = 0;
    }
    const bool ordered = uniform_i(splat_ordered(0, n, G) ? 1 : 0) != 0;  // full resolution: the reference-order build, or an image of at most 2048 pixels
    const bool lazy = ordered || uniform_i(splat_lazy_ok(rows, cols, G) ? 1 : 0) != 0;  // see solve_warp
    if (!lazy)
    for (int idx = tid + rank * SF_NT; idx < n; idx += SF_NT * G) {  // agent-scope stores: see solve_warp
        if (G > 1) {
            __hip_atomic_store(acc_d + idx, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(acc_i + idx, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            gst(acc_d, idx, 0ll);
            gst(acc_i, idx, 0ll);
        }
    }
    cluster_rendezvous(cs, tid);

    const float inv_f_i = 2.f * a.tan_half_fovh / float(cols);
    SplatGeom g;
    g.f = float(cols) / (2.f * a.tan_half_fovh);
    g.disp_u_i = 0.5f * (cols - 1);
    g.disp_v_i = 0.5f * (rows - 1);
    g.cols_lim = 100 * (cols - 1);
    g.rows_lim = 100 * (rows - 1);
    g.rows_i = rows;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) g.T[r * 4 + c] = uniform_f(s.Tinv[r + 4 * c]);

    struct Src {
        gptr<const float> dbuf, ibuf, dcur;
        float inv_f_i, disp_u_i, disp_v_i;
        __device__ __forceinline__ bool load(int v, int u, int idx, float &z, float &xr, float &yr, float &iw) const {
            z = gld(dbuf, idx);
            iw = gld(ibuf, idx);
            const float dc = gld(dcur, idx);
            xr = (inv_f_i * (float(u) - disp_u_i)) * z;  // xxBuffer / yyBuffer (:922-926)
            yr = (inv_f_i * (float(v) - disp_v_i)) * z;
            return z != 0.f && dc != 0.f;
        }
    } src{dbuf, ibuf, dcur, inv_f_i, g.disp_u_i, g.disp_v_i};
    if (ordered) {
        LevelCoord lc0 = level_coord(a, 0);
        // an image of at most SPLAT_TV rows in LDS tiles, else the per-cell lists. (Round 4 called ro_splat directly here after
        // this instantiation had faulted at address 0: a miscompiled divergent branch, profiles/HISTORY.md round 5 -- the
        // conditions of ordered_splat are scalar now and tools/diag/exec_lint.py checks every built library for the pattern.)
        ordered_splat(a, g, lc0, rows, cols, src, acc_d, acc_i, ro_list_of(a, rb, b), s.win, tid, &st.prof[PF_ORDERED_FALLBACKS]);
    } else
        tiled_splat(g, rows, cols, src, acc_d, acc_i, s.win, s.marks, tid, rank, G, lazy, &st.prof[PF_SPLAT_REPLAYS]);
    cluster_rendezvous(cs, tid);

#if SF_REFORDER
    // residuals, cluster-wise (:1036-1068), as the reference sums them: one float per cluster, pixels j outer / i inner
    {
        const float kph_ro = a.p.k_photometric_res;
        RoLabelAcc la{0.f, 0, 0, 0};
        for (int base = 0; base < n; base += RO_CHUNK) {
            for (int q = tid; q < RO_CHUNK; q += SF_NT) {
                const int idx = base + q;
                int lab = SF_INVALID_LABEL;
                float val = 0.f;
                if (idx < n) {
                    const long long sd = gld_agent_i64(acc_d, idx), si = gld_agent_i64(acc_i, idx);
                    const float dc = gld(dcur, idx), db = gld(dbuf, idx), ic = gld(icur, idx);
                    const int lb = level0_label(a, labels0, idx);
                    if (push) {
                        gst(dpush, idx, dc);
                        gst(ipush, idx, ic);
                    }
                    if (si != 0 && dc != 0.f) {
                        float dw, iw;
                        if (ordered)
                            ro_unpack_cell(sd, dw, iw);
                        else
                            normalise_acc(sd, si, dw, iw);
                        if (dw != 0.f && lb < SF_NC) {
                            const float idiff = (db != 0.f) ? ic : 0.f;  // intensity_diff (:937,1022)
                            val = fabsf(dc - dw) + kph_ro * fabsf(idiff - iw);
                            lab = lb;
#if !SF_RO_LABSUM
                            lds_add(&s.lab_sum[lab], to_fix(val, FIX_RES, 1.0e6f));
#endif
                        }
                    }
                }
                s.ro.val[q] = val;
                s.ro.lab[q] = (uint8_t)lab;
                s.ro.flag[q] = 1;
            }
            __syncthreads();
            ro_label_walk(s.ro, min(RO_CHUNK, n - base), tid, la);
            __syncthreads();
        }
        if (tid < SF_NC && commit_ok(cs)) {
#if SF_RO_LABSUM
            const float sum = la.sum;
#else
            const float sum = (float)((double)s.lab_sum[tid] * (1.0 / 4294967296.0));
#endif
            st.cluster_res[tid] = (la.n_val > 0) ? sum / float(2 * (la.n_val + 1)) : __int_as_float(0x7fc00000);
        }
        cluster_barrier(cs, tid);
        return;
    }
#endif

    // residuals, cluster-wise (:1036-1068): per-lane running sums per label, flushed to the workgroup bins
    // (integer LDS atomics) when the label changes; SF_LOAD_BATCH pixels per trip with all loads issued first
    const float kph = a.p.k_photometric_res;
    int cur_lab = 0, cur_cnt = 0;
    long long cur_sum = 0;
    int px_begin, px_end;
    cluster_range(cs, n, 1, px_begin, px_end);  // this workgroup's share of the level
    for (int base = px_begin + tid; base < px_end; base += SF_NT * SF_LOAD_BATCH) {
        long long sd[SF_LOAD_BATCH], si[SF_LOAD_BATCH];
        float dc[SF_LOAD_BATCH], db[SF_LOAD_BATCH], ic[SF_LOAD_BATCH];
        int lb[SF_LOAD_BATCH];
#pragma unroll
        for (int k = 0; k < SF_LOAD_BATCH; k++) {
            const int idx = min(base + k * SF_NT, px_end - 1);
            sd[k] = gld_agent_i64(acc_d, idx);
            si[k] = gld_agent_i64(acc_i, idx);
            dc[k] = gld(dcur, idx);
            db[k] = gld(dbuf, idx);
            ic[k] = gld(icur, idx);
            lb[k] = level0_label(a, labels0, idx);
        }
        if (push) {
#pragma unroll
            for (int k = 0; k < SF_LOAD_BATCH; k++) {
                const int idx = base + k * SF_NT;
                if (idx < px_end) {
                    gst(dpush, idx, dc[k]);
                    gst(ipush, idx, ic[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < SF_LOAD_BATCH; k++) {
            if (!(base + k * SF_NT < px_end && si[k] != 0 && dc[k] != 0.f)) continue;
            float dw, iw;
            if (ordered)
                ro_unpack_cell(sd[k], dw, iw);
            else
                normalise_acc(sd[k], si[k], dw, iw);
            if (dw == 0.f || lb[k] >= SF_NC) continue;
            // intensity_diff is intensityCurrent where both depths are valid, else 0 (:937,1022)
            const float idiff = (db[k] != 0.f) ? ic[k] : 0.f;
            const float cumulative = fabsf(dc[k] - dw) + kph * fabsf(idiff - iw);
            if (lb[k] != cur_lab) {
                if (cur_cnt) {
                    lds_add(&s.lab_sum[cur_lab], cur_sum);
                    lds_add(&s.lab_cnt[cur_lab], cur_cnt);
                }
                cur_lab = lb[k];
                cur_sum = 0;
                cur_cnt = 0;
            }
            cur_sum += to_fix(cumulative, FIX_RES, 1.0e6f);
            cur_cnt++;
        }
    }
    if (cur_cnt) {
        lds_add(&s.lab_sum[cur_lab], cur_sum);
        lds_add(&s.lab_cnt[cur_lab], cur_cnt);
    }
    __syncthreads();
    // per-label sums (exact integers) and counts over the workgroups of the cluster
    if (tid < SF_NC) {
        put_i64(&cs.in[2 * tid], s.lab_sum[tid]);
        cs.in[2 * SF_NC + tid] = (unsigned)s.lab_cnt[tid];
    }
    cluster_gather(cs, 3 * SF_NC, tid);
    if (tid < SF_NC && cl_writer(cs) && commit_ok(cs)) {
        long long t = 0;
        int c = 0;
        for (int p = 0; p < G; p++) {
            t += get_i64(&cs.all[p * 3 * SF_NC + 2 * tid]);
            c += (int)cs.all[p * 3 * SF_NC + 2 * SF_NC + tid];
        }
        const float sum = (float)((double)t * (1.0 / 4294967296.0));
        st.cluster_res[tid] = (c > 0) ? sum / float(2 * (c + 1)) : __int_as_float(0x7fc00000);
    }
    cluster_barrier(cs, tid);  // perClusterAverageResidual is visible to buildSegmImage in every workgroup
}

__device__ __noinline__ void stage_segm_image(const KArgs &a, int b, int tid, LDS ClusterShared &cs) {
    const StreamState &st = a.state[b];
    const int n = a.ln[0];
    const int G = cl_G(cs), rank = cl_rank(cs);
    const auto labels0 = as_global((const uint8_t *)a.labels + (size_t)b * a.n_tot);
    const auto out = as_global(a.b_img + (size_t)b * a.n0);
    // The value of a pixel depends on its label only: lane l of every wave works out label l's once (lane SF_NC: "assume static
    // for invalid cluster"), a pixel then fetches its label's from that lane over the LDS crossbar -- not, as before, b_segm and
    // cluster_res from the stream's state in global memory behind the load of its label, twice per pixel.
    const int lane = tid & 63;
    float mine = 1.f;
    if (lane < SF_NC) {
        mine = std_max(0.f, std_min(1.f, st.b_segm[lane]));
        if ((double)st.cluster_res[lane] < 0.017) mine = std_max(mine, 1.0f - mine);
    }
    // The trip count is the WAVE's (bounded on its first lane): ds_bpermute returns 0 for a source lane that EXEC has switched
    // off, so no lane may leave before the wave's last pixel is served -- a last trip of r pixels with r % 64 in 1 .. SF_NC would
    // otherwise read 0.0 for every label >= r % 64 (48 x 43, 36 x 116: n0 % 64 = 16).
    for (int base = tid + rank * SF_NT * SF_LOAD_BATCH; base - lane < n; base += SF_NT * SF_LOAD_BATCH * G) {
        int lab[SF_LOAD_BATCH];
#pragma unroll
        for (int k = 0; k < SF_LOAD_BATCH; k++) lab[k] = level0_label(a, labels0, min(base + k * SF_NT, n - 1));
#pragma unroll
        for (int k = 0; k < SF_LOAD_BATCH; k++) {
            const int idx = base + k * SF_NT;
            // (every lane of the wave takes part in the exchange -- see the loop bound; labels are 0 .. SF_NC)
            const float bb = __int_as_float(__builtin_amdgcn_ds_bpermute(lab[k] << 2, __float_as_int(mine)));
            if (idx < n) gst(out, idx, bb);
        }
    }
}

// copy_images = false: the residual stage of this launch has stored the images already, only the pose is pushed
__device__ __noinline__ void stage_push_history(const KArgs &a, int b, int im_count, bool copy_images, int tid, LDS ClusterShared &cs) {
    const int G = cl_G(cs), rank = cl_rank(cs);
    StreamState &st = a.state[b];
    const int slot = im_count % SF_HISTORY, n = copy_images ? a.ln[0] : 0;
    const auto dcur = as_global(pyr_level(a, b, 0, 0, 0)), icur = as_global(pyr_level(a, b, 0, 1, 0));
    const auto dbuf = as_global(a.hist_d + ((size_t)slot * a.batch + b) * a.n0);
    const auto ibuf = as_global(a.hist_i + ((size_t)slot * a.batch + b) * a.n0);
    for (int base = tid * 4 + rank * SF_NT * 8; base < n; base += SF_NT * 4 * 2 * G) {  // 16-byte copies, two per trip
        typedef float __attribute__((ext_vector_type(4))) f4;
        typedef __attribute__((address_space(1))) const f4 gcf4;
        typedef __attribute__((address_space(1))) f4 gf4;
        const int i0 = base, i1 = base + SF_NT * 4;
        const bool in1 = i1 < n;
        typedef __attribute__((address_space(1))) const char gcc;
        typedef __attribute__((address_space(1))) char gc;
        const unsigned o0 = (unsigned)i0 * 4u, o1 = (unsigned)i1 * 4u;  // SGPR base + 32-bit byte offset (gld, sf_device_common.h)
        const f4 d0 = *(gcf4 *)((gcc *)dcur + o0), c0 = *(gcf4 *)((gcc *)icur + o0);
        f4 d1 = d0, c1 = c0;
        if (in1) {
            d1 = *(gcf4 *)((gcc *)dcur + o1);
            c1 = *(gcf4 *)((gcc *)icur + o1);
        }
        *(gf4 *)((gc *)dbuf + o0) = d0;
        *(gf4 *)((gc *)ibuf + o0) = c0;
        if (in1) {
            *(gf4 *)((gc *)dbuf + o1) = d1;
            *(gf4 *)((gc *)ibuf + o1) = c1;
        }
    }
    if (tid < 16 && cl_writer(cs) && commit_ok(cs)) st.hist_T[slot][tid] = st.T[tid];
}
