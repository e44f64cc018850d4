// sf_residuals.h — the stages that follow runSolver in the drivers' frame loop:
//   computeResidualsAgainstPreviousImage  (reference FrontEnd.cpp:896-1069)
//   buildSegmImage                        (reference SegmentationBackground.cpp:176-197)
//   ring-buffer push                      (reference StaticFusion-datasets.cpp:182-184)
#pragma once

#include "sf_device_common.h"
#include "sf_smallmath.h"

struct ResShared {
    float Tinv[16];
    long long lab_sum[SF_NC];
    int lab_cnt[SF_NC];
    double dwork[32];
    SplatWin win;
};

__device__ __forceinline__ int level0_label(const KArgs &a, gptr<const uint8_t> labels0, int idx) {
    // without segmentation the reference's clusterAllocation[0] stays at its constructor value 0
    return a.p.segmentation_enabled ? (int)labels0[idx] : 0;
}

__device__ __noinline__ void stage_residuals(const KArgs &a, int b, int index, LDS ResShared &s, int tid) {
    const int lane = tid & 63;
    StreamState &st = a.state[b];
    const int rows = a.lrows[0], cols = a.lcols[0], n = a.ln[0];
    const size_t sb = (size_t)b * a.n_tot, rb = (size_t)b * a.n0;
    const auto dcur = as_global((const float *)a.pyr_new[0] + sb), icur = as_global((const float *)a.pyr_new[1] + sb);  // depthCurrent / intensityCurrent
    const auto labels0 = as_global((const uint8_t *)a.labels + sb);
    const int idx_to_warp = (index - SF_HISTORY) % SF_HISTORY;
    const auto dbuf = as_global((const float *)a.hist_d + ((size_t)idx_to_warp * a.batch + b) * a.n0);
    const auto ibuf = as_global((const float *)a.hist_i + ((size_t)idx_to_warp * a.batch + b) * a.n0);
    const auto acc_d = as_global(a.acc_d + rb), acc_i = as_global(a.acc_i + rb);
    const auto acc_w = as_global(a.acc_w + rb);

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
    for (int idx = tid; idx < n; idx += SF_NT) {
        acc_d[idx] = 0;
        acc_i[idx] = 0;
        acc_w[idx] = 0;
    }
    __syncthreads();

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
            z = dbuf[idx];
            iw = ibuf[idx];
            const float dc = dcur[idx];
            xr = (inv_f_i * (float(u) - disp_u_i)) * z;  // xxBuffer / yyBuffer (:922-926)
            yr = (inv_f_i * (float(v) - disp_v_i)) * z;
            return z != 0.f && dc != 0.f;
        }
    } src{dbuf, ibuf, dcur, inv_f_i, g.disp_u_i, g.disp_v_i};
    tiled_splat(g, rows, cols, src, acc_d, acc_i, acc_w, s.win, tid);
    __syncthreads();

    // residuals, cluster-wise (:1036-1068)
    const float kph = a.p.k_photometric_res;
    for (int base = 0; base < n; base += SF_NT) {
        const int idx = base + tid;
        bool ok = false;
        int lab = 0;
        long long fx = 0;
        if (idx < n) {
            const uint32_t w = __hip_atomic_load(acc_w + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float dc = dcur[idx];
            if (w != 0 && dc != 0.f) {
                const long long sd = __hip_atomic_load(acc_d + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const long long si = __hip_atomic_load(acc_i + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float dw, iw;
                normalise_acc(sd, si, w, dw, iw);
                if (dw != 0.f) {
                    // intensity_diff is intensityCurrent where both depths are valid, else 0 (:937,1022)
                    const float idiff = (dbuf[idx] != 0.f) ? icur[idx] : 0.f;
                    const float cumulative = fabsf(dc - dw) + kph * fabsf(idiff - iw);
                    ok = true;
                    lab = level0_label(a, labels0, idx);
                    fx = to_fix(cumulative, FIX_RES, 1.0e6f);
                }
            }
        }
        ok = ok && lab < SF_NC;
        wave_label_add_i64(ok, lab, fx, s.lab_sum, lane);
        wave_label_count(ok, lab, s.lab_cnt, lane);
    }
    __syncthreads();
    if (tid < SF_NC) {
        const int c = s.lab_cnt[tid];
        const float sum = (float)((double)s.lab_sum[tid] * (1.0 / 4294967296.0));
        st.cluster_res[tid] = (c > 0) ? sum / float(2 * (c + 1)) : __int_as_float(0x7fc00000);
    }
    __syncthreads();
}

__device__ __noinline__ void stage_segm_image(const KArgs &a, int b, int tid) {
    const StreamState &st = a.state[b];
    const int n = a.ln[0];
    const auto labels0 = as_global((const uint8_t *)a.labels + (size_t)b * a.n_tot);
    const auto out = as_global(a.b_img + (size_t)b * a.n0);
    for (int idx = tid; idx < n; idx += SF_NT) {
        const int lab = level0_label(a, labels0, idx);
        float bb = 1.f;  // "assume static for invalid cluster"
        if (lab != SF_NC) {
            bb = std_max(0.f, std_min(1.f, st.b_segm[lab]));
            if ((double)st.cluster_res[lab] < 0.017) bb = std_max(bb, 1.0f - bb);
        }
        out[idx] = bb;
    }
}

__device__ __noinline__ void stage_push_history(const KArgs &a, int b, int im_count, int tid) {
    StreamState &st = a.state[b];
    const int slot = im_count % SF_HISTORY, n = a.ln[0];
    const auto dcur = as_global((const float *)a.pyr_new[0] + (size_t)b * a.n_tot), icur = as_global((const float *)a.pyr_new[1] + (size_t)b * a.n_tot);
    const auto dbuf = as_global(a.hist_d + ((size_t)slot * a.batch + b) * a.n0);
    const auto ibuf = as_global(a.hist_i + ((size_t)slot * a.batch + b) * a.n0);
    for (int idx = tid; idx < n; idx += SF_NT) {
        dbuf[idx] = dcur[idx];
        ibuf[idx] = icur[idx];
    }
    if (tid < 16) st.hist_T[slot][tid] = st.T[tid];
}
