// sf_reforder_solver.h — the solver stages of the REFERENCE-ORDER build (sf_reforder.h has the what and why).
// Included by sf_solver.h behind SolveShared / IrlsCtx; replaces solve_seg_prior, the initial mean |res| and the two IRLS
// passes. One pixel per lane and trip, plain indexed loads: nothing here is tuned.
#pragma once

#if SF_REFORDER

// ---------------------------------------------------------------------------------------------
//  one pixel of validPixels: its two Jacobian rows (reference FrontEnd.cpp:544-585)
// ---------------------------------------------------------------------------------------------
struct RoPixel {
#if SF_RO_ROWS
    float ac[6], bc, ad[6], bd;  // A(cont, 0..5), B(cont) of the colour row and of the geometry row
#else
    PixFact<float> p;            // the product's factored form
#endif
};

struct RoRec {
    float dn, dw, dcu, dcv, dct, ddu, ddv;
    int lab;  // cluster of a validPixels entry, SF_INVALID_LABEL otherwise
};
struct RoPlanes {
    gptr<const float> p[R_COUNT], dnew;
    gptr<const uint8_t> lab;
};
__device__ __forceinline__ RoPlanes ro_planes(const KArgs &a, int b, int L, const LDS SolveShared &s) {
    RoPlanes r;
    const size_t rb = (size_t)uniform_i(s.rec_slot) * a.n0;
#pragma unroll
    for (int q = 0; q < R_COUNT; q++) r.p[q] = as_global((const float *)a.rec[q] + rb);
    r.dnew = as_global(pyr_level(a, b, 0, 0, L));
    r.lab = as_global((const uint8_t *)a.rec_lab + rb);
    return r;
}
__device__ __forceinline__ void ro_load(const RoPlanes &pl, int idx, RoRec &r) {
    r.dn = gld(pl.dnew, idx);
    r.dw = gld(pl.p[R_DW], idx);
    r.dcu = gld(pl.p[R_DCU], idx);
    r.dcv = gld(pl.p[R_DCV], idx);
    r.dct = gld(pl.p[R_DCT], idx);
    r.ddu = gld(pl.p[R_DDU], idx);
    r.ddv = gld(pl.p[R_DDV], idx);
    r.lab = (int)gld(pl.lab, idx);
#if !SF_RO_BEHIND
    r.dw = fabsf(r.dw);  // the product's records carry validPixels in the sign
#endif
}

__device__ __forceinline__ void ro_pixel(const LevelGeom &g, int idx, const RoRec &r, RoPixel &o) {
    float fu, fv;
    split_index(g, idx, fu, fv);
#if SF_RO_ROWS
    // Inter coordinates (calculateCoord, :402-404) from the pyramid's and the warp's xx / yy (:385-386, :874-880; on the
    // Warped := Pred iteration the warped coordinates are the prediction pyramid's, :1107-1108)
    const float xn = (g.inv_f_pyr * (fu - g.disp_u_i)) * r.dn, yn = (g.inv_f_pyr * (fv - g.disp_v_i)) * r.dn;
    float xw, yw;
    if (g.first) {
        xw = (g.inv_f_pyr * (fu - g.disp_u_i)) * r.dw;
        yw = (g.inv_f_pyr * (fv - g.disp_v_i)) * r.dw;
    } else {
        xw = (fu - g.disp_u_i) * r.dw * g.inv_f_w;
        yw = (fv - g.disp_v_i) * r.dw * g.inv_f_w;
    }
    const float d = 0.5f * (r.dn + r.dw), x = 0.5f * (xn + xw), y = 0.5f * (yn + yw);
    const float ddt_ = r.dn - r.dw;
    // computeWeights (:487-509): sqrtf(1 / (error_m + error_l)), then the whole plane times 1 / maximum
    const float error_l_c = 10.f * (fabsf(r.dct) + fabsf(r.dcu) + fabsf(r.dcv));
    const float error_l_d = 200.f * (fabsf(ddt_) + fabsf(r.ddu) + fabsf(r.ddv));
    const float weight_c = g.inv_max_c * vrsq(1.f + error_l_c);
    const float weight_d = g.inv_max_d * vrsq(0.01f + error_l_d);
    const float inv_d = vrcpw(d);
    {   // colour row (:552-566)
        const float dycomp = r.dcu * g.f_inv * inv_d, dzcomp = r.dcv * g.f_inv * inv_d;
        const float tw = weight_c * g.kph;
        o.ac[0] = tw * (-dycomp);
        o.ac[1] = tw * (-dzcomp);
        o.ac[2] = tw * (dycomp * x * inv_d + dzcomp * y * inv_d);
        o.ac[3] = tw * (dycomp * inv_d * y * x + dzcomp * (y * y * inv_d + d));
        o.ac[4] = tw * (-dycomp * (x * x * inv_d + d) - dzcomp * inv_d * y * x);
        o.ac[5] = tw * (dycomp * y - dzcomp * x);
        o.bc = tw * (-r.dct);
    }
    {   // geometry row (:570-585)
        const float dycomp = r.ddu * g.f_inv * inv_d, dzcomp = r.ddv * g.f_inv * inv_d;
        const float tw = weight_d;
        o.ad[0] = tw * (-dycomp);
        o.ad[1] = tw * (-dzcomp);
        o.ad[2] = tw * (1.f + dycomp * x * inv_d + dzcomp * y * inv_d);
        o.ad[3] = tw * (y + dycomp * inv_d * y * x + dzcomp * (y * y * inv_d + d));
        o.ad[4] = tw * (-x - dycomp * (x * x * inv_d + d) - dzcomp * inv_d * y * x);
        o.ad[5] = tw * (dycomp * y - dzcomp * x);
        o.bd = tw * (-ddt_);
    }
#else
    fact_from_record<float>(g, fu, fv, r.dn, r.dw, r.dcu, r.dcv, r.dct, r.ddu, r.ddv, o.p);
#endif
}

// res = -B; res += Var(k) * A.col(k), k = 0..5 (:644-646), for both rows of the pixel
__device__ __forceinline__ void ro_residuals(const RoPixel &o, const float (&V)[6], float &res_c, float &res_d) {
#if SF_RO_ROWS
    res_c = -o.bc;
    res_d = -o.bd;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        res_c += V[k] * o.ac[k];
        res_d += V[k] * o.ad[k];
    }
#else
    fact_residuals<float>(o.p, V, res_c, res_d);
#endif
}
__device__ __forceinline__ void ro_abs_b(const RoPixel &o, float &abs_c, float &abs_d) {
#if SF_RO_ROWS
    abs_c = fabsf(-o.bc);
    abs_d = fabsf(-o.bd);
#else
    abs_c = fabsf(o.p.bct);
    abs_d = fabsf(o.p.bdt);
#endif
}
// Aw = res_weight * A, Bw = res_weight * B of both rows (:627-636): aw[0..5] the row, aw[6] its right-hand side
__device__ __forceinline__ void ro_weighted_rows(const RoPixel &o, float w_c, float w_d, float (&awc)[7], float (&awd)[7]) {
#if SF_RO_ROWS
#pragma unroll
    for (int k = 0; k < 6; k++) {
        awc[k] = w_c * o.ac[k];
        awd[k] = w_d * o.ad[k];
    }
    awc[6] = w_c * o.bc;
    awd[6] = w_d * o.bd;
#else
    const PixFact<float> &p = o.p;
    {
        const float P = w_c * p.pc, Q = w_c * p.qc;
        awc[0] = -P;
        awc[1] = -Q;
        awc[2] = vfma(P, p.xd, Q * p.yd);
        awc[3] = vfma(P, p.xyd, Q * p.yyd);
        awc[4] = -vfma(P, p.xxd, Q * p.xyd);
        awc[5] = vfma(P, p.y, -(Q * p.x));
        awc[6] = -(w_c * p.bct);
    }
    {
        const float W = w_d * p.twd, Pd = w_d * p.pd, Qd = w_d * p.qd;
        awd[0] = -Pd;
        awd[1] = -Qd;
        awd[2] = vfma(Pd, p.xd, vfma(Qd, p.yd, W));
        awd[3] = vfma(Pd, p.xyd, vfma(Qd, p.yyd, W * p.y));
        awd[4] = -vfma(Pd, p.xxd, vfma(Qd, p.xyd, W * p.x));
        awd[5] = vfma(Pd, p.y, -(Qd * p.x));
        awd[6] = -(w_d * p.bdt);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
//  computeSegPrior (reference SegmentationBackground.cpp:53-103): `b_prior[l] += 1 - kz |ddt|` over the level, u outer /
//  v inner, as a sequential float sum per cluster
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ void ro_seg_prior(const KArgs &a, int b, int L, LDS SolveShared &s, LDS ClusterShared &cs, int tid) {
    const int n = a.ln[L];
    const float kz = a.p.kz;
    const size_t sb = (size_t)b * a.n_tot + a.loff[L], rb = (size_t)cl_slot(cs) * a.n0;
    const auto dnew = as_global(pyr_level(a, b, 0, 0, L));
    const auto dwp = as_global((const float *)a.rec[R_DW] + rb);
    const auto labp = as_global((const uint8_t *)a.labels + sb);
    const auto vlab = as_global((const uint8_t *)a.rec_lab + rb);
    RoLabelAcc la{0.f, 0, 0, 0};
#if !SF_RO_LABSUM
    if (tid < SF_NC) s.prior_sum[tid] = 0;
#endif
    __syncthreads();
    for (int base = 0; base < n; base += RO_CHUNK) {
        for (int q = tid; q < RO_CHUNK; q += SF_NT) {
            const int idx = base + q;
            int lab = SF_INVALID_LABEL, flag = 0;
            float val = 0.f;
            if (idx < n) {
                const int l = (int)gld(labp, idx);
                if (l != SF_NC) {  // labels_ref(v, u) != NUM_CLUSTERS
                    lab = l;
                    const float dn = gld(dnew, idx);
                    float dw = gld(dwp, idx);
#if !SF_RO_BEHIND
                    dw = fabsf(dw);
#endif
                    if (dn != 0.f && dw != 0.f) {  // Null(v, u) == 0
                        flag |= 1;
                        val = 1.f - kz * fabsf(dn - dw);
#if !SF_RO_LABSUM
                        lds_add(&s.prior_sum[l], to_fix(val, FIX_RES, 1.0e6f));
#endif
                    }
                    if ((int)gld(vlab, idx) != SF_INVALID_LABEL) flag |= 2;
                }
            }
            s.ro.val[q] = val;
            s.ro.lab[q] = (uint8_t)lab;
            s.ro.flag[q] = (uint8_t)flag;
        }
        __syncthreads();
        ro_label_walk(s.ro, min(RO_CHUNK, n - base), tid, la);
        __syncthreads();
    }
    if (tid < SF_NC) {  // reference SegmentationBackground.cpp:84-102
        const int l = tid;
        s.valid_cnt[l] = la.n_valid;
        float bp = 0.f, lt = 0.f;
        if (la.n_all != 0) {
            const float ratio = float(la.n_val) / float(la.n_all);
            if (ratio < 0.1f) {
                lt = 0.1f;
                bp = -1.f;
            } else {
                lt = ratio;
#if SF_RO_LABSUM
                const float sum = la.sum;
#else
                const float sum = (float)((double)s.prior_sum[l] * (1.0 / 4294967296.0));
#endif
                bp = std_max(-1.f, std_min(2.f, sum / la.n_val));
            }
        }
        s.b_prior[l] = bp;
        s.lambda_t_w[l] = lt;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
//  aver_res = res.cwiseAbs().sumAll() / res.size() with res = -B (:588-590): [C1] fp64 sum of the float |B|
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ void ro_initial_residual(const KArgs &a, int b, int L, LDS SolveShared &s, int tid) {
    const IrlsCtx c = make_irls_ctx(a, b, L, s);
    const RoPlanes pl = ro_planes(a, b, L, s);
    const int lane = tid & 63, wave = tid >> 6;
    double t = 0.0;
#if SF_RO_SEQ64
    for (int base = c.begin; base < c.n; base += RO_CHUNK) {
        for (int q = tid; q < RO_CHUNK; q += SF_NT) {
            const int idx = base + q;
            int lab = SF_INVALID_LABEL;
            if (idx < c.n) {
                RoRec r;
                ro_load(pl, idx, r);
                if (r.lab != SF_INVALID_LABEL) {
                    RoPixel px;
                    ro_pixel(c.g, idx, r, px);
                    float ac, ad;
                    ro_abs_b(px, ac, ad);
                    s.ro2.rc[q] = ac;
                    s.ro2.rd[q] = ad;
                    lab = r.lab;
                }
            }
            s.ro2.c.lab[q] = (uint8_t)lab;
        }
        __syncthreads();
        if (tid == 0) {  // res.cwiseAbs().sumAll(): row after row ([C1])
            const int m = min(RO_CHUNK, c.n - base);
            for (int q = 0; q < m; q++) {
                if (s.ro2.c.lab[q] == SF_INVALID_LABEL) continue;
                t += (double)s.ro2.rc[q];
                t += (double)s.ro2.rd[q];
            }
        }
        __syncthreads();
    }
#else
    for (int idx = c.begin + tid; idx < c.n; idx += SF_NT) {
        RoRec r;
        ro_load(pl, idx, r);
        if (r.lab == SF_INVALID_LABEL) continue;
        RoPixel px;
        ro_pixel(c.g, idx, r, px);
        float ac, ad;
        ro_abs_b(px, ac, ad);
        t += (double)ac;
        t += (double)ad;
    }
#endif
    t = wave_sum_f64(t);  // (row-by-row sums: lane 0 of wave 0 holds the sum, every other lane 0.0)
    if (lane == 0) s.red[wave][0] = t;
    __syncthreads();
    if (tid == 0) {
        double q = 0.0;
        for (int w = 0; w < SF_NW; w++) q += s.red[w][0];
        s.aver_res = (float)q / float(2 * c.N);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
//  pass 1: Cauchy x b weights, AtA / AtB (:615-641) -> s.red[wave][0..26]
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ void ro_pass1(const KArgs &a, int b, int L, LDS SolveShared &s, int tid) {
    const IrlsCtx c = make_irls_ctx(a, b, L, s);
    const RoPlanes pl = ro_planes(a, b, L, s);
    const int lane = tid & 63, wave = tid >> 6;
    const float inv_c_Cauchy = 1.f / (a.p.kc_Cauchy * uniform_f(s.aver_res));
    float Vr[6];
#pragma unroll
    for (int q = 0; q < 6; q++) Vr[q] = uniform_f(s.Var[q]);
#if SF_RO_SEQ64 && SF_RO_P1_FP64
    // AtA / AtB row by row ([C1]): the weighted rows of a chunk of pixels go to LDS, lane q < 27 owns sum q and adds the products
    // of the colour row, then of the depth row, pixel after pixel
    int ei = 0, ej = 6;  // the two row entries whose product this lane sums (21 + i: entry i times Bw)
    if (tid < 21) {
        int q = 0;
        for (int i = 0; i < 6; i++)
            for (int j = i; j < 6; j++, q++)
                if (q == tid) { ei = i; ej = j; }
    } else if (tid < 27) {
        ei = tid - 21;
    }
    double sum = 0.0;
    for (int base = c.begin; base < c.n; base += RO_ROWS_CHUNK) {
        for (int q = tid; q < RO_ROWS_CHUNK; q += SF_NT) {
            const int idx = base + q;
            bool ok = false;
            if (idx < c.n) {
                RoRec r;
                ro_load(pl, idx, r);
                if (r.lab != SF_INVALID_LABEL) {
                    RoPixel px;
                    ro_pixel(c.g, idx, r, px);
                    float res_c, res_d;
                    ro_residuals(px, Vr, res_c, res_d);  // the residuals of the previous iteration's solution (-B for the first)
                    const float b_weight = std_max(0.f, std_min(1.f, s.b_segm[r.lab]));
                    const float w_c = b_weight * vrsq(1.f + sqf(res_c * inv_c_Cauchy));
                    const float w_d = b_weight * vrsq(1.f + sqf(res_d * inv_c_Cauchy));
                    float awc[7], awd[7];
                    ro_weighted_rows(px, w_c, w_d, awc, awd);
#pragma unroll
                    for (int k = 0; k < 7; k++) {
                        s.rows.aw[k][q] = awc[k];
                        s.rows.aw[7 + k][q] = awd[k];
                    }
                    ok = true;
                }
            }
            s.rows.ok[q] = ok;
        }
        __syncthreads();
        if (tid < 27) {
            const int m = min(RO_ROWS_CHUNK, c.n - base);
            for (int q = 0; q < m; q++) {
                if (!s.rows.ok[q]) continue;
                sum += (double)s.rows.aw[ei][q] * (double)s.rows.aw[ej][q];
                sum += (double)s.rows.aw[7 + ei][q] * (double)s.rows.aw[7 + ej][q];
            }
        }
        __syncthreads();
    }
    if (lane < 27) s.red[wave][lane] = wave == 0 ? sum : 0.0;
}

#else
    double acc[27];
#pragma unroll
    for (int q = 0; q < 27; q++) acc[q] = 0.0;
#if !SF_RO_P1_FP64
    float acc32[27];
#pragma unroll
    for (int q = 0; q < 27; q++) acc32[q] = 0.f;
    int since = 0;
#endif
    for (int idx = c.begin + tid; idx < c.n; idx += SF_NT) {
        RoRec r;
        ro_load(pl, idx, r);
        if (r.lab == SF_INVALID_LABEL) continue;
        RoPixel px;
        ro_pixel(c.g, idx, r, px);
        float res_c, res_d;
        ro_residuals(px, Vr, res_c, res_d);  // the residuals of the previous iteration's solution (-B for the first)
        const float b_weight = std_max(0.f, std_min(1.f, s.b_segm[r.lab]));
        const float w_c = b_weight * vrsq(1.f + sqf(res_c * inv_c_Cauchy));
        const float w_d = b_weight * vrsq(1.f + sqf(res_d * inv_c_Cauchy));
        float awc[7], awd[7];
        ro_weighted_rows(px, w_c, w_d, awc, awd);
#if SF_RO_P1_FP64
#pragma unroll
        for (int row = 0; row < 2; row++) {
            const float(&aw)[7] = row ? awd : awc;
            int q = 0;
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = i; j < 6; j++) acc[q++] += (double)aw[i] * (double)aw[j];
#pragma unroll
            for (int i = 0; i < 6; i++) acc[21 + i] += (double)aw[i] * (double)aw[6];
        }
#else
        accum_row(acc32, awc);
        accum_row(acc32, awd);
        if (++since == 2 * SF_P1_FLUSH) {  // a lane's fp32 sums hold as many terms as the product's (pixel pairs there)
            since = 0;
#pragma unroll
            for (int q = 0; q < 27; q++) {
                acc[q] += (double)acc32[q];
                acc32[q] = 0.f;
            }
        }
#endif
    }
#if !SF_RO_P1_FP64
#pragma unroll
    for (int q = 0; q < 27; q++) acc[q] += (double)acc32[q];
#endif
#pragma unroll
    for (int q = 0; q < 27; q++) {
        const double t = wave_sum_f64(acc[q]);
        if (lane == 0) s.red[wave][q] = t;
    }
}
#endif  // SF_RO_SEQ64 && SF_RO_P1_FP64

// ---------------------------------------------------------------------------------------------
//  pass 2: residuals with the new solution, per-cluster sums of |res_c| + |res_d| in validPixels order, ||res||^2
//  (:644-667, :689)
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ void ro_pass2(const KArgs &a, int b, int L, LDS SolveShared &s, int tid) {
    const IrlsCtx c = make_irls_ctx(a, b, L, s);
    const RoPlanes pl = ro_planes(a, b, L, s);
    const int lane = tid & 63, wave = tid >> 6;
    float Vr[6];
#pragma unroll
    for (int q = 0; q < 6; q++) Vr[q] = uniform_f(s.Var[q]);
    double sq = 0.0;
    RoLabelAcc la{0.f, 0, 0, 0};
    (void)la;
    for (int base = c.begin; base < c.n; base += RO_CHUNK) {
        for (int q = tid; q < RO_CHUNK; q += SF_NT) {
            const int idx = base + q;
            int lab = SF_INVALID_LABEL;
            float val = 0.f;
            if (idx < c.n) {
                RoRec r;
                ro_load(pl, idx, r);
                if (r.lab != SF_INVALID_LABEL) {
                    RoPixel px;
                    ro_pixel(c.g, idx, r, px);
                    float res_c, res_d;
                    ro_residuals(px, Vr, res_c, res_d);
#if SF_RO_SEQ64
                    s.ro2.rc[q] = res_c;
                    s.ro2.rd[q] = res_d;
#else
                    sq += (double)res_c * (double)res_c;
                    sq += (double)res_d * (double)res_d;
#endif
                    val = fabsf(res_c) + fabsf(res_d);
                    lab = r.lab;
#if !SF_RO_LABSUM
                    lds_add(&s.lab_sum[lab], (long long)to_fix32_pos(val));
#endif
                }
            }
            s.ro.val[q] = val;
            s.ro.lab[q] = (uint8_t)lab;
            s.ro.flag[q] = 1;
        }
        __syncthreads();
#if SF_RO_LABSUM
        ro_label_walk(s.ro, min(RO_CHUNK, c.n - base), tid, la);
#endif
#if SF_RO_SEQ64
        if (tid == SF_NC) {  // res.squaredNorm(): row after row ([C1]), on the lane next to the 24 of the per-cluster sums
            const int m = min(RO_CHUNK, c.n - base);
            for (int q = 0; q < m; q++) {
                if (s.ro.lab[q] == SF_INVALID_LABEL) continue;
                sq += (double)s.ro2.rc[q] * (double)s.ro2.rc[q];
                sq += (double)s.ro2.rd[q] * (double)s.ro2.rd[q];
            }
        }
#endif
        __syncthreads();
    }
#if SF_RO_LABSUM
    if (tid < SF_NC) s.aver_res_label[tid] = la.sum;
#endif
    sq = wave_sum_f64(sq);  // (row-by-row: one lane holds the sum, every other lane 0.0)
    if (lane == 0) s.red[wave][27] = sq;
}

#endif  // SF_REFORDER
