// sf_pyramid.h — depth / intensity / xx / yy pyramid of one stream.
// Replaces StaticFusion::createImagePyramid (reference FrontEnd.cpp:256-391).
// One output pixel per thread, column-major (v fastest) so that a wave reads and writes
// consecutive addresses.  Every per-pixel expression keeps the reference's operation order
// (-ffp-contract=off), which makes the planes bit-identical to the CPU path.
#pragma once

#include "sf_cluster.h"
#include "sf_device_common.h"

// convMask(k) = v_mask(i)*v_mask(j)/36.f with k = i + 4j  (reference FrontEnd.cpp:146-149)
__device__ __forceinline__ float conv_mask(int k) {
    const float vi = ((k & 3) == 0 || (k & 3) == 3) ? 1.f : 2.f;
    const float vj = ((k >> 2) == 0 || (k >> 2) == 3) ? 1.f : 2.f;
    return vi * vj / 36.f;
}

// which: bit 0 = the Pred pyramid (createImagePyramid(true)), bit 1 = the new one (createImagePyramid(false)); both bits: the
// two pyramids level by level together, one barrier per level for both (what a cluster pays a rendezvous for)
__device__ __noinline__ void stage_pyramid(const KArgs &a, int b, int which, int tid, LDS ClusterShared &cs) {
    const int G = cl_G(cs), rank = cl_rank(cs);  // a cluster's workgroups take every G-th block of SF_NT pixels of a level
    const float max_depth_dif = 0.1f;

    for (int L = 0; L < a.levels; L++) {
        const int rows_i = a.lrows[L], cols_i = a.lcols[L], n = a.ln[L];
        if (L > 0) cluster_barrier(cs, tid);  // level L-1 complete (written by this workgroup / by the cluster's workgroups)
        for (int si = 0; si < 2; si++) {
            if (L == 0 || !((which >> si) & 1)) continue;
            float *const *set = (si == 0) ? a.pyr_pred : a.pyr_new;
            const auto depth = as_global(set[0] + (size_t)b * a.n_tot);
            const auto inten = as_global(set[1] + (size_t)b * a.n_tot);
            const auto d_here = depth + a.loff[L], i_here = inten + a.loff[L];
            {
            const auto d_prev = depth + a.loff[L - 1], i_prev = inten + a.loff[L - 1];
            const int rows_p = a.lrows[L - 1];
            for (int idx = tid + rank * SF_NT; idx < n; idx += SF_NT * G) {
                const int u = idx / rows_i, v = idx - u * rows_i;
                const int u2 = 2 * u, v2 = 2 * v;
                float dout, iout;
                if ((v > 0) && (v < rows_i - 1) && (u > 0) && (u < cols_i - 1)) {
                    float db[16], ib[16];
#pragma unroll
                    for (int c = 0; c < 4; c++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int src = (v2 - 1 + r) + (u2 - 1 + c) * rows_p;
                            db[r + 4 * c] = d_prev[src];
                            ib[r + 4 * c] = i_prev[src];
                        }
                    float d0 = db[5], d1 = db[6], d2 = db[9], d3 = db[10];
                    if (d1 < d0) { const float t = d1; d1 = d0; d0 = t; }
                    if (d3 < d2) { const float t = d3; d3 = d2; d2 = t; }
                    const float dcenter = (d3 < d1) ? std_max(d3, d0) : std_max(d1, d2);
                    if (dcenter != 0.f) {
                        float sum_d = 0.f, sum_c = 0.f, weight = 0.f;
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const float abs_dif = fabsf(db[k] - dcenter);
                            if (abs_dif < max_depth_dif) {
                                const float aux_w = conv_mask(k) * (max_depth_dif - abs_dif);
                                weight += aux_w;
                                sum_d += aux_w * db[k];
                                sum_c += aux_w * ib[k];
                            }
                        }
                        dout = sum_d / weight;
                        iout = sum_c / weight;
                    } else {
                        float lane4[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float m0 = conv_mask(j) * ib[j];
                            const float m1 = conv_mask(4 + j) * ib[4 + j];
                            const float m2 = conv_mask(8 + j) * ib[8 + j];
                            const float m3 = conv_mask(12 + j) * ib[12 + j];
                            lane4[j] = (m0 + m1) + (m2 + m3);
                        }
                        iout = (lane4[0] + lane4[2]) + (lane4[1] + lane4[3]);
                        dout = 0.f;
                    }
                } else {
                    float db[4], ib[4];
#pragma unroll
                    for (int c = 0; c < 2; c++)
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            const int src = (v2 + r) + (u2 + c) * rows_p;
                            db[r + 2 * c] = d_prev[src];
                            ib[r + 2 * c] = i_prev[src];
                        }
                    iout = 0.25f * ((ib[0] + ib[2]) + (ib[1] + ib[3]));
                    float new_d = 0.f;
                    unsigned cont = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (db[k] != 0.f) {
                            new_d += db[k];
                            cont++;
                        }
                    dout = (cont != 0) ? new_d / float(cont) : 0.f;
                }
                d_here[idx] = dout;
                i_here[idx] = iout;  // xx / yy (:385-386) are recomputed by their consumers: level_coord()
            }
            }
        }  // level 0 is the input itself
    }
    cluster_barrier(cs, tid);
}
