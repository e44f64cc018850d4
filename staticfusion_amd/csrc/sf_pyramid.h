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
    const float gap_max = 0.1f;

    for (int L = 0; L < a.levels; L++) {
        const int rows_i = a.lrows[L], cols_i = a.lcols[L], n = a.ln[L];
        const LevelCoord lcL = level_coord(a, L);
        if (L > 0) cluster_barrier(cs, tid);  // level L-1 complete (written by this workgroup / by the cluster's workgroups)
        for (int si = 0; si < 2; si++) {
            if (L == 0 || !((which >> si) & 1)) continue;
            const auto depth = as_global(pyr_plane(a, b, si == 0 ? 1 : 0, 0));
            const auto inten = as_global(pyr_plane(a, b, si == 0 ? 1 : 0, 1));
            const auto d_here = depth + a.loff[L], i_here = inten + a.loff[L];
            {
            const auto d_prev = as_global(pyr_level(a, b, si == 0 ? 1 : 0, 0, L - 1)), i_prev = as_global(pyr_level(a, b, si == 0 ? 1 : 0, 1, L - 1));
            const int rows_p = a.lrows[L - 1];
            // A wave produces 62 consecutive pixels of the level (column-major: consecutive rows v of a column u); lanes 0 and 63
            // are halo lanes. Every lane loads rows 2v, 2v+1 of the four input columns 2u-1 .. 2u+2 as 8-byte pairs (8 loads
            // for both channels instead of 32 single taps) and gets the rows above / below, 2v-1 and 2v+2, from its neighbour
            // lanes over the DPP network (wave_shr / wave_shl): the 4 x 4 block of the reference, no tap fetched twice by a
            // wave. Where the neighbour lane sits in another column the pixel is a border pixel and does not use those rows.
            const int lane = tid & 63, wave = tid >> 6;
            const int n_chunks = (n + 61) / 62;
            typedef __attribute__((address_space(1))) const vfloat2 gcf2;
            for (int chunk = wave + rank * SF_NW; chunk < n_chunks; chunk += SF_NW * G) {
                const int idx = chunk * 62 + lane - 1;
                const bool produce = lane >= 1 && lane <= 62 && idx < n;
                const int idc = min(max(idx, 0), n - 1);
                int u, v;
                split_uv(lcL, idc, u, v);  // reciprocal multiply + correction instead of an integer division
                const int u2 = 2 * u, v2 = 2 * v;
                float dout, iout;
                float db[16], ib[16];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int col = min(max(u2 - 1 + c, 0), 2 * cols_i - 1);  // clamped: a border pixel does not use what is outside
                    const vfloat2 dd = *(gcf2 *)((__attribute__((address_space(1))) const char *)d_prev + (unsigned)(v2 + col * rows_p) * 4u);
                    const vfloat2 ii = *(gcf2 *)((__attribute__((address_space(1))) const char *)i_prev + (unsigned)(v2 + col * rows_p) * 4u);
                    db[1 + 4 * c] = dd.x;
                    db[2 + 4 * c] = dd.y;
                    ib[1 + 4 * c] = ii.x;
                    ib[2 + 4 * c] = ii.y;
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {  // row 2v-1 = the lane above's row 2(v-1)+1, row 2v+2 = the lane below's row 2(v+1)
                    db[0 + 4 * c] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(db[2 + 4 * c]), 0x138, 0xf, 0xf, false));
                    ib[0 + 4 * c] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ib[2 + 4 * c]), 0x138, 0xf, 0xf, false));
                    db[3 + 4 * c] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(db[1 + 4 * c]), 0x130, 0xf, 0xf, false));
                    ib[3 + 4 * c] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ib[1 + 4 * c]), 0x130, 0xf, 0xf, false));
                }
                if (!produce) continue;
                if ((v > 0) && (v < rows_i - 1) && (u > 0) && (u < cols_i - 1)) {
                    float d0 = db[5], d1 = db[6], d2 = db[9], d3 = db[10];
                    if (d1 < d0) { const float t = d1; d1 = d0; d0 = t; }
                    if (d3 < d2) { const float t = d3; d3 = d2; d2 = t; }
                    const float z_mid = (d3 < d1) ? std_max(d3, d0) : std_max(d1, d2);
                    if (z_mid != 0.f) {
                        float acc_z = 0.f, acc_g = 0.f, w_all = 0.f;  // depth, grey value and weight sums of the 16 taps
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const float gap = fabsf(db[k] - z_mid);
                            if (gap < gap_max) {
                                const float tap_w = conv_mask(k) * (gap_max - gap);
                                w_all += tap_w;
                                acc_z += tap_w * db[k];
                                acc_g += tap_w * ib[k];
                            }
                        }
                        dout = acc_z / w_all;
                        iout = acc_g / w_all;
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
                    // the 2 x 2 block (rows 2v, 2v+1 of columns 2u, 2u+1): part of what the lane holds
                    const float b4d[4] = {db[1 + 4 * 1], db[2 + 4 * 1], db[1 + 4 * 2], db[2 + 4 * 2]};
                    const float b4i[4] = {ib[1 + 4 * 1], ib[2 + 4 * 1], ib[1 + 4 * 2], ib[2 + 4 * 2]};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        db[q] = b4d[q];
                        ib[q] = b4i[q];
                    }
                    iout = 0.25f * ((ib[0] + ib[2]) + (ib[1] + ib[3]));
                    float z_sum = 0.f;
                    unsigned z_cnt = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (db[k] != 0.f) {
                            z_sum += db[k];
                            z_cnt++;
                        }
                    dout = (z_cnt != 0) ? z_sum / float(z_cnt) : 0.f;
                }
                gst(d_here, idx, dout);
                gst(i_here, idx, iout);  // xx / yy (:385-386) are recomputed by their consumers: level_coord()
            }
            }
        }  // level 0 is the input itself
    }
    cluster_barrier(cs, tid);
}
