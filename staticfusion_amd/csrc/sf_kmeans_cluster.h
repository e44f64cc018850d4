// sf_kmeans_cluster.h — the geometric clustering of sf_kmeans.h shared by the G workgroups of a cluster (sf_cluster.h).
// Same arithmetic, same results (labels, centres, connectivity bit for bit: tests/test_gpu_parity.py on the cluster
// build); what changes is who does what:
//
//   assignment     (seed labels, every Lloyd iteration, the level-0 labels) is split by PIXELS: workgroup r labels the
//                  r-th share of the level, four consecutive pixels per thread, one 32-bit label word per thread.
//   ordered sums   KMeans.cpp:215-221 adds the members of a cluster in pixel order in float32: a sum cannot be split.
//                  They are split by CLUSTER instead: workgroup r owns the clusters c with c % G == r, walks the labels of
//                  the level in pixel order (chunks of 4096 pixels, a quad per thread), compacts the members of its
//                  clusters into LDS in that order and continues its 3 x (24 / G) running sums front to back. It walks only
//                  the shares that hold members of its clusters: the rendezvous after the assignment carries a 24-bit mask
//                  per workgroup ("labels that occur in my share"); a compact cluster spans 4-5 of 24 shares.
//   medians        of the seeds (radix select) likewise by seed ownership.
// Label words that another workgroup reads within the launch are written and read with agent-scope (sc1) accesses and
// handed over through a cluster_gather rendezvous after the writing waves have drained: no fence needed
// (MI355X_MICROARCH.md: sc1 stores + sc1 loads on both sides); depth is read-only here. What is left of the labels for the
// stages that follow is made visible by the cluster_barrier after the stage (sf_frame_kernels.hip).
#pragma once

#include "sf_cluster.h"
#include "sf_kmeans.h"

#define KMC_CHUNK (4 * SF_NT)  // pixels per collection round: one quad of consecutive pixels per thread
#define KMC_MAX_OWN 12         // clusters a workgroup can own: 24 / G with G >= 2
#define KMC_R 5                // chunks loaded and ranked together (the whole level 1 at QVGA: 4800 quads = 4.7 x 1024)

struct KmClShared {
    alignas(16) float run[3][KMC_CHUNK];  // (z, x, y) of the members of this workgroup's clusters in the chunk: cluster by cluster, pixel order
    int wcnt[KMC_R][SF_NW][KMC_MAX_OWN];
    unsigned present;  // labels that occur in this workgroup's share of level 1 after an assignment (bit l = label l)
};
struct KmClusterShared {
    KmShared km;
    KmClShared kc;
};

typedef __attribute__((address_space(1))) unsigned gu32w;
typedef __attribute__((address_space(1))) const unsigned char gcu8b;
__device__ __forceinline__ unsigned ld_word_agent(gu32w *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_word_agent(gu32w *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_byte_agent(gcu8b *p) {
    return __hip_atomic_load((__attribute__((address_space(1))) unsigned char *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the label words this workgroup wrote have left; everybody has written theirs
__device__ __forceinline__ void labels_rendezvous(LDS ClusterShared &cs, int tid) { cluster_rendezvous(cs, tid); }
// The same hand-over after a Lloyd assignment, with a payload: which labels occur in this workgroup's share of the level.
// Afterwards cs.all[p] holds the mask of workgroup p's share: the owner of a cluster collects its members from the shares
// that contain any (a compact cluster spans 4-5 of 24 shares) instead of walking the whole level.
__device__ __forceinline__ void labels_rendezvous_with_mask(LDS ClusterShared &cs, LDS unsigned &present, int tid) {
    if (cl_G(cs) > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // present complete
    if (tid == 0) cs.in[0] = present;
    cluster_gather(cs, 1, tid);
}

// One Lloyd iteration's ordered sums of the clusters this workgroup owns (KMeans.cpp:215-221), in pixel order, and the new
// centres of those clusters -> cs.in[4 q + r] (r < 3: coordinate, 3: member count). Its own function: its own registers.
// R = chunks loaded and ranked together; [qlo, qhi) = the quads to walk (the shares that hold members of the owned clusters)
template <int R>
__device__ __noinline__ void kmc_collect_and_sum(LDS KmClShared &kc, LDS ClusterShared &cs, gu32w *lab1w_, const __attribute__((address_space(1))) void *depth1q_,
                                                 const LevelCoord lc1, int qlo, int qhi, int G, int rank, int nown, int tid,
                                                 long long *prof_out) {
    typedef __attribute__((address_space(1))) const vfloat4 gcf4;
    gu32w *lab1w = uniform_ptr(lab1w_);
    gcf4 *depth1q = uniform_ptr((gcf4 *)depth1q_);
    const int lane = tid & 63, wave = tid >> 6;
    G = uniform_i(G);
    rank = uniform_i(rank);
    nown = uniform_i(nown);
    qlo = uniform_i(qlo);
    qhi = uniform_i(qhi);
    const int n_chunks = (qhi - qlo + SF_NT - 1) / SF_NT;
#ifdef SF_KMC_FINE
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tph = clock64();
#define PH(i) do { if (tid == 0) { const long long n_ = clock64(); ph[i] += n_ - tph; tph = n_; } } while (0)
#else
#define PH(i) do {} while (0)
#endif
        // ---- ordered sums of the clusters this workgroup owns (KMeans.cpp:215-221), in pixel order.
        // R chunks (a quad of consecutive pixels per thread and chunk) are loaded and ranked together; the members of the
        // owned clusters are then compacted into LDS and added GROUP by group, a group being as many consecutive chunks as fit
        // the LDS runs (at QVGA the whole level in one go unless this workgroup's clusters hold more than KMC_CHUNK pixels):
        // two barriers and one long front-to-back pass per group instead of per chunk.
        float acc = 0.f;  // thread (q, r) = tid < 3 nown: running sum of coordinate r over the members of cluster rank + q G
        int total = 0;    // ... and the member count (all three threads of a cluster count)
        for (int ch0 = 0; ch0 < n_chunks; ch0 += R) {
            // per chunk only the quad's label word stays in a register; which pixels are members of an owned cluster is
            // re-derived from it where needed and the depth of a member is loaded when it is scattered (registers: the
            // function must not spill)
            unsigned word[R];
#pragma unroll
            for (int c5 = 0; c5 < R; c5++) {
                const int q = qlo + (ch0 + c5) * SF_NT + tid;
                word[c5] = (q < qhi) ? ld_word_agent(lab1w + q) : 0xffffffffu;  // label 255: nobody's
            }
            // own index (0 .. nown-1) of pixel k of a quad, or -1. A label < 24 implies a valid depth (invalid pixels carry 24).
            auto own_index = [&](unsigned w, int k) -> int {
                const unsigned lb = (w >> (8 * k)) & 255u;
                return (lb < SF_NC && (int)(lb % (unsigned)G) == rank) ? (int)(lb / (unsigned)G) : -1;
            };
            const unsigned long long lt = (1ull << lane) - 1ull;
            // the wave's member counts per chunk and owned cluster (lane c < nown holds cluster c's)
#pragma unroll
            for (int c5 = 0; c5 < R; c5++) {
                int cnt_lane = 0;
                for (int c = 0; c < nown; c++) {
                    int tot = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) tot += __popcll(__ballot(own_index(word[c5], k) == c));
                    if (lane == c) cnt_lane = tot;
                }
                if (lane < nown) kc.wcnt[c5][wave][lane] = cnt_lane;
            }
            PH(0);
            __syncthreads();  // also: the previous group's sums have consumed kc.run
            PH(1);
            // the depths of the quads: issued now, in flight during the offset arithmetic, consumed by the scatter
            vfloat4 dz4[R];
#pragma unroll
            for (int c5 = 0; c5 < R; c5++) {
                const int q = qlo + (ch0 + c5) * SF_NT + tid;
                dz4[c5] = depth1q[(q < qhi) ? q : 0];
            }
            // members of cluster c per chunk in earlier waves / in the whole chunk: lane w reads wave w's count, a DPP scan
            // over the first 16 lanes gives every wave's prefix (one LDS read per (chunk, cluster) instead of one per wave);
            // the results go back to the layout "lane c holds cluster c's numbers"
            int before[R], members[R], chunk_all[R];
#pragma unroll
            for (int c5 = 0; c5 < R; c5++) {
                before[c5] = 0;
                members[c5] = 0;
                chunk_all[c5] = 0;
                for (int c = 0; c < nown; c++) {
                    const int cnt = (lane < SF_NW) ? kc.wcnt[c5][lane][c] : 0;
                    int incl = cnt;
                    incl += dpp_i32<0x111, 0xf>(incl);
                    incl += dpp_i32<0x112, 0xf>(incl);
                    incl += dpp_i32<0x114, 0xf>(incl);
                    incl += dpp_i32<0x118, 0xf>(incl);
                    static_assert(SF_NW <= 16, "one DPP row of waves");
                    const int bef = __builtin_amdgcn_readlane(incl - cnt, wave), mem = __builtin_amdgcn_readlane(incl, 15);
                    if (lane == c) {
                        before[c5] = bef;
                        members[c5] = mem;
                    }
                    chunk_all[c5] += mem;
                }
            }
            PH(2);
            // groups of consecutive chunks whose members (all owned clusters together) fit the LDS runs
            int g0 = 0;
            while (g0 < R) {  // uniform: every lane derives the same group bounds from the same counts
                int g1 = g0, fill = 0;
#pragma unroll
                for (int c5 = 0; c5 < R; c5++) {
                    if (c5 >= g0 && c5 == g1 && (fill + chunk_all[c5] <= KMC_CHUNK || g1 == g0)) {
                        fill += chunk_all[c5];
                        g1 = c5 + 1;
                    }
                }
                // within the group: cluster c's run = its members of chunk g0, then of chunk g0 + 1, ...
                int grp_members = 0;
#pragma unroll
                for (int c5 = 0; c5 < R; c5++) grp_members += (c5 >= g0 && c5 < g1) ? members[c5] : 0;
                int incl = grp_members;  // inclusive scan over the owned clusters (nown <= 12: one DPP row)
                incl += dpp_i32<0x111, 0xf>(incl);
                incl += dpp_i32<0x112, 0xf>(incl);
                incl += dpp_i32<0x114, 0xf>(incl);
                incl += dpp_i32<0x118, 0xf>(incl);
                const int run_start = incl - grp_members;
                int earlier = 0;  // members of cluster c in the group's earlier chunks
#pragma unroll 1
                for (int c5 = g0; c5 < g1; c5++) {  // a real loop (one copy of the body); the chunk's values are selected
                    unsigned w5 = word[0];
                    int bef5 = before[0], mem5 = members[0];
                    vfloat4 d5 = dz4[0];
#pragma unroll
                    for (int e = 1; e < R; e++) {
                        w5 = (c5 == e) ? word[e] : w5;
                        bef5 = (c5 == e) ? before[e] : bef5;
                        mem5 = (c5 == e) ? members[e] : mem5;
                        d5 = (c5 == e) ? dz4[e] : d5;
                    }
                    const int my_base = run_start + earlier + bef5;  // lane c: where this wave's members of cluster c go
                    const int q = qlo + (ch0 + c5) * SF_NT + tid;
                    int qk[4], rk[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        qk[k] = own_index(w5, k);
                        rk[k] = 0;
                    }
                    // rank of every member among the wave's members of ITS cluster in pixel order: lower lanes first, then lower k
                    for (int c = 0; c < nown; c++) {
                        int lower = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) lower += __popcll(__ballot(qk[k] == c) & lt);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            rk[k] = (qk[k] == c) ? lower : rk[k];
                            lower += (qk[k] == c) ? 1 : 0;
                        }
                    }
                    if (__any((qk[0] & qk[1] & qk[2] & qk[3]) >= 0)) {  // uniform: this wave holds a member in this chunk
                        int u, v;
                        split_uv(lc1, 4 * q, u, v);  // the quad's first pixel; the others follow down the column
                        const float pz[4] = {d5.x, d5.y, d5.z, d5.w};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int base_k = __builtin_amdgcn_ds_bpermute(max(qk[k], 0) << 2, my_base);
                            if (qk[k] >= 0) {
                                const int pos = base_k + rk[k];
                                kc.run[0][pos] = pz[k];
                                kc.run[1][pos] = coord_x(lc1, u, pz[k]);
                                kc.run[2][pos] = coord_y(lc1, v, pz[k]);
                            }
                            v++;
                            if (v == lc1.rows_i) {
                                v = 0;
                                u++;
                            }
                        }
                    }
                    earlier += mem5;
                }
                const int sum_c = (tid < 3 * nown) ? tid / 3 : 0;
                const int sum_n = __builtin_amdgcn_ds_bpermute(sum_c << 2, grp_members);
                const int sum_o = __builtin_amdgcn_ds_bpermute(sum_c << 2, run_start);
                PH(3);
                __syncthreads();
                PH(4);
                if (tid < 3 * nown) {  // strictly front to back per sum
                    const int r = tid - 3 * sum_c;
                    const LDS float *src = &kc.run[r][sum_o];
                    const int n = sum_n;
                    int j = 0;
                    const int head = min(n, (4 - (sum_o & 3)) & 3);  // up to the first 16-byte boundary of the run
                    for (; j < head; j++) acc += src[j];
                    // sixteen values per trip from four 16-byte LDS reads, the next sixteen in flight meanwhile: the chain of
                    // additions is what remains
                    typedef LDS const vfloat4 lcf4;
                    vfloat4 v0, v1, v2, v3, w0, w1, w2, w3;
#define KMC_LOAD16(a0, a1, a2, a3, at)      \
    a0 = *(lcf4 *)(src + (at));             \
    a1 = *(lcf4 *)(src + (at) + 4);         \
    a2 = *(lcf4 *)(src + (at) + 8);         \
    a3 = *(lcf4 *)(src + (at) + 12);
#define KMC_ADD16(a0, a1, a2, a3)                                \
    acc += a0.x; acc += a0.y; acc += a0.z; acc += a0.w;          \
    acc += a1.x; acc += a1.y; acc += a1.z; acc += a1.w;          \
    acc += a2.x; acc += a2.y; acc += a2.z; acc += a2.w;          \
    acc += a3.x; acc += a3.y; acc += a3.z; acc += a3.w;
                    // two register sets take turns (no copies): while one is added the other is in flight
                    if (j + 16 <= n) { KMC_LOAD16(v0, v1, v2, v3, j) }
                    for (; j + 48 <= n; j += 32) {
                        KMC_LOAD16(w0, w1, w2, w3, j + 16)
                        __builtin_amdgcn_sched_barrier(0);
                        KMC_ADD16(v0, v1, v2, v3)
                        KMC_LOAD16(v0, v1, v2, v3, j + 32)
                        __builtin_amdgcn_sched_barrier(0);
                        KMC_ADD16(w0, w1, w2, w3)
                    }
                    if (j + 32 <= n) {
                        KMC_LOAD16(w0, w1, w2, w3, j + 16)
                        __builtin_amdgcn_sched_barrier(0);
                        KMC_ADD16(v0, v1, v2, v3)
                        KMC_ADD16(w0, w1, w2, w3)
                        j += 32;
                    } else if (j + 16 <= n) {
                        KMC_ADD16(v0, v1, v2, v3)
                        j += 16;
                    }
#undef KMC_LOAD16
#undef KMC_ADD16
                    for (; j < n; j++) acc += src[j];
                    total += n;
                }
                PH(5);
                g0 = g1;
                if (g0 < R) __syncthreads();  // the next group's scatter may overwrite the runs
            }
        }
        // ---- the owned centres -> cs.in (KMeans.cpp:219-226)
        if (tid < 4 * KMC_MAX_OWN) cs.in[tid] = 0u;
        __syncthreads();
        if (tid < 3 * nown) {
            const int c = tid / 3, r = tid - 3 * c;
            if (total > 0) acc /= float(total);
            cs.in[4 * c + r] = __float_as_uint(acc);
            if (r == 0) cs.in[4 * c + 3] = (unsigned)total;
        }
    PH(6);
#ifdef SF_KMC_FINE
    if (tid == 0 && prof_out)
        for (int i = 0; i < 7; i++) prof_out[i] += ph[i];
#endif
#undef PH
}

__device__ __noinline__ void stage_kmeans_cluster(const KArgs &a, int b, LDS KmClusterShared &sh, LDS ClusterShared &cs, int tid) {
    LDS KmShared &s = sh.km;
    LDS KmClShared &kc = sh.kc;
    const int lane = tid & 63, wave = tid >> 6;
    const int G = cl_G(cs), rank = cl_rank(cs);
    const bool writer = cl_writer(cs);
    const int nown = (SF_NC - rank + G - 1) / G;  // clusters (seeds) rank, rank + G, ... : local index q <-> cluster rank + q G
    const size_t sb = (size_t)b * a.n_tot;
    const auto depth = as_global((const float *)pyr_plane(a, b, 0, 0));
    const LevelCoord lc0 = level_coord(a, 0), lc1 = level_coord(a, 1);
    const auto labels = as_global(a.labels + sb);
    StreamState &st = a.state[b];
    long long kt = wall_clock64();
#define KMC_MARK(slot)                             \
    do {                                           \
        if (tid == 0 && writer) {                  \
            const long long now_ = wall_clock64(); \
            st.prof[slot] += now_ - kt;            \
            kt = now_;                             \
        }                                          \
    } while (0)

    const int rows_km = a.lrows[1], cols_km = a.lcols[1], n1 = a.ln[1], o1 = a.loff[1];
    const int nq1 = n1 / 4;                                  // quads of level 1 (every level holds a multiple of 4 pixels)
    gu32w *lab1w = (gu32w *)(labels + o1);                   // level-1 labels as words (o1 = n0 is a multiple of 4)
    typedef __attribute__((address_space(1))) const vfloat4 gcf4;
    gcf4 *depth1q = (gcf4 *)(depth + o1);
    int qb1, qe1;  // this workgroup's share of the level-1 quads
    cluster_range(cs, nq1, 1, qb1, qe1);

    // ------------------------------------------------------------------ initializeKMeans (K1): seed labels by pixel share
    if (tid < SF_NC) {
        s.useed[tid] = km_seed_u(cols_km, tid);
        s.vseed[tid] = km_seed_v(rows_km, tid);
        s.prefix[tid] = 0;
    }
    __syncthreads();
    {
        const auto seed_w = as_global((const unsigned *)a.km_seed_lab);  // the table of sf_kmeans.h, four pixels per word
        for (int q = qb1 + tid; q < qe1; q += SF_NT) {
            const vfloat4 dz4 = depth1q[q];
            const float dz[4] = {dz4.x, dz4.y, dz4.z, dz4.w};
            const unsigned sw = seed_w[q];
            unsigned word = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) word |= ((dz[k] != 0.f) ? ((sw >> (8 * k)) & 255u) : (unsigned)SF_NC) << (8 * k);
            st_word_agent(lab1w + q, word);
        }
    }
    labels_rendezvous(cs, tid);

    // per-seed median depth of the seeds this workgroup owns: radix select over ALL pixels of the level
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        for (int q = tid; q < nown * 256; q += SF_NT) s.hist[q] = 0;
        __syncthreads();
        for (int q = tid; q < nq1; q += SF_NT) {
            const unsigned word = ld_word_agent(lab1w + q);
            const vfloat4 dz4 = depth1q[q];
            const unsigned bits[4] = {__float_as_uint(dz4.x), __float_as_uint(dz4.y), __float_as_uint(dz4.z), __float_as_uint(dz4.w)};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned lb = (word >> (8 * k)) & 255u;
                if (lb < SF_NC && (int)(lb % (unsigned)G) == rank) {
                    const unsigned qi = lb / (unsigned)G;
                    if (pass == 0 || (bits[k] >> (shift + 8)) == s.prefix[qi]) lds_add(&s.hist[qi * 256 + ((bits[k] >> shift) & 255u)], 1u);
                }
            }
        }
        __syncthreads();
        for (int l = wave; l < nown; l += SF_NW) {  // one wave per owned seed (see stage_kmeans)
            typedef unsigned __attribute__((ext_vector_type(4))) vuint4;
            const vuint4 h4 = *(const LDS vuint4 *)&s.hist[l * 256 + 4 * lane];
            const int p0 = (int)h4.x, p1 = p0 + (int)h4.y, p2 = p1 + (int)h4.z, p3 = p2 + (int)h4.w;
            int incl = p3;
            SF_DPP_REDUCE(incl, dpp_i32, sf_op_add)
            const int size = __builtin_amdgcn_readlane(incl, 63);
            const unsigned k = (pass == 0) ? (unsigned)size / 2u : s.krank[l];
            if (pass == 0 && lane == 0) s.count[l] = size;
            if ((pass == 0 ? size : s.count[l]) > 0) {
                const unsigned long long over = __ballot((unsigned)incl > k);
                const int src = __ffsll((long long)over) - 1;
                if (lane == src) {
                    const unsigned cum0 = (unsigned)(incl - p3);
                    int bin;
                    unsigned cum;
                    if (cum0 + (unsigned)p0 > k) { bin = 0; cum = cum0; }
                    else if (cum0 + (unsigned)p1 > k) { bin = 1; cum = cum0 + (unsigned)p0; }
                    else if (cum0 + (unsigned)p2 > k) { bin = 2; cum = cum0 + (unsigned)p1; }
                    else { bin = 3; cum = cum0 + (unsigned)p2; }
                    s.prefix[l] = (s.prefix[l] << 8) | (unsigned)(4 * lane + bin);
                    s.krank[l] = k - cum;
                }
            }
        }
        __syncthreads();
    }
    // medians + member counts of the owned seeds -> everybody
    if (tid < KMC_MAX_OWN) {
        cs.in[2 * tid] = (tid < nown) ? s.prefix[tid] : 0u;
        cs.in[2 * tid + 1] = (tid < nown) ? (unsigned)s.count[tid] : 0u;
    }
    cluster_gather(cs, 2 * KMC_MAX_OWN, tid);
    if (tid < SF_NC) {
        const int p = tid % G, qi = tid / G;
        const unsigned zbits = cs.all[p * 2 * KMC_MAX_OWN + 2 * qi];
        const int cnt = (int)cs.all[p * 2 * KMC_MAX_OWN + 2 * qi + 1];
        const float inv_f_i = 2.f * a.tan_half_fovh / float(cols_km);
        const float disp_u_i = 0.5f * (cols_km - 1);
        const float disp_v_i = 0.5f * (rows_km - 1);
        float z = 0.f, x = 0.f, y = 0.f;
        if (cnt > 0) {
            z = __uint_as_float(zbits);
            x = (s.useed[tid] - disp_u_i) * z * inv_f_i;
            y = (s.vseed[tid] - disp_v_i) * z * inv_f_i;
        }
        s.cent_a[3 * tid] = z;
        s.cent_a[3 * tid + 1] = x;
        s.cent_a[3 * tid + 2] = y;
    }
    __syncthreads();
    KMC_MARK(PF_KM_INIT);

    // ------------------------------------------------------------------ Lloyd iterations (K2)
    int iters = 0;
    for (int it = 0; it < 9; it++) {
        iters++;
        if (tid == 0) kc.present = 0;  // ordered before the assignment loop by the barriers of km_sort_centres
        km_sort_centres(s, tid);
        KMC_MARK(PF_KM_SORT);
        // ---- assignment of this workgroup's share (KMeans.cpp:187-213)
        unsigned present = 0;  // labels this thread has seen in its quads
        for (int base = qb1; base < qe1; base += SF_NT) {
            const int q = base + tid;
            const bool in = q < qe1;
            const int qq = in ? q : qb1;
            const unsigned word = ld_word_agent(lab1w + qq);
            const vfloat4 dz4 = depth1q[qq];
            float pz[4] = {dz4.x, dz4.y, dz4.z, dz4.w}, px[4], py[4];
            int old[4], best[4];
            bool valid[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int u, v;
                split_uv(lc1, 4 * qq + k, u, v);
                px[k] = coord_x(lc1, u, pz[k]);
                py[k] = coord_y(lc1, v, pz[k]);
                valid[k] = in && pz[k] != 0.f;
                old[k] = valid[k] ? (int)((word >> (8 * k)) & 255u) : 0;
            }
            km_search_n<4>(s, old, pz, px, py, valid, best);
            unsigned out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned lb = valid[k] ? (unsigned)best[k] : ((word >> (8 * k)) & 255u);
                out |= lb << (8 * k);
                if (in && lb < SF_NC) present |= 1u << lb;
            }
            if (in) st_word_agent(lab1w + q, out);
        }
        {   // the wave's labels in one LDS atomic (an OR over the lanes on the DPP network: 0 shifts in at the row edges)
            int m = (int)present;
            SF_DPP_REDUCE(m, dpp_i32, sf_op_ori)
            if (lane == 63 && m) lds_or(&kc.present, (unsigned)m);
        }
        labels_rendezvous_with_mask(cs, kc.present, tid);
        KMC_MARK(PF_KM_ASSIGN);
        // ---- ordered sums of the clusters this workgroup owns (KMeans.cpp:215-221) -> cs.in
        {
            // the shares (workgroup p labelled quads [p per, (p + 1) per)) that hold members of an owned cluster
            unsigned own_mask = 0;
            for (int c = 0; c < nown; c++) own_mask |= 1u << (rank + c * G);
            int first_p = G, last_p = -1;
            for (int p = 0; p < G; p++)
                if (cs.all[p] & own_mask) {
                    first_p = min(first_p, p);
                    last_p = p;
                }
            const int per = (nq1 + G - 1) / G;  // cluster_range(cs, nq1, 1, ...)
            const int qlo = (last_p < 0) ? 0 : min(nq1, first_p * per), qhi = (last_p < 0) ? 0 : min(nq1, (last_p + 1) * per);
            const int n_walk = (uniform_i(qhi) - uniform_i(qlo) + SF_NT - 1) / SF_NT;
            long long *pr = writer ? &st.prof[16] : nullptr;
            if (n_walk <= 1) kmc_collect_and_sum<1>(kc, cs, lab1w, depth1q, lc1, qlo, qhi, G, rank, nown, tid, pr);
            else if (n_walk <= 2) kmc_collect_and_sum<2>(kc, cs, lab1w, depth1q, lc1, qlo, qhi, G, rank, nown, tid, pr);
            else kmc_collect_and_sum<KMC_R>(kc, cs, lab1w, depth1q, lc1, qlo, qhi, G, rank, nown, tid, pr);
        }
        cluster_gather(cs, 4 * KMC_MAX_OWN, tid);
        if (tid < 3 * SF_NC) {
            const int c = tid / 3, r = tid - 3 * c;
            s.cent_b[tid] = __uint_as_float(cs.all[(c % G) * 4 * KMC_MAX_OWN + 4 * (c / G) + r]);
        }
        __syncthreads();
        KMC_MARK(PF_KM_SUM);
        if (tid < 64) {
            float dmax = 0.f;
            for (int q = tid; q < 3 * SF_NC; q += 64) dmax = std_max(dmax, fabsf(s.cent_a[q] - s.cent_b[q]));
            dmax = wave_max_f32(dmax);
            if (tid == 0) s.stop = (dmax < 1e-2f) ? 1 : 0;
        }
        __syncthreads();
        if (tid < 3 * SF_NC) s.cent_a[tid] = s.cent_b[tid];
        const int stop = __builtin_amdgcn_readfirstlane(s.stop);
        __syncthreads();
        if (stop) break;
    }
    if (writer && commit_ok(cs)) {
        if (tid < 3 * SF_NC) st.kmeans[tid] = s.cent_a[tid];
        if (tid == 0) a.stats[b].kmeans_iters = iters;
    }

    // ------------------------------------------------------------------ labels at full resolution, by pixel share
    km_sort_centres(s, tid);
    const int rows0 = a.lrows[0], cols0 = a.lcols[0], n0 = a.ln[0];
    {
        gu32w *lab0w = (gu32w *)labels;
        gcf4 *depth0q = (gcf4 *)depth;
        gcu8b *lab1b = (gcu8b *)(labels + o1);
        int qb0, qe0;
        cluster_range(cs, n0 / 4, 1, qb0, qe0);
        for (int base = qb0; base < qe0; base += SF_NT) {
            const int q = base + tid;
            const bool in = q < qe0;
            const int qq = in ? q : qb0;
            const vfloat4 dz4 = depth0q[qq];
            float pz[4] = {dz4.x, dz4.y, dz4.z, dz4.w}, px[4], py[4];
            int start[4], lab[4];
            bool act[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int idx = 4 * qq + k;
                const int u = idx / rows0, v = idx - u * rows0;
                px[k] = coord_x(lc0, u, pz[k]);
                py[k] = coord_y(lc0, v, pz[k]);
                const int low = (int)ld_byte_agent(lab1b + (v / 2) + (u / 2) * rows_km);
                act[k] = in && pz[k] != 0.f;
                start[k] = (low == SF_NC) ? 0 : low;
            }
            km_search_n<4>(s, start, pz, px, py, act, lab);
            unsigned out = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) out |= (unsigned)(act[k] ? lab[k] : SF_NC) << (8 * k);
            if (in) st_word_agent(lab0w + q, out);
        }
    }
    if (tid < SF_NC) s.conn[tid] = 1u << tid;
    labels_rendezvous(cs, tid);
    KMC_MARK(PF_KM_LABEL0);

    // ------------------------------------------------------------------ computeRegionConnectivity (K3), by pixel share
    {
        gcu8b *lab0b = (gcu8b *)labels;
        const float dist2_threshold = sqf(0.03f * 120.f / float(rows0));
        int pb, pe;
        cluster_range(cs, n0, 1, pb, pe);
        for (int base = pb + tid; base < pe; base += SF_NT * SF_LOAD_BATCH) {
            float dz[SF_LOAD_BATCH], dzd[SF_LOAD_BATCH], dzr[SF_LOAD_BATCH];
            int la[SF_LOAD_BATCH], ld[SF_LOAD_BATCH], lr[SF_LOAD_BATCH], uu[SF_LOAD_BATCH], vv[SF_LOAD_BATCH];
            bool in[SF_LOAD_BATCH];
#pragma unroll
            for (int q = 0; q < SF_LOAD_BATCH; q++) {
                const int idx = min(base + q * SF_NT, pe - 1);
                split_uv(lc0, idx, uu[q], vv[q]);
                in[q] = (base + q * SF_NT < pe) && uu[q] < cols0 - 1 && vv[q] < rows0 - 1;
                const int i1 = in[q] ? idx + 1 : idx, i2 = in[q] ? idx + rows0 : idx;
                dz[q] = depth[idx];
                dzd[q] = depth[i1];
                dzr[q] = depth[i2];
                la[q] = (int)ld_byte_agent(lab0b + idx);
                ld[q] = (int)ld_byte_agent(lab0b + i1);
                lr[q] = (int)ld_byte_agent(lab0b + i2);
            }
#pragma unroll
            for (int q = 0; q < SF_LOAD_BATCH; q++) {
                if (!in[q] || dz[q] == 0.f) continue;
                const int u = uu[q], v = vv[q];
                const float yc = coord_y(lc0, v, dz[q]), yd = coord_y(lc0, v + 1, dzd[q]);
                const float xc = coord_x(lc0, u, dz[q]), xr = coord_x(lc0, u + 1, dzr[q]);
                if (la[q] != ld[q] && ld[q] != SF_NC) {
                    const float disty = sqf(dz[q] - dzd[q]) + sqf(yc - yd);
                    if (disty < dist2_threshold) {
                        lds_or(&s.conn[la[q]], 1u << ld[q]);
                        lds_or(&s.conn[ld[q]], 1u << la[q]);
                    }
                }
                if (la[q] != lr[q] && lr[q] != SF_NC) {
                    const float distx = sqf(dz[q] - dzr[q]) + sqf(xc - xr);
                    if (distx < dist2_threshold) {
                        lds_or(&s.conn[la[q]], 1u << lr[q]);
                        lds_or(&s.conn[lr[q]], 1u << la[q]);
                    }
                }
            }
        }
        __syncthreads();
        if (tid < SF_NC) cs.in[tid] = s.conn[tid];
        cluster_gather(cs, SF_NC, tid);
        if (tid < SF_NC) {
            unsigned m = 0;
            for (int p = 0; p < G; p++) m |= cs.all[p * SF_NC + tid];
            s.conn[tid] = m;
            if (writer && commit_ok(cs)) st.conn[tid] = m;
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ createClustersPyramidUsingKMeans (K4), every G-th block
    for (int L = 2; L < a.levels; L++) {
        const int n = a.ln[L], o = a.loff[L];
        const LevelCoord lc = level_coord(a, L);
        for (int idx = tid + rank * SF_NT; idx < n; idx += SF_NT * G) {
            const float pz = depth[o + idx];
            int lab = SF_NC;
            if (pz != 0.f) {
                int u, v;
                split_uv(lc, idx, u, v);
                const float px = coord_x(lc, u, pz), py = coord_y(lc, v, pz);
                int label = 0;
                float min_dist = sqdist3(s.cent_a[0], s.cent_a[1], s.cent_a[2], pz, px, py);
                for (int l = 1; l < SF_NC; l++) {
                    if (s.pair_dist[label * SF_NC + l] > 4.f * min_dist) continue;
                    const float dh = sqdist3(s.cent_a[3 * l], s.cent_a[3 * l + 1], s.cent_a[3 * l + 2], pz, px, py);
                    if (dh < min_dist) {
                        label = l;
                        min_dist = dh;
                    }
                }
                lab = label;
            }
            labels[o + idx] = (uint8_t)lab;
        }
    }
    __syncthreads();
    KMC_MARK(PF_KM_CONN_PYR);
#undef KMC_MARK
}
