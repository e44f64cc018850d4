// sf_kmeans.h — geometric clustering of one stream: K-means(24) on (z, x, y), region connectivity,
// label pyramid.  Replaces initializeKMeans / kMeans3DCoord / computeRegionConnectivity /
// createClustersPyramidUsingKMeans (reference KMeans.cpp:63-391).
//
// Bit-exact labels need the reference's SEQUENTIAL float32 centre sums (KMeans.cpp:215-221, pixels
// in column-major order).  The assignment step is data parallel; the centre update is made
// parallel across the 72 (cluster, coordinate) sums while each sum stays strictly sequential:
//   pass A  assign labels, count members per (wave range, label) with ballots
//   scan    exclusive offsets per (wave range, label)
//   pass B  stable partition: write (z,x,y) of every pixel to its cluster's contiguous run, in
//           pixel order (rank = members in earlier wave ranges + earlier iterations + lower lanes)
//   sum     lane (c, r) adds the run of cluster c, coordinate r, front to back
// The per-seed median depth (std::nth_element at size/2, KMeans.cpp:104-123) is an exact
// 4 x 8-bit radix select on the float bit patterns.
// The sorted centre-distance lists (std::sort, KMeans.cpp:172-183) use a stable rank sort; it
// equals std::sort whenever no two distances in a row are exactly equal.
#pragma once

#include "sf_device_common.h"

#define KM_CHUNK (SF_NT * SF_LOAD_BATCH)  // pixels per chunk of the Lloyd pass: SF_LOAD_BATCH per lane
#define KM_ROW (SF_NC + 1)                // entries per row of the candidate table (one of padding: LDS banks)

struct KmShared {
    float cent_a[3 * SF_NC], cent_b[3 * SF_NC];
    union {  // the radix-select histogram (initialisation only) and the centre-distance tables (Lloyd iterations onwards)
        unsigned hist[SF_NC * 256];
        struct {
            // row l, entry j: the j-th nearest other centre of centre l as (distance to it, its z, x, y) -- ONE 16-byte LDS read
            // whose address does not depend on what the previous read returned (round 5; it was (distance, index) and the
            // centre a second, dependent read) -- and its index, read once, after the walk, for the winner only
            // (rows of KM_ROW = 25 entries: with 24 the rows of different centres start 96 dwords apart, i.e. on TWO groups of LDS
            // banks for all of them; 100 dwords apart they start on 16 different ones)
            vfloat4 cand4[SF_NC * KM_ROW];
            uint8_t cand_i[SF_NC * KM_ROW];
            float pair_dist[SF_NC * SF_NC];
            float chunk[3][KM_CHUNK];       // (z, x, y) of one chunk of pixels, stably partitioned by cluster
        };
    };
    vfloat4 cent4[SF_NC];          // (z, x, y, 0) of centre l: one 16-byte LDS read
    int wcnt[SF_NW][SF_NC];        // members per (wave range, label); then exclusive offsets
    int count[SF_NC];
    unsigned conn[SF_NC];
    unsigned useed[SF_NC], vseed[SF_NC];
    unsigned prefix[SF_NC], krank[SF_NC];
    float red[SF_NW];
    int stop;
};

__device__ __forceinline__ float sqdist3(float a0, float a1, float a2, float b0, float b1, float b2) {
    const float d0 = a0 - b0, d1 = a1 - b1, d2 = a2 - b2;
    return (d0 * d0 + d1 * d1) + d2 * d2;
}

// pairwise centre distances + per-row stable rank sort (KMeans.cpp:172-183)
__device__ __noinline__ void km_sort_centres(LDS KmShared &s, int tid) {
    for (int q = tid; q < SF_NC * SF_NC; q += SF_NT) {
        const int l = q / SF_NC, li = q - l * SF_NC;
        s.pair_dist[q] = sqdist3(s.cent_a[3 * l], s.cent_a[3 * l + 1], s.cent_a[3 * l + 2], s.cent_a[3 * li],
                                 s.cent_a[3 * li + 1], s.cent_a[3 * li + 2]);
    }
    __syncthreads();
    for (int q = tid; q < SF_NC * SF_NC; q += SF_NT) {
        const int l = q / SF_NC, li = q - l * SF_NC;
        const float d = s.pair_dist[q];
        int rank = 0;
        for (int lj = 0; lj < SF_NC; lj++) {
            const float dj = s.pair_dist[l * SF_NC + lj];
            rank += (dj < d || (dj == d && lj < li)) ? 1 : 0;
        }
        s.cand4[l * KM_ROW + rank] = vfloat4{d, s.cent_a[3 * li], s.cent_a[3 * li + 1], s.cent_a[3 * li + 2]};
        s.cand_i[l * KM_ROW + rank] = (uint8_t)li;
    }
    if (tid < SF_NC) s.cent4[tid] = vfloat4{s.cent_a[3 * tid], s.cent_a[3 * tid + 1], s.cent_a[3 * tid + 2], 0.f};
    __syncthreads();
}

// pruned nearest-centre search starting from `last` (KMeans.cpp:196-212 and :263-285)
__device__ __forceinline__ int km_search(const LDS KmShared &s, int last, float pz, float px, float py) {
    int best_j = 0;  // position of the best candidate in row `last` (0: `last` itself)
    const vfloat4 c0 = s.cent4[last];
    const float d_last = sqdist3(c0.x, c0.y, c0.z, pz, px, py);
    float best_d = d_last;
    const float lim = 4.f * d_last;
    vfloat4 cd = s.cand4[last * KM_ROW + 1];
    for (int li = 1; li < SF_NC; li++) {
        if (cd.x > lim) break;
        const vfloat4 cc = cd;
        if (li + 1 < SF_NC) cd = s.cand4[last * KM_ROW + li + 1];  // next candidate in flight during the distance
        const float dl = sqdist3(cc.y, cc.z, cc.w, pz, px, py);
        if (dl < best_d) {
            best_d = dl;
            best_j = li;
        }
    }
    return best_j ? (int)s.cand_i[last * KM_ROW + best_j] : last;
}

// The same search for N independent pixels of a lane in lock step: the dependent LDS reads of one pixel
// (candidate -> centre) overlap those of the others. Per pixel the candidate sequence, the comparisons and the
// exit rule (first candidate farther than 4 d_last ends the search) are those of km_search(); pixels whose
// search has ended ride along without effect. act[k] = false: pixel k is not searched (best[k] = last[k]).
template <int N>
__device__ __forceinline__ void km_search_n(const LDS KmShared &s, const int (&last)[N], const float (&pz)[N], const float (&px)[N],
                                            const float (&py)[N], const bool (&act)[N], int (&best)[N], int *trips = nullptr) {
    float best_d[N], lim[N];
    int best_j[N], row[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const vfloat4 c0 = s.cent4[last[k]];
        best_j[k] = 0;
        row[k] = last[k] * KM_ROW;
        best_d[k] = sqdist3(c0.x, c0.y, c0.z, pz[k], px[k], py[k]);
        // the rows of candidates are sorted by distance, so "this candidate is farther than 4 d_last" stays true once it is:
        // no per-pixel "still running" flag is needed, and a pixel that is not searched gets a limit nothing passes
        lim[k] = act[k] ? 4.f * best_d[k] : -1.f;
    }
    for (int li = 1; li < SF_NC; li++) {
        // the trip's N reads (candidate li of every pixel's row: distance and centre in one 16-byte entry) are issued together;
        // nothing is carried from trip to trip but the running best (a prefetch of the next trip's entries cost 16 more
        // registers and spilled inside the chunk loop of the 96-register kernel: measured 6.5 % slower than round 4's table)
        vfloat4 cd[N];
#pragma unroll
        for (int k = 0; k < N; k++) cd[k] = s.cand4[row[k] + li];
        __builtin_amdgcn_sched_barrier(0);
        bool any = false;
#pragma unroll
        for (int k = 0; k < N; k++) any = any || !(cd[k].x > lim[k]);
        if (!__any(any)) break;
#ifdef SF_KM_FINE_PROFILE
        if (trips) (*trips)++;
#endif
#pragma unroll
        for (int k = 0; k < N; k++) {
            const float dl = sqdist3(cd[k].y, cd[k].z, cd[k].w, pz[k], px[k], py[k]);
            const bool upd = !(cd[k].x > lim[k]) && (dl < best_d[k]);
            best_d[k] = upd ? dl : best_d[k];
            best_j[k] = upd ? li : best_j[k];
        }
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int c = (int)s.cand_i[row[k] + best_j[k]];  // (entry 0 of a row is not read as a candidate: best_j = 0 means `last`)
        best[k] = best_j[k] ? c : last[k];
    }
}

__device__ __noinline__ void stage_kmeans(const KArgs &a, int b, LDS KmShared &s, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const size_t sb = (size_t)b * a.n_tot;
    const auto depth = as_global((const float *)pyr_plane(a, b, 0, 0));
    const auto depth0 = as_global(pyr_level(a, b, 0, 0, 0));  // level 0 may lie in the caller's frame pool (pyr_level)
    const LevelCoord lc0 = level_coord(a, 0), lc1 = level_coord(a, 1);
    const auto labels = as_global(a.labels + sb);
    StreamState &st = a.state[b];
    long long kt = wall_clock64();
#define KM_MARK(slot)                              \
    do {                                           \
        if (tid == 0) {                            \
            const long long now_ = wall_clock64(); \
            st.prof[slot] += now_ - kt;            \
            kt = now_;                             \
        }                                          \
    } while (0)
#ifdef SF_KM_FINE_PROFILE  // profiling build (make kmprof): phases of the Lloyd chunk loop (register accumulators, flushed once)
    long long fine[SF_PROF_SLOTS] = {};
#define KM_FINE(slot)                              \
    do {                                           \
        if (tid == 0) {                            \
            const long long now_ = wall_clock64(); \
            fine[slot] += now_ - kt;               \
            kt = now_;                             \
        }                                          \
    } while (0)
#else
#define KM_FINE(slot) do {} while (0)
#endif

    // ------------------------------------------------------------------ initializeKMeans (K1)
    const int rows_km = a.lrows[1], cols_km = a.lcols[1], n1 = a.ln[1], o1 = a.loff[1];
    if (tid < SF_NC) {
        s.useed[tid] = km_seed_u(cols_km, tid);
        s.vseed[tid] = km_seed_v(rows_km, tid);
        s.prefix[tid] = 0;
    }
    {
        const auto seed_lab = as_global(a.km_seed_lab);
        for (int base = tid; base < n1; base += SF_NT * SF_LOAD_BATCH) {
            float dz[SF_LOAD_BATCH];
            unsigned sl[SF_LOAD_BATCH];
#pragma unroll
            for (int k = 0; k < SF_LOAD_BATCH; k++) {
                const int idx = min(base + k * SF_NT, n1 - 1);
                dz[k] = gld(depth, o1 + idx);
                sl[k] = gld(seed_lab, idx);
            }
#pragma unroll
            for (int k = 0; k < SF_LOAD_BATCH; k++) {
                const int idx = base + k * SF_NT;
                if (idx < n1) gst(labels, o1 + idx, (uint8_t)((dz[k] != 0.f) ? sl[k] : SF_NC));
            }
        }
    }
    __syncthreads();
    // per-seed median depth: radix select, most significant byte first
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        for (int q = tid; q < SF_NC * 256; q += SF_NT) s.hist[q] = 0;
        __syncthreads();
        for (int base = tid; base < n1; base += SF_NT * SF_LOAD_BATCH) {
            unsigned lb[SF_LOAD_BATCH], bits[SF_LOAD_BATCH];
#pragma unroll
            for (int k = 0; k < SF_LOAD_BATCH; k++) {
                const int idx = min(base + k * SF_NT, n1 - 1);
                lb[k] = gld(labels, o1 + idx);
                bits[k] = __float_as_uint(gld(depth, o1 + idx));
            }
#pragma unroll
            for (int k = 0; k < SF_LOAD_BATCH; k++)
                if (base + k * SF_NT < n1 && lb[k] < SF_NC && (pass == 0 || (bits[k] >> (shift + 8)) == s.prefix[lb[k]]))
                    lds_add(&s.hist[lb[k] * 256 + ((bits[k] >> shift) & 255u)], 1u);
        }
        __syncthreads();
        // per label: the bin that holds rank k. One wave per label (four bins per lane, an inclusive DPP scan over the lane
        // totals, a ballot for the first lane whose running count exceeds k) instead of 24 lanes walking 256 bins each.
        for (int l = wave; l < SF_NC; l += SF_NW) {
            typedef unsigned __attribute__((ext_vector_type(4))) vuint4;
            const vuint4 h4 = *(const LDS vuint4 *)&s.hist[l * 256 + 4 * lane];
            const int p0 = (int)h4.x, p1 = p0 + (int)h4.y, p2 = p1 + (int)h4.z, p3 = p2 + (int)h4.w;
            int incl = p3;
            SF_DPP_REDUCE(incl, dpp_i32, sf_op_add)  // inclusive scan: lane 63 holds the label's size (pass 0) / the bucket's size
            const int size = __builtin_amdgcn_readlane(incl, 63);
            const unsigned k = (pass == 0) ? (unsigned)size / 2u : s.krank[l];
            if (pass == 0 && lane == 0) s.count[l] = size;
            if ((pass == 0 ? size : s.count[l]) > 0) {
                const unsigned long long over = __ballot((unsigned)incl > k);
                const int src = __ffsll((long long)over) - 1;  // first lane whose bins reach rank k
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
    if (tid < SF_NC) {
        const float inv_f_i = 2.f * a.tan_half_fovh / float(cols_km);
        const float disp_u_i = 0.5f * (cols_km - 1);
        const float disp_v_i = 0.5f * (rows_km - 1);
        float z = 0.f, x = 0.f, y = 0.f;
        if (s.count[tid] > 0) {
            z = __uint_as_float(s.prefix[tid]);
            x = (s.useed[tid] - disp_u_i) * z * inv_f_i;
            y = (s.vseed[tid] - disp_v_i) * z * inv_f_i;
        }
        s.cent_a[3 * tid] = z;
        s.cent_a[3 * tid + 1] = x;
        s.cent_a[3 * tid + 2] = y;
    }
    __syncthreads();

    KM_MARK(PF_KM_INIT);
    // ------------------------------------------------------------------ Lloyd iterations (K2)
    // One pass over the level per iteration, in chunks of KM_CHUNK pixels (pixel order): assign, partition the
    // chunk stably by cluster in LDS, then 72 lanes -- one per (cluster, coordinate) -- extend their running float
    // sums front to back (KMeans.cpp:215-221: centers_b.col(best_label) += p in pixel order; strictly sequential
    // per sum, parallel across sums). Nothing but the labels goes back to memory.
    const int n_chunks = (n1 + KM_CHUNK - 1) / KM_CHUNK;
    // pixel index steps as (column, row) steps: + 64 between a lane's pixels, + KM_CHUNK - 64 SF_LOAD_BATCH to the next chunk
    const int step64_u = 64 / rows_km, step64_v = 64 - step64_u * rows_km;
    const int stepc = KM_CHUNK - 64 * SF_LOAD_BATCH, stepc_u = stepc / rows_km, stepc_v = stepc - stepc_u * rows_km;
    int iters = 0;
    for (int it = 0; it < 9; it++) {
        iters++;
        km_sort_centres(s, tid);
        KM_MARK(PF_KM_SORT);
        float acc = 0.f;   // lane (c, r) = tid < 72: running sum of coordinate r over the members of cluster c
        int total = 0;     // lane l < 24 of wave 0: members of cluster l so far
        float pz[SF_LOAD_BATCH], nz[SF_LOAD_BATCH];
        int old[SF_LOAD_BATCH], nold[SF_LOAD_BATCH];
#pragma unroll
        for (int k = 0; k < SF_LOAD_BATCH; k++) {  // wave w owns pixels [256 w, 256 (w + 1)) of the chunk, k-major
            const int idx = min(wave * (64 * SF_LOAD_BATCH) + k * 64 + lane, n1 - 1);
            nz[k] = gld(depth, o1 + idx);
            nold[k] = gld(labels, o1 + idx);
        }
        int cu, cv;  // column / row of the lane's next pixel (beyond the level at the end: such pixels are not valid)
        split_uv(lc1, wave * (64 * SF_LOAD_BATCH) + lane, cu, cv);
        for (int ch = 0; ch < n_chunks; ch++) {
            const int base = ch * KM_CHUNK + wave * (64 * SF_LOAD_BATCH);
            float px[SF_LOAD_BATCH], py[SF_LOAD_BATCH];
            bool valid[SF_LOAD_BATCH];
            int best[SF_LOAD_BATCH];
#pragma unroll
            for (int k = 0; k < SF_LOAD_BATCH; k++) {
                pz[k] = nz[k];
                old[k] = nold[k];
            }
            if (ch + 1 < n_chunks) {  // the next chunk's loads are in flight during this one
#pragma unroll
                for (int k = 0; k < SF_LOAD_BATCH; k++) {
                    const int idx = min(base + KM_CHUNK + k * 64 + lane, n1 - 1);
                    nz[k] = gld(depth, o1 + idx);
                    nold[k] = gld(labels, o1 + idx);
                }
            }
#pragma unroll
            for (int k = 0; k < SF_LOAD_BATCH; k++) {
                const int idx = base + k * 64 + lane;
                px[k] = coord_x(lc1, cu, pz[k]);  // (cu, cv) = column / row of pixel idx, stepped instead of divided
                py[k] = coord_y(lc1, cv, pz[k]);
                valid[k] = (idx < n1) && pz[k] != 0.f;
                old[k] = valid[k] ? old[k] : 0;  // a safe table row for pixels that are not searched
                cv += step64_v;
                cu += step64_u;
                if (cv >= rows_km) {
                    cv -= rows_km;
                    cu++;
                }
            }
            cv += stepc_v;  // from idx + 4 * 64 on to the lane's first pixel of the next chunk
            cu += stepc_u;
            if (cv >= rows_km) {
                cv -= rows_km;
                cu++;
            }
#ifdef SF_KM_FINE_PROFILE
            int trips = 0;
            km_search_n<SF_LOAD_BATCH>(s, old, pz, px, py, valid, best, &trips);
            if (tid == 0) {
                fine[21] += trips;
                fine[22] += 1;
            }
#else
            km_search_n<SF_LOAD_BATCH>(s, old, pz, px, py, valid, best);
#endif
            KM_FINE(PF_KM_ASSIGN);
            // members of this wave's part of the chunk per label (lane l < 24 holds the count of label l), and for every
            // pixel its rank among the wave's members of the same label in pixel order (k-major, then lane): one pass of
            // ballots gives both
            int cnt = 0;
            int rank[SF_LOAD_BATCH];
#pragma unroll
            for (int k = 0; k < SF_LOAD_BATCH; k++) {
                const int idx = base + k * 64 + lane;
                rank[k] = 0;
                if (valid[k]) gst(labels, o1 + idx, (uint8_t)best[k]);
                unsigned long long rem = __ballot(valid[k]);
                while (rem) {
                    const int src = __ffsll((long long)rem) - 1;
                    const int l = __builtin_amdgcn_readlane(best[k], src);
                    const unsigned long long m = __ballot(valid[k] && best[k] == l);
                    const int start = __builtin_amdgcn_readlane(cnt, l);  // members of label l among the wave's earlier pixels
                    if (valid[k] && best[k] == l) rank[k] = start + __popcll(m & ((1ull << lane) - 1ull));
                    if (lane == l) cnt += __popcll(m);
                    rem &= ~m;
                }
            }
            if (lane < SF_NC) s.wcnt[wave][lane] = cnt;
            __syncthreads();  // also: the previous chunk's sums have consumed s.chunk
            // every wave derives the offsets itself: lane l < 24 sums the waves' counts of label l (members in earlier waves,
            // members in the chunk), an inclusive DPP scan over the labels gives the cluster's run start
            int before = 0, members = 0;
            if (lane < SF_NC) {
                int c[SF_NW];
#pragma unroll
                for (int w = 0; w < SF_NW; w++) c[w] = s.wcnt[w][lane];
#pragma unroll
                for (int w = 0; w < SF_NW; w++) {
                    before += (w < wave) ? c[w] : 0;
                    members += c[w];
                }
            }
            int incl = members;  // lanes >= 24 hold 0: the scan over the first two rows is the scan over the labels
            incl += dpp_i32<0x111, 0xf>(incl);
            incl += dpp_i32<0x112, 0xf>(incl);
            incl += dpp_i32<0x114, 0xf>(incl);
            incl += dpp_i32<0x118, 0xf>(incl);
            incl += dpp_i32<0x142, 0xa>(incl);
            const int run_start = incl - members;         // lane l: where cluster l's run begins in s.chunk
            const int my_base = run_start + before;       // ... and where this wave's members of it begin
            // stable positions: cluster run start + members in earlier waves + members earlier in this wave
#pragma unroll
            for (int k = 0; k < SF_LOAD_BATCH; k++) {
                const int base_k = __builtin_amdgcn_ds_bpermute(best[k] << 2, my_base);
                if (valid[k]) {
                    const int pos = base_k + rank[k];
                    s.chunk[0][pos] = pz[k];
                    s.chunk[1][pos] = px[k];
                    s.chunk[2][pos] = py[k];
                }
            }
            const int sum_c = (tid < 3 * SF_NC) ? tid / 3 : 0;  // lane (c, r) of the ordered sums: its cluster's run
            const int sum_n = __builtin_amdgcn_ds_bpermute(sum_c << 2, members);
            const int sum_o = __builtin_amdgcn_ds_bpermute(sum_c << 2, run_start);
            __syncthreads();
            KM_FINE(PF_KM_PARTITION);
            if (tid < 3 * SF_NC) {  // the ordered sums, continued over this chunk's members: strictly front to back per sum;
                                    // the next eight values are in flight while the current eight are added
                const int r = tid - 3 * sum_c;
                const LDS float *src = &s.chunk[r][sum_o];
                const int n = sum_n;
                float v[8], w[8];
                int j = 0;
                if (n >= 8) {
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = src[q];
                }
                while (j + 24 <= n) {  // v = elements [j, j + 8); the two register sets take turns (no copies)
#pragma unroll
                    for (int q = 0; q < 8; q++) w[q] = src[j + 8 + q];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 8; q++) acc += v[q];
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = src[j + 16 + q];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 8; q++) acc += w[q];
                    j += 16;
                }
                if (j + 16 <= n) {
#pragma unroll
                    for (int q = 0; q < 8; q++) w[q] = src[j + 8 + q];
#pragma unroll
                    for (int q = 0; q < 8; q++) acc += v[q];
#pragma unroll
                    for (int q = 0; q < 8; q++) acc += w[q];
                    j += 16;
                } else if (j + 8 <= n) {
#pragma unroll
                    for (int q = 0; q < 8; q++) acc += v[q];
                    j += 8;
                }
                for (; j < n; j++) acc += src[j];
            }
            if (tid < SF_NC) total += members;
            KM_FINE(PF_KM_SUM);
        }
        __syncthreads();
        if (tid < SF_NC) s.count[tid] = total;
        __syncthreads();
        KM_MARK(PF_KM_ASSIGN);
        if (tid < 3 * SF_NC) {
            const int n = s.count[tid / 3];
            if (n > 0) acc /= float(n);
            s.cent_b[tid] = acc;  // cent_b[r + 3c] with tid = 3c + r
        }
        __syncthreads();
        KM_MARK(PF_KM_SUM);
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
    if (tid < 3 * SF_NC) st.kmeans[tid] = s.cent_a[tid];
    if (tid == 0) a.stats[b].kmeans_iters = iters;
#ifdef SF_KM_FINE_PROFILE
    if (tid == 0) {
        st.prof[PF_KM_ASSIGN] += fine[PF_KM_ASSIGN];
        st.prof[PF_KM_PARTITION] += fine[PF_KM_PARTITION];
        st.prof[PF_KM_SUM] += fine[PF_KM_SUM];
        st.prof[21] += fine[21];
        st.prof[22] += fine[22];
    }
#endif

    // ------------------------------------------------------------------ labels at full resolution
    km_sort_centres(s, tid);
    {
        // A wave labels 8 x 8 pixel blocks (lane = 8 columns x 8 rows), SF_LOAD_BATCH blocks in lock step: the trips of the
        // lock-step search are set by the pixel that needs most, and a compact block mostly lies inside one cluster where a
        // 64-pixel column segment crosses two or three borders. (No order to respect here, unlike in the Lloyd pass.)
        const int rows0 = a.lrows[0], cols0 = a.lcols[0];
        const int bv = (rows0 + 7) / 8, bu = (cols0 + 7) / 8, n_blocks = bv * bu;
        const int dv = lane & 7, du = lane >> 3;
        for (int blk0 = wave * SF_LOAD_BATCH; blk0 < n_blocks; blk0 += SF_NW * SF_LOAD_BATCH) {
            float pz[SF_LOAD_BATCH], px[SF_LOAD_BATCH], py[SF_LOAD_BATCH];
            int low[SF_LOAD_BATCH], pidx[SF_LOAD_BATCH];
            bool in[SF_LOAD_BATCH];
#pragma unroll
            for (int q = 0; q < SF_LOAD_BATCH; q++) {
                const int blk = min(blk0 + q, n_blocks - 1);
                const int u = (blk / bv) * 8 + du, v = (blk - (blk / bv) * bv) * 8 + dv;
                in[q] = (blk0 + q < n_blocks) && u < cols0 && v < rows0;
                const int uc = min(u, cols0 - 1), vc = min(v, rows0 - 1);
                pidx[q] = vc + uc * rows0;
                pz[q] = gld(depth0, pidx[q]);
                px[q] = coord_x(lc0, uc, pz[q]);
                py[q] = coord_y(lc0, vc, pz[q]);
                low[q] = gld(labels, o1 + (vc / 2) + (uc / 2) * rows_km);
            }
            bool act[SF_LOAD_BATCH];
            int start[SF_LOAD_BATCH], lab[SF_LOAD_BATCH];
#pragma unroll
            for (int q = 0; q < SF_LOAD_BATCH; q++) {
                act[q] = in[q] && pz[q] != 0.f;
                start[q] = (low[q] == SF_NC) ? 0 : low[q];
            }
            km_search_n<SF_LOAD_BATCH>(s, start, pz, px, py, act, lab);
#pragma unroll
            for (int q = 0; q < SF_LOAD_BATCH; q++)
                if (in[q]) gst(labels, pidx[q], (uint8_t)(act[q] ? lab[q] : SF_NC));
        }
    }
    if (tid < SF_NC) s.conn[tid] = 1u << tid;
    __syncthreads();
    KM_MARK(PF_KM_LABEL0);

    // ------------------------------------------------------------------ computeRegionConnectivity (K3)
    {
        const int rows0 = a.lrows[0], cols0 = a.lcols[0], n0 = a.ln[0];
        const float dist2_threshold = sqf(0.03f * 120.f / float(rows0));
        for (int base = tid; base < n0; base += SF_NT * SF_LOAD_BATCH) {  // SF_LOAD_BATCH pixels per lane: 6 x 4 loads in flight
            float dz[SF_LOAD_BATCH], dzd[SF_LOAD_BATCH], dzr[SF_LOAD_BATCH];
            int la[SF_LOAD_BATCH], ld[SF_LOAD_BATCH], lr[SF_LOAD_BATCH], uu[SF_LOAD_BATCH], vv[SF_LOAD_BATCH];
            bool in[SF_LOAD_BATCH];
#pragma unroll
            for (int q = 0; q < SF_LOAD_BATCH; q++) {
                const int idx = min(base + q * SF_NT, n0 - 1);
                split_uv(lc0, idx, uu[q], vv[q]);
                in[q] = (base + q * SF_NT < n0) && uu[q] < cols0 - 1 && vv[q] < rows0 - 1;
                const int i1 = in[q] ? idx + 1 : idx, i2 = in[q] ? idx + rows0 : idx;  // the last row / column has no neighbour
                dz[q] = gld(depth0, idx);
                dzd[q] = gld(depth0, i1);
                dzr[q] = gld(depth0, i2);
                la[q] = gld(labels, idx);
                ld[q] = gld(labels, i1);
                lr[q] = gld(labels, i2);
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
    }
    __syncthreads();
    if (tid < SF_NC) st.conn[tid] = s.conn[tid];

    // ------------------------------------------------------------------ createClustersPyramidUsingKMeans (K4)
    // pair_dist holds |kmeans_la - kmeans_lb|^2 from the last km_sort_centres (cent_a == kmeans)
    for (int L = 2; L < a.levels; L++) {
        const int n = a.ln[L], o = a.loff[L];
        const LevelCoord lc = level_coord(a, L);
        for (int idx = tid; idx < n; idx += SF_NT) {
            const float pz = gld(depth, o + idx);
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
            gst(labels, o + idx, (uint8_t)lab);
        }
    }
    __syncthreads();
    KM_MARK(PF_KM_CONN_PYR);
#undef KM_MARK
#undef KM_FINE
}
