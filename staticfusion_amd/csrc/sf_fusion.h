// sf_fusion.h — the surfel map on gfx950 without OpenGL: the fusion half of
// Reconstruction::fuseFrame (reference Reconstruction.cpp:264-311):
//   IndexMap::predictIndices  IndexMap.cpp:117-184     Shaders/index_map.vert / .frag
//   GlobalModel::fuse         GlobalModel.cpp:322-492  Shaders/data.vert / .geom / .frag, update.vert
//   GlobalModel::clean        GlobalModel.cpp:494-601  Shaders/copy_unstable.vert / .geom
//
// What GL does with rasterisation, render targets and transform feedback becomes:
//   * index image: one lane per surfel, ONE 64-bit atomicMin on key = epoch tag | window depth | surfel index into a
//     4 rows x 4 cols key image (in-order GL_LESS: nearest wins, the earlier surfel wins a tie; a later image's tag is
//     smaller, so it overwrites whatever an earlier image left: the 9.8 MB image is never cleared, only its 153 KB of
//     occupancy bits, and readers look at occupied texels only). The reference also
//     renders the surfel's camera-frame position / colour-time / normal-radius into three RGBA32F images (184 MB of
//     render targets at QVGA x 4); here the consumers recompute them from the winning surfel: same operations, same bits.
//   * data association: only pixels with (x, y) % 2 == tick % 2 can emit (data.vert:114), so one lane per CANDIDATE
//     pixel, in the reference's emission order (x outer, y inner). The 3072 x 3072 x 3 "update map" render targets
//     (453 MB cleared per frame in the reference) become one atomicMin per associated pixel on winner[surfel] = the
//     candidate's order index: the first fragment passes the depth test, later ones at the same texel fail it.
//   * merge: one lane per model surfel reads winner[] and the winning candidate's record (update.vert).
//   * clean: transform feedback = ORDERED stream compaction of [model surfels ..., candidates ...] by the
//     copy_unstable test: flags + per-256 block counts, one-workgroup scan of the block counts, ordered scatter
//     (ballot ranks within a wave, LDS prefix over the waves of the block).
// Every float expression repeats the shader's association; no contraction; exp / log are sf_detmath.h's: bit-identical
// to the CPU oracle (oracle/sf_oracle_fusion.cpp states the choices made where GL leaves room).
#pragma once
#include "../../include/sf_detmath.h"
#include "sf_predict.h"

#define SF_FUSE_NONE 0xffffffffu

struct FuseArgs {
    // frame (one stream of the handle)
    const float *depth_metric;    // rows x cols row-major
    const float *depth_filtered;  // column-major
    const uint8_t *color;         // rows x cols x 3
    const float *b_img;           // column-major
    int rows, cols;
    // uniforms
    float pose[16], t_inv[16];
    float cx, cy, fx, fy, camz, camw;  // camz = float(1.0 / double(fx))
    float max_depth, conf_threshold, weighting;
    int time, time_delta;
    // model
    const float *src;   // surfels in (count x 12)
    float *dst;         // surfels out
    int count, capacity;
    unsigned long long *keys;  // 4 rows x 4 cols, column-major
    unsigned long long *occ;   // occupancy bits of the key image: column tu owns occ_words words, bit tv of them = a surfel was drawn there
    int occ_words;             // ceil(4 rows / 64)
    unsigned tag_first, tag_merged;  // epoch tags (top byte of the keys) of the two index images of this fuse
    unsigned *winner;          // [count]
    // candidates: pixels (2 i' + par, 2 j' + par), order index q = j' + i' * cand_rows
    int par, cand_rows, cand_cols, n_cand;
    float *rec;       // n_cand x 12
    unsigned *meta;   // n_cand x 2: update_id, best
    // clean
    unsigned char *flags;  // [count + n_cand]
    int *block_counts;     // [ceil((count + n_cand) / SF_CLEAN_BLOCK)]  (becomes the exclusive offsets)
    int *result;           // [0] count after clean (clamped), [1] unclamped, [2] emitted, [3] associated, [4] merged surfels
    float *out;            // where clean writes the map back (= the buffer src points to)
};
// Every kernel takes a TABLE of FuseArgs and works on entry blockIdx.y: one launch serves the maps of many streams
// (sf_map_fuse_frames); grid.x is sized for the largest map, workgroups beyond a map's own extent leave at once.

__device__ __forceinline__ float gl_minf(float x, float y) { return y < x ? y : x; }
__device__ __forceinline__ float gl_maxf(float x, float y) { return x < y ? y : x; }
__device__ __forceinline__ PV3 xform3(const float *T, PV3 v) {
    return {T[0] * v.x + T[4] * v.y + T[8] * v.z + T[12], T[1] * v.x + T[5] * v.y + T[9] * v.z + T[13], T[2] * v.x + T[6] * v.y + T[10] * v.z + T[14]};
}
__device__ __forceinline__ PV3 rotate3(const float *T, PV3 v) {
    return {T[0] * v.x + T[4] * v.y + T[8] * v.z, T[1] * v.x + T[5] * v.y + T[9] * v.z, T[2] * v.x + T[6] * v.y + T[10] * v.z};
}
__device__ __forceinline__ float plength(PV3 a) { return sqrtf(pdot(a, a)); }
__device__ __forceinline__ int nearest_texel(float u, int size) {  // GL_NEAREST, clamp to edge
    const float t = floorf(u * float(size));
    return t < 0.f ? 0 : (t > float(size - 1) ? size - 1 : int(t));
}
__device__ __forceinline__ float encode_color3(float r, float g, float b) {
    int rgb = int(roundf(r * 255.0f));
    rgb = (rgb << 8) + int(roundf(g * 255.0f));
    rgb = (rgb << 8) + int(roundf(b * 255.0f));
    return float(rgb);
}
__device__ __forceinline__ PV3 decode_color3(float c) {
    const int k = int(c);
    return {float((k >> 16) & 0xFF) / 255.0f, float((k >> 8) & 0xFF) / 255.0f, float(k & 0xFF) / 255.0f};
}
// ---- IndexMap::predictIndices -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sf_index_clear_kernel(const FuseArgs *tab) {
    const FuseArgs &a = tab[blockIdx.y];
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o < (size_t)a.cols * 4 * a.occ_words) a.occ[o] = 0ull;  // the keys themselves stay: the next image's tag beats them
}
// start of a fuse: clear the index image, the update-map winners and the counters in one launch
__global__ __launch_bounds__(256) void sf_fuse_begin_kernel(const FuseArgs *tab) {
    const FuseArgs &a = tab[blockIdx.y];
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o < (size_t)a.cols * 4 * a.occ_words) a.occ[o] = 0ull;
    if (o < (size_t)a.count) a.winner[o] = SF_FUSE_NONE;
    if (o < 8) a.result[o] = 0;
}
// surfels: the buffer the index image is rendered from (src before the merge, dst after it)
__device__ __forceinline__ long long sf_op_or64(long long a, long long b) { return a | b; }
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
    long long s = (long long)v;
    SF_DPP_REDUCE(s, dpp_i64, sf_op_or64)
    const int lo = __builtin_amdgcn_readlane((int)(s & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(s >> 32), 63);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
// index_map.vert + the depth test for surfel s at world position p with last-seen time t_last; every lane of the wave calls it
// (live = false: nothing to draw)
__device__ __forceinline__ void index_splat_lane(const FuseArgs &a, unsigned tag, int s, bool live, PV3 p, float t_last) {
    const int W4 = a.cols * 4, H4 = a.rows * 4;
    bool draw = live;
    int px = 0, py = 0;
    if (draw) {
        const PV3 h = xform3(a.t_inv, p);                                                           // index_map.vert:38
        draw = !(h.z > a.max_depth || h.z < 0.f || float(a.time) - t_last > float(a.time_delta));   // :43-48
        const float camx = a.cx * 4.f, camy = a.cy * 4.f, camz = a.fx * 4.f, camw = a.fy * 4.f;     // IndexMap.cpp:136-139
        const float fcols = float(a.cols) * 4.f, frows = float(a.rows) * 4.f;
        const float ndc_x = ((((camz * h.x) / h.z) + camx) - (fcols * 0.5f)) / (fcols * 0.5f);      // :51-52
        const float ndc_y = ((((camw * h.y) / h.z) + camy) - (frows * 0.5f)) / (frows * 0.5f);
        const float ndc_z = h.z / a.max_depth;                                                      // :57
        draw = draw && (ndc_x >= -1.f && ndc_x <= 1.f && ndc_y >= -1.f && ndc_y <= 1.f && ndc_z >= -1.f && ndc_z <= 1.f);
        const float xw = (ndc_x + 1.f) * (fcols * 0.5f), yw = (ndc_y + 1.f) * (frows * 0.5f);
        const float fx_ = floorf(xw), fy_ = floorf(yw);
        draw = draw && (fx_ >= 0.f && fx_ < float(W4) && fy_ >= 0.f && fy_ < float(H4));
        const float depth = ndc_z * 0.5f + 0.5f;  // in [0.5, 1]: bit patterns 0x3f000000 .. 0x3f800000, 24 bits after the offset
        draw = draw && depth < 1.0f;              // GL_LESS against the cleared depth buffer (1.0): a surfel AT maxDepth is not drawn
        if (draw) {
            px = int(fx_);
            py = int(fy_);
            const unsigned long long key = ((unsigned long long)tag << 56) | ((unsigned long long)(__float_as_uint(depth) - 0x3f000000u) << 32) | (unsigned)s;
            __hip_atomic_fetch_min(as_global(a.keys) + (size_t)px * H4 + py, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // column-major key image
        }
    }
    // occupancy bits: the lanes of a wave are neighbours in the map's point order, a dozen of them share one 64-bit word of a
    // column -- OR their bits on the DPP network and let one lane per distinct word do the atomic
    const int word = draw ? px * a.occ_words + (py >> 6) : -1;
    const unsigned long long bit = draw ? 1ull << (py & 63) : 0ull;
    const int lane = threadIdx.x & 63;
    unsigned long long rem = __ballot(draw);
    while (rem) {
        const int src = __ffsll((long long)rem) - 1;
        const int w = __builtin_amdgcn_readlane(word, src);
        const bool mine = word == w;
        const unsigned long long bits = wave_or_u64(mine ? bit : 0ull);
        if (lane == src) __hip_atomic_fetch_or(as_global(a.occ) + w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rem &= ~__ballot(mine);
    }
}
// the index image of the model as it stands (a.src): first predictIndices
__global__ __launch_bounds__(256) void sf_index_splat_kernel(const FuseArgs *tab) {
    const FuseArgs &a = tab[blockIdx.y];
    if ((int)blockIdx.x * 256 >= a.count) return;  // a workgroup of a larger map of the batch
    const int s = blockIdx.x * 256 + threadIdx.x;
    const bool live = s < a.count;
    const auto q = as_global(a.src) + (size_t)(live ? s : 0) * 12;  // typed global pointers: global_load, not flat_load
    index_splat_lane(a, a.tag_first, s, live, PV3{q[0], q[1], q[2]}, q[7]);
}
__global__ __launch_bounds__(256) void sf_index_export_kernel(const unsigned long long *keys, const unsigned long long *occ, int occ_words, unsigned *out,
                                                              int W4, int H4) {  // -> row-major
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= (size_t)W4 * H4) return;
    const int tv = (int)(o / W4), tu = (int)(o - (size_t)tv * W4);
    const bool occupied = (occ[(size_t)tu * occ_words + (tv >> 6)] >> (tv & 63)) & 1ull;
    out[o] = occupied ? (unsigned)(keys[(size_t)tu * H4 + tv] & 0xffffffffull) : 0u;
}

// what the index image's other three textures hold for surfel idx (index_map.vert:56-59)
struct IndexTexelD {
    PV3 pos;
    float conf, t_init, t_last;
    PV3 normal;
};
template <class P>
__device__ __forceinline__ IndexTexelD index_texel_pos(const FuseArgs &a, P surfels, unsigned idx) {
    const auto q = surfels + (size_t)idx * 12;
    IndexTexelD t;
    t.pos = xform3(a.t_inv, PV3{q[0], q[1], q[2]});
    t.conf = q[3];
    t.t_init = q[6];
    t.t_last = q[7];
    return t;
}
template <class P>
__device__ __forceinline__ PV3 index_texel_normal(const FuseArgs &a, P surfels, unsigned idx) {
    const auto q = surfels + (size_t)idx * 12;
    return pnormalize(rotate3(a.t_inv, PV3{q[8], q[9], q[10]}));
}

// ---- the association windows (data.vert:133-135, copy_unstable.vert:62-64) ----------------------------------------
// Both shaders step a texture coordinate in HALF texels by repeated float addition: 17 x 17 samples over about 9 x 9 texels
// of the 4x index image, of which ~6 % hold a surfel. The kernels repeat the float sequence of each axis once (so that the
// texels visited and how often are exactly the shader's), then look only at OCCUPIED texels: the index splat keeps one
// occupancy bit per texel (column-major like the keys), a column of the window is one or two 64-bit loads, and only set
// bits cost a key load and a surfel gather. Texels of one axis are consecutive integers (the coordinate advances half a
// texel per step), so a run list is (first texel, count <= 12, 5-bit multiplicities <= 17).
struct AxisRuns {
    int lo, n;                 // texels lo .. lo + n - 1
    unsigned long long mult;   // 5-bit field k: how many samples fell on texel lo + k
};
__device__ __forceinline__ AxisRuns axis_runs(float centre, float reach, float step, int size) {
    AxisRuns r{0, 0, 0ull};
    int prev = -1;
    for (float w = centre - reach; w < centre + reach; w += step) {  // the shader's loop, verbatim
        const int t = nearest_texel(w, size);
        if (t != prev) {
            if (r.n == 0) r.lo = t;
            r.n = t - r.lo + 1;
            prev = t;
        }
        r.mult += 1ull << (5 * (t - r.lo));
    }
    return r;
}
// occupancy bits of texels v.lo .. v.lo + v.n - 1 of column tu (bit k = texel v.lo + k)
__device__ __forceinline__ unsigned occupancy_bits(const FuseArgs &a, int tu, const AxisRuns &v) {
    const auto col = as_global(a.occ) + (size_t)tu * a.occ_words;
    const int w0 = v.lo >> 6, b0 = v.lo & 63, w1 = (v.lo + v.n - 1) >> 6;
    unsigned long long bits = col[w0] >> b0;
    if (w1 != w0) bits |= col[w1] << (64 - b0);  // b0 > 0 here
    return (unsigned)bits & ((1u << v.n) - 1u);
}

// ---- GlobalModel::fuse, data association (data.vert) --------------------------------------------------------------
__global__ __launch_bounds__(64) void sf_fuse_data_kernel(const FuseArgs *tab) {  // 64: 19 200 candidates at QVGA are 300 single-wave workgroups
    const FuseArgs &a = tab[blockIdx.y];
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= a.n_cand) return;
    const int ic = q / a.cand_rows, jc = q - ic * a.cand_rows;
    const int i = 2 * ic + a.par, j = 2 * jc + a.par;
    const int rows = a.rows, cols = a.cols;
    const float W = float(cols), H = float(rows);
    const float tx = float(double(float(i) / W) + 1.0 / double(2 * W));  // GlobalModel.cpp:81-82
    const float ty = float(double(float(j) / H) + 1.0 / double(2 * H));
    const float x = tx * W, y = ty * H;                                   // data.vert:79-80
    auto Draw = [&](int ii, int jj) { return a.depth_metric[(size_t)min(max(jj, 0), rows - 1) * cols + min(max(ii, 0), cols - 1)]; };
    auto Dfil = [&](int ii, int jj) { return a.depth_filtered[min(max(jj, 0), rows - 1) + (size_t)min(max(ii, 0), cols - 1) * rows]; };
    auto vertex = [&](float z, float xx, float yy) { return PV3{(xx - a.cx) * z * a.camz, (yy - a.cy) * z * a.camw, z}; };
    const PV3 vPosLocal = vertex(Draw(i, j), x, y);   // :83
    const PV3 world = xform3(a.pose, vPosLocal);      // :84
    const PV3 vf = vertex(Dfil(i, j), x, y);          // :87
    const float probIsStatic = a.b_img[j + (size_t)i * rows];
    const uint8_t *c = a.color + ((size_t)j * cols + i) * 3;
    const float color = encode_color3(float(c[0]) / 255.0f, float(c[1]) / 255.0f, float(c[2]) / 255.0f);
    const PV3 xf = vertex(Dfil(i + 1, j), x + 1.f, y), xb = vertex(Dfil(i - 1, j), x - 1.f, y);
    const PV3 yf = vertex(Dfil(i, j + 1), x, y + 1.f), yb = vertex(Dfil(i, j - 1), x, y - 1.f);
    auto half_sum = [](PV3 p, PV3 r) { return PV3{(p.x + r.x) / 2.f, (p.y + r.y) / 2.f, (p.z + r.z) / 2.f}; };
    const PV3 vNormLocal = pnormalize(pcross(psub(half_sum(xb, vf), half_sum(xf, vf)), psub(half_sum(yb, vf), half_sum(yf, vf))));
    const float meanFocal = ((1.0f / fabsf(a.camz)) + (1.0f / fabsf(a.camw))) / 2.0f;
    const float radius0 = (vf.z / meanFocal) * 1.41421356237f;
    const float radius = gl_minf(2.0f * radius0, radius0 / fabsf(vNormLocal.z));
    const PV3 nWorld = rotate3(a.pose, vNormLocal);
    const float pcx = x - a.cx, pcy = y - a.cy;
    const float radialDist = sqrtf(pcx * pcx + pcy * pcy) / 200.0f;
    const float radialConf = sf_exp_neg((radialDist * radialDist) / (2.0f * 0.72f));
    float conf = gl_minf(probIsStatic, gl_minf(a.weighting, radialConf));
    float t_last = 0.f;
    unsigned update_id = 0, best = 0;
    const float ftime = float(a.time);
    const bool neighbours = !(Draw(i - 1, j) == 0.f) && !(Draw(i, j - 1) == 0.f) && !(Draw(i + 1, j) == 0.f) && !(Draw(i, j + 1) == 0.f);
    // the (x, y) % 2 == time % 2 test of :114 is what makes this pixel a candidate
    if (neighbours && vPosLocal.z > 0.f && vPosLocal.z <= a.max_depth) {
        int counter = 0;
        const float scale = 4.0f;
        const float indexXStep = (1.0f / (W * scale)) * 0.5f;
        const float indexYStep = (1.0f / (H * scale)) * 0.5f;
        float bestDist = 1000.f;
        const float windowMultiplier = 2.f;
        const float xl = (x - a.cx) * a.camz, yl = (y - a.cy) * a.camw;
        const float lambda = sqrtf(xl * xl + yl * yl + 1.f);
        const PV3 ray{xl, yl, 1.f};
        const float ray_len = plength(ray);
        const float nl_len = plength(vNormLocal);
        // a repeated texel offers the same candidate at the same distance and only a STRICTLY nearer one replaces the best:
        // every distinct texel is looked at once, in the shader's order (u outer, v inner, ascending)
        const int W4 = cols * 4, H4 = rows * 4;
        const AxisRuns ur = axis_runs(tx, scale * indexXStep * windowMultiplier, indexXStep, W4);
        const AxisRuns vr = axis_runs(ty, scale * indexYStep * windowMultiplier, indexYStep, H4);
        const auto g_keys = as_global((const unsigned long long *)a.keys);
        const auto g_src = as_global(a.src);
        for (int cu = 0; cu < ur.n; cu++) {
            const int tu = ur.lo + cu;
            unsigned bits = occupancy_bits(a, tu, vr);
            while (bits) {
                const int k = __ffs((int)bits) - 1;
                bits &= bits - 1u;
                const unsigned current = (unsigned)(g_keys[(size_t)tu * H4 + vr.lo + k] & 0xffffffffull);  // occupied: a key of this image
                if (current > 0U) {
                    const IndexTexelD t = index_texel_pos(a, g_src, current);
                    if (fabsf((t.pos.z * lambda) - (vPosLocal.z * lambda)) < 0.05f) {
                        const float dist = plength(pcross(ray, t.pos)) / ray_len;
                        if (dist < bestDist) {
                            const PV3 tn = index_texel_normal(a, g_src, current);
                            bool angle_ok = fabsf(tn.z) < 0.75f;
                            if (!angle_ok) {
                                const float cs = pdot(tn, vNormLocal) / (plength(tn) * nl_len);
                                angle_ok = cs > 0.87758256189f && cs <= 1.0f;
                            }
                            if (angle_ok) {
                                counter++;
                                bestDist = dist;
                                best = current;
                            }
                        }
                    }
                }
            }
        }
        if (counter > 0) {
            update_id = 1;
            t_last = -1.f;
        } else {
            update_id = 2;
            t_last = -2.f;
            conf = 0.f;
            if (probIsStatic > 0.5f) conf = 0.08f;
        }
    }
    float *o = a.rec + (size_t)q * 12;
    o[0] = world.x; o[1] = world.y; o[2] = world.z; o[3] = conf;
    o[4] = color; o[5] = 1.0f; o[6] = ftime; o[7] = t_last;
    o[8] = nWorld.x; o[9] = nWorld.y; o[10] = nWorld.z; o[11] = radius;
    a.meta[(size_t)q * 2] = update_id;
    a.meta[(size_t)q * 2 + 1] = best;
    if (update_id == 1) atomicMin(a.winner + best, (unsigned)q);
    // counters (order-free sums)
    const unsigned long long em = __ballot(update_id > 0), as = __ballot(update_id == 1);
    if ((threadIdx.x & 63) == 0) {
        if (em) atomicAdd(a.result + 2, (int)__popcll(em));
        if (as) atomicAdd(a.result + 3, (int)__popcll(as));
    }
}

// ---- GlobalModel::fuse, merge (update.vert): src -> dst for every model surfel ----------------------------------
// ... and, in the same pass, the index image of the MERGED model (second predictIndices, Reconstruction.cpp:300): the key image
// was cleared by sf_index_clear_kernel after the association read it
__global__ __launch_bounds__(256) void sf_fuse_update_kernel(const FuseArgs *tab) {
    const FuseArgs &a = tab[blockIdx.y];
    if ((int)blockIdx.x * 256 >= a.count) return;  // a workgroup of a larger map of the batch
    const int s = blockIdx.x * 256 + threadIdx.x;
    const bool live = s < a.count;
    const auto q = as_global(a.src) + (size_t)(live ? s : 0) * 12;
    const unsigned w_ = live ? as_global((const unsigned *)a.winner)[s] : SF_FUSE_NONE;
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = q[k];
    if (w_ != SF_FUSE_NONE) {
        const auto d = as_global((const float *)a.rec) + (size_t)w_ * 12;
        float c_k = v[3];
        float aa = d[3];
        const float hist = v[5];
        const float max_val = 0.99f, min_val = 0.01f;
        aa = gl_maxf(min_val, gl_minf(0.53f, 2.f * aa * aa));
        c_k = gl_maxf(min_val, gl_minf(c_k, max_val));
        float ltm = sf_log_det(1.0f / (1.0f - c_k) - 1.0f);
        ltm = ltm + sf_log_det(aa / (1.0f - aa));
        const float c_k1 = 1.0f - (1.0f / (1.0f + sf_exp_det(ltm)));
        if (d[11] < (1.0f + 0.5f) * v[11]) {
            const float w = hist * c_k, den = hist * c_k + aa;
            const PV3 oldCol = decode_color3(v[4]), newCol = decode_color3(d[4]);
            const PV3 n = pnormalize(PV3{((w * v[8]) + (aa * d[8])) / den, ((w * v[9]) + (aa * d[9])) / den, ((w * v[10]) + (aa * d[10])) / den});
            v[11] = ((w * v[11]) + (aa * d[11])) / den;
            v[0] = ((w * v[0]) + (aa * d[0])) / den;
            v[1] = ((w * v[1]) + (aa * d[1])) / den;
            v[2] = ((w * v[2]) + (aa * d[2])) / den;
            v[4] = encode_color3(((w * oldCol.x) + (aa * newCol.x)) / den, ((w * oldCol.y) + (aa * newCol.y)) / den,
                                 ((w * oldCol.z) + (aa * newCol.z)) / den);
            v[8] = n.x; v[9] = n.y; v[10] = n.z;
        }
        v[3] = c_k1;
        v[5] = hist + 1.0f;
        v[7] = float(a.time);
    }
    if (live) {
        const auto o = as_global(a.dst) + (size_t)s * 12;
#pragma unroll
        for (int k = 0; k < 12; k++) o[k] = v[k];
    }
    const unsigned long long mg = __ballot(w_ != SF_FUSE_NONE);
    if ((threadIdx.x & 63) == 0 && mg) atomicAdd(a.result + 4, (int)__popcll(mg));
    index_splat_lane(a, a.tag_merged, s, live, PV3{v[0], v[1], v[2]}, v[7]);
}

// ---- GlobalModel::clean (copy_unstable.vert): element e of [merged model (in dst) ..., candidates ...] ------------
__device__ __forceinline__ gptr<const float> clean_element(const FuseArgs &a, int e, bool &present) {
    if (e < a.count) {
        present = true;
        return as_global((const float *)a.dst) + (size_t)e * 12;
    }
    const int q = e - a.count;
    present = as_global((const unsigned *)a.meta)[(size_t)q * 2] > 0;  // data.geom emits only updateId > 0
    return as_global((const float *)a.rec) + (size_t)q * 12;
}
#define SF_CLEAN_BLOCK 256  // compaction granule: small, so that a QVGA map (93 k elements) still covers the 256 CUs
__global__ __launch_bounds__(SF_CLEAN_BLOCK) void sf_clean_flag_kernel(const FuseArgs *tab) {
    __shared__ int block_total;
    const FuseArgs &a = tab[blockIdx.y];
    const int e = blockIdx.x * SF_CLEAN_BLOCK + threadIdx.x;
    const int n = a.count + a.n_cand;
    if ((int)blockIdx.x * SF_CLEAN_BLOCK >= n) return;  // a workgroup of a larger map of the batch
    if (threadIdx.x == 0) block_total = 0;
    __syncthreads();
    bool keep = false;
    if (e < n) {
        bool present;
        const auto q = clean_element(a, e, present);
        if (present) {
            int test = 1;
            const PV3 localPos = xform3(a.t_inv, PV3{q[0], q[1], q[2]});
            const float W = float(a.cols), H = float(a.rows);
            const float x = ((a.fx * localPos.x) / localPos.z) + a.cx;
            const float y = ((a.fy * localPos.y) / localPos.z) + a.cy;
            const float scale = 4.0f;
            const float indexXStep = (1.0f / (W * scale)) * 0.5f;
            const float indexYStep = (1.0f / (H * scale)) * 0.5f;
            const float windowMultiplier = 2.f;
            int count = 0, zCount = 0;
            const float ftime = float(a.time), fdelta = float(a.time_delta);
            const float conf_v = q[3], t_init_v = q[6], rad_v = q[11];
            float t_last_v = q[7];
            // :108-116 decide without the window for merged candidates (w == -1), zero-confidence points and stale unstable
            // surfels: whatever the two counts say, the vertex is dropped (unless the time-window override keeps it) -- skip the scan
            const float t_fixed = t_last_v == -2.f ? ftime : t_last_v;
            const bool dropped_anyway = (t_fixed == -1.f || ((ftime - t_fixed) > 10.f && conf_v < 0.5f)) || (conf_v == 0.0f);
            const bool kept_anyway = t_fixed > 0.f && ftime - t_fixed > fdelta;
            if (!dropped_anyway && !kept_anyway && ftime - t_last_v < fdelta && localPos.z > 0.f && x > 0.f && y > 0.f && x < W && y < H) {
                // a texel sampled m_u x m_v times counts m_u x m_v times: each occupied texel is evaluated once with that weight
                const int W4 = a.cols * 4, H4 = a.rows * 4;
                const AxisRuns ur = axis_runs(x / W, scale * indexXStep * windowMultiplier, indexXStep, W4);
                const AxisRuns vr = axis_runs(y / H, scale * indexYStep * windowMultiplier, indexYStep, H4);
                const auto g_keys = as_global((const unsigned long long *)a.keys);
                const auto g_dst = as_global((const float *)a.dst);
                for (int cu = 0; cu < ur.n; cu++) {
                    const int tu = ur.lo + cu, mu = (int)((ur.mult >> (5 * cu)) & 31ull);
                    unsigned bits = occupancy_bits(a, tu, vr);
                    while (bits) {
                        const int k = __ffs((int)bits) - 1;
                        bits &= bits - 1u;
                        const int mult = mu * (int)((vr.mult >> (5 * k)) & 31ull);
                        const unsigned current = (unsigned)(g_keys[(size_t)tu * H4 + vr.lo + k] & 0xffffffffull);  // occupied: a key of this image
                        if (current > 0U) {
                            const IndexTexelD t = index_texel_pos(a, g_dst, current);
                            const float dx = t.pos.x - localPos.x, dy = t.pos.y - localPos.y;
                            if (t.t_init < t_init_v && t.conf > a.conf_threshold && t.pos.z > localPos.z && t.pos.z - localPos.z < 0.01f &&
                                sqrtf(dx * dx + dy * dy) < rad_v * 1.4f)
                                count += mult;
                            if (t.t_last == ftime && t.conf > 0.4f * a.conf_threshold && t.pos.z > localPos.z && t.pos.z - localPos.z > 0.01f) zCount += mult;
                        }
                    }
                }
            }
            if (count > 6 || zCount > 5) test = 0;
            if (t_last_v == -2.f) t_last_v = ftime;
            if ((t_last_v == -1.f || ((ftime - t_last_v) > 10.f && conf_v < 0.5f)) || (conf_v == 0.0f)) test = 0;
            if (t_last_v > 0.f && ftime - t_last_v > fdelta) test = 1;
            keep = test > 0;
        }
        a.flags[e] = keep ? 1 : 0;
    }
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&block_total, (int)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) a.block_counts[blockIdx.x] = block_total;
}
// exclusive scan of the block counts in place (one workgroup), totals into result[0..1]
__global__ __launch_bounds__(1024) void sf_clean_scan_kernel(const FuseArgs *tab) {
    __shared__ int wsum[16];
    __shared__ int base;
    const FuseArgs &a = tab[blockIdx.x];  // one workgroup per map
    const int n_blocks = (a.count + a.n_cand + SF_CLEAN_BLOCK - 1) / SF_CLEAN_BLOCK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_blocks; c0 += 1024) {
        const int idx = c0 + tid;
        const int v = idx < n_blocks ? a.block_counts[idx] : 0;
        int incl = v;  // inclusive scan within the wave (DPP-free, shuffle based: not a hot loop)
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; w++) off += wsum[w];
        if (idx < n_blocks) a.block_counts[idx] = off + incl - v;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; w++) t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.result[1] = base;
        a.result[0] = min(base, a.capacity);  // transform feedback stops at the end of its buffer
    }
}
// ordered scatter into src (the buffer the merged model was NOT written to becomes the map again)
__global__ __launch_bounds__(SF_CLEAN_BLOCK) void sf_clean_write_kernel(const FuseArgs *tab) {
    __shared__ int wcount[SF_CLEAN_BLOCK / 64];
    const FuseArgs &a = tab[blockIdx.y];
    float *out = a.out;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = blockIdx.x * SF_CLEAN_BLOCK + tid;
    const int n = a.count + a.n_cand;
    if ((int)blockIdx.x * SF_CLEAN_BLOCK >= n) return;
    const bool keep = e < n && a.flags[e] != 0;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wcount[wave] = (int)__popcll(m);
    __syncthreads();
    if (!keep) return;
    int pos = a.block_counts[blockIdx.x];
    for (int w = 0; w < wave; w++) pos += wcount[w];
    pos += (int)__popcll(m & ((1ull << lane) - 1ull));
    if (pos >= a.capacity) return;
    bool present;
    const auto q = clean_element(a, e, present);
    const auto o4 = (gptr<vfloat4>)(as_global(out) + (size_t)pos * 12);  // 48-byte records, 16-byte aligned: three 16-byte moves
    const auto q4 = (gptr<const vfloat4>)q;
    const vfloat4 v0 = q4[0], v2 = q4[2];
    vfloat4 v1 = q4[1];
    if (v1.w == -2.f) v1.w = float(a.time);  // copy_unstable.vert:101-104
    o4[0] = v0;
    o4[1] = v1;
    o4[2] = v2;
}
