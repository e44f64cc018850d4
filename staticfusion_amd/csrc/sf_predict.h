// sf_predict.h — frame-to-model prediction on gfx950 without OpenGL:
// Reconstruction::getPredictedImages (reference Reconstruction.cpp:628-720) = two point-sprite renderings of
// the surfel model (IndexMap::combinedPredict IndexMap.cpp:221-300; Shaders/splat.vert, combo_splat.frag),
// the density test (Resize + denseEnough :218-233), the fill-in passes (Shaders/FillIn.cpp, fill_*.frag) and
// the depth extraction (extract_depth.frag).
//
// A GL rasteriser resolves visibility with an ordered depth test; here every surfel is one lane that walks
// the pixels of its sprite and does ONE 64-bit atomicMin per surviving fragment on
//     key = (bits of gl_FragDepth) << 32 | surfel index
// (gl_FragDepth is a positive float: its bit pattern is monotonic; the index makes the earlier surfel win a
// tie, as in-order GL_LESS does). A second kernel re-evaluates the winning fragment per pixel and runs the
// per-pixel fill-in / extraction. Both confidence levels are rendered by the same pass (two key buffers).
// Every float expression repeats the shader's association; no contraction: bit-identical to the CPU oracle.
#pragma once
#include "sf_device_common.h"

struct PredictArgs {
    const float *surfels;  // count x 12
    int count;
    float t_inv[16];       // column-major
    float cx, cy, fx, fy, max_depth, conf_low, conf_high, extract_max_depth;
    int time, max_time, time_delta;
    int rows, cols;
    unsigned long long *key_low, *key_high;  // rows x cols, column-major
    int *dense_count;                        // [1]
    const uint16_t *filtered_mm;             // rows x cols row-major
    const uint8_t *color;                    // rows x cols x 3
    const float *b_img;                      // column-major
    float *depth_pred, *inten_pred;          // column-major
    const vfloat4 *rays;                     // per pixel, column-major: normalize((x + 0.5 - cx) / fx, (y + 0.5 - cy) / fy, 1)
};

#define SF_PRED_EMPTY 0xffffffffffffffffull

struct PV3 { float x, y, z; };
__device__ __forceinline__ float pdot(PV3 a, PV3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ PV3 padd(PV3 a, PV3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ PV3 psub(PV3 a, PV3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ PV3 pmul(PV3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ PV3 pnormalize(PV3 a) {
    const float n = sqrtf(pdot(a, a));
    return {a.x / n, a.y / n, a.z / n};
}
__device__ __forceinline__ PV3 pcross(PV3 a, PV3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

// camera-frame position, normal and radius of surfel s (splat.vert:55,68); false if culled by :57
__device__ __forceinline__ bool surfel_to_camera(const PredictArgs &a, int s, float conf_threshold, PV3 &h, PV3 &n, float &rad) {
    const auto q = as_global(a.surfels) + (size_t)s * 12;  // typed global pointer: global_load, not flat_load
    const float *T = a.t_inv;
    const PV3 vp{q[0], q[1], q[2]};
    const float conf = q[3], tlast = q[7];
    h = PV3{T[0] * vp.x + T[4] * vp.y + T[8] * vp.z + T[12], T[1] * vp.x + T[5] * vp.y + T[9] * vp.z + T[13],
            T[2] * vp.x + T[6] * vp.y + T[10] * vp.z + T[14]};
    if (h.z > a.max_depth || h.z < 0.4f || conf < conf_threshold || float(a.time) - tlast > float(a.time_delta) || tlast > float(a.max_time))
        return false;
    const PV3 nin{q[8], q[9], q[10]};
    n = pnormalize(PV3{T[0] * nin.x + T[4] * nin.y + T[8] * nin.z, T[1] * nin.x + T[5] * nin.y + T[9] * nin.z,
                       T[2] * nin.x + T[6] * nin.y + T[10] * nin.z});
    rad = q[11];
    return true;
}

// combo_splat.frag:37-49,63 for the fragment at pixel (i, j); false = discard
__device__ __forceinline__ bool surfel_fragment(const PredictArgs &a, PV3 h, PV3 n, float rad, int i, int j, float &z, float &depth) {
    // the view ray of the pixel (combo_splat.frag:37-39) does not depend on the surfel: five divisions and a square root per
    // fragment become one 16-byte load from a table built with the same expression (sf_predict_rays_kernel)
    const vfloat4 lr = as_global(a.rays)[j + (size_t)i * a.rows];
    const PV3 l{lr.x, lr.y, lr.z};
    const PV3 corrected = pmul(l, pdot(h, n) / pdot(l, n));
    const PV3 diff = psub(corrected, h);
    if (pdot(diff, diff) > rad * rad) return false;
    z = corrected.z;
    depth = (corrected.z / (2.f * a.max_depth)) + 0.5f;
    return depth >= 0.f && depth < 1.f;  // depth range clip, and GL_LESS against the cleared depth 1.0: a fragment AT 1.0 fails
}

__global__ __launch_bounds__(256) void sf_predict_rays_kernel(vfloat4 *rays, int rows, int cols, float cx, float cy, float fx, float fy) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // j + i * rows
    if (idx >= rows * cols) return;
    const int i = idx / rows, j = idx - i * rows;
    const float fx_ = float(i) + 0.5f, fy_ = float(j) + 0.5f;  // gl_FragCoord
    const PV3 l = pnormalize(PV3{(fx_ - cx) / fx, (fy_ - cy) / fy, 1.0f});
    rays[idx] = vfloat4{l.x, l.y, l.z, 0.f};
}
// Every kernel takes a TABLE of PredictArgs and works on entry blockIdx.y (one launch renders the maps of many streams).
__global__ __launch_bounds__(256) void sf_predict_clear_kernel(const PredictArgs *tab) {
    const PredictArgs &a = tab[blockIdx.y];
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o < a.rows * a.cols) {
        a.key_low[o] = SF_PRED_EMPTY;
        a.key_high[o] = SF_PRED_EMPTY;
    }
    if (o == 0) *a.dense_count = 0;
}

// One lane per surfel: sprite extent (splat.vert:64-85), then one atomicMin per surviving fragment. A pixel is covered by
// the sprites of ~20 surfels and the L2 atomics are what the pass costs (46 us per QVGA map with them, 3 us without), so
// a workgroup first resolves its own fragments in LDS: the 512 surfels of a workgroup are neighbours in the map's point
// order (x outer, y inner: about two image columns), their sprites fit a tile of ~2000 pixels, ds_min_u64 on the tile is
// cheap, and only the tile's winners go to the key images (4-5x fewer global atomics: 46 -> 19 us per map in a batch,
// 0.105 -> 0.089 ms for one map; 256 / 1024 surfels per workgroup measured slower). A workgroup whose bounding box does
// not fit the tile (an incoherent stretch of the map) falls back to global atomics per fragment. min is associative and
// commutative: the key images end up the same either way. (A buffer with 16 surfels on every pixel -- tools/predict_bench.py
// -- is slower this way, 0.55 instead of 0.30 ms for 1.2 M surfels: GlobalModel::clean does not let a map get that dense.)
#define SF_SPLAT_NT 512
#define SF_SPLAT_TILE 3072  // pixels; 2 targets x 8 B = 48 KB of LDS
__global__ __launch_bounds__(SF_SPLAT_NT) void sf_predict_splat_kernel(const PredictArgs *tab) {
    __shared__ unsigned long long tile[2][SF_SPLAT_TILE];
    __shared__ int bb[4];
    const PredictArgs &a = tab[blockIdx.y];
    if ((int)blockIdx.x * SF_SPLAT_NT >= a.count) return;  // a workgroup of a larger map of the batch
    const int tid = threadIdx.x;
    const int s = blockIdx.x * SF_SPLAT_NT + tid;
    PV3 h{0.f, 0.f, 1.f}, n{0.f, 0.f, 1.f};
    float rad = 0.f;
    bool valid = s < a.count && surfel_to_camera(a, s, a.conf_low, h, n, rad);  // conf_low <= conf_high: the low pass is the superset
    bool high = false;
    int i0 = 0, i1 = -1, j0 = 0, j1 = -1;
    if (valid) {
        high = !(as_global(a.surfels)[(size_t)s * 12 + 3] < a.conf_high);
        const float fcols = float(a.cols), frows = float(a.rows);
        const float ndc_x = ((((a.fx * h.x) / h.z) + a.cx) - (fcols * 0.5f)) / (fcols * 0.5f);
        const float ndc_y = ((((a.fy * h.y) / h.z) + a.cy) - (frows * 0.5f)) / (frows * 0.5f);
        valid = (ndc_x >= -1.f && ndc_x <= 1.f && ndc_y >= -1.f && ndc_y <= 1.f);
        if (valid) {
            const float xw = (ndc_x + 1.f) * (fcols * 0.5f), yw = (ndc_y + 1.f) * (frows * 0.5f);
            const PV3 x1 = pmul(pmul(pnormalize(PV3{n.y - n.z, -n.x, n.x}), rad), 1.41421356f);
            const PV3 y1 = pcross(n, x1);
            const PV3 c1 = padd(h, x1), c2 = padd(h, y1), c3 = psub(h, y1), c4 = psub(h, x1);
            const float p1x = ((a.fx * c1.x) / c1.z) + a.cx, p1y = ((a.fy * c1.y) / c1.z) + a.cy;
            const float p2x = ((a.fx * c2.x) / c2.z) + a.cx, p2y = ((a.fy * c2.y) / c2.z) + a.cy;
            const float p3x = ((a.fx * c3.x) / c3.z) + a.cx, p3y = ((a.fy * c3.y) / c3.z) + a.cy;
            const float p4x = ((a.fx * c4.x) / c4.z) + a.cx, p4y = ((a.fy * c4.y) / c4.z) + a.cy;
            const float xDiff = fabsf(fmaxf(p1x, fmaxf(p2x, fmaxf(p3x, p4x))) - fminf(p1x, fminf(p2x, fminf(p3x, p4x))));
            const float yDiff = fabsf(fmaxf(p1y, fmaxf(p2y, fmaxf(p3y, p4y))) - fminf(p1y, fminf(p2y, fminf(p3y, p4y))));
            const float size = fmaxf(0.f, fmaxf(xDiff, yDiff));
            valid = size > 0.f;
            if (valid) {
                const float half = size * 0.5f;
                i0 = max(0, (int)ceilf(xw - half - 0.5f)); i1 = min(a.cols - 1, (int)floorf(xw + half - 0.5f));
                j0 = max(0, (int)ceilf(yw - half - 0.5f)); j1 = min(a.rows - 1, (int)floorf(yw + half - 0.5f));
                // The sprite is a SQUARE of the longer extent (splat.vert:85), but a fragment survives only inside the disc
                // (combo_splat.frag:47), and the disc lies inside the diamond whose corners were just projected: pixels outside
                // their bounding rectangle (+ 1 pixel against rounding at the rim) would be discarded anyway -- skip them.
                const float xlo = fminf(p1x, fminf(p2x, fminf(p3x, p4x))), xhi = fmaxf(p1x, fmaxf(p2x, fmaxf(p3x, p4x)));
                const float ylo = fminf(p1y, fminf(p2y, fminf(p3y, p4y))), yhi = fmaxf(p1y, fmaxf(p2y, fmaxf(p3y, p4y)));
                i0 = max(i0, (int)floorf(xlo - 1.5f)); i1 = min(i1, (int)ceilf(xhi + 0.5f));
                j0 = max(j0, (int)floorf(ylo - 1.5f)); j1 = min(j1, (int)ceilf(yhi + 0.5f));
                valid = i0 <= i1 && j0 <= j1;
            }
        }
    }
    // bounding box of the workgroup's sprites
    if (tid == 0) {
        bb[0] = bb[1] = 0x7fffffff;
        bb[2] = bb[3] = -1;
    }
    __syncthreads();
    if (valid) {
        atomicMin(&bb[0], i0);
        atomicMin(&bb[1], j0);
        atomicMax(&bb[2], i1);
        atomicMax(&bb[3], j1);
    }
    __syncthreads();
    const int bi0 = bb[0], bj0 = bb[1], bw = bb[2] - bi0 + 1, bh = bb[3] - bj0 + 1;
    if (bw <= 0 || bh <= 0) return;  // nothing to draw (uniform: read from LDS after the barrier)
    const bool tiled = bw * bh <= SF_SPLAT_TILE;
    // tile layout along the longer side of the box: that is the direction in which neighbouring lanes' surfels follow each
    // other (a map in the reference's point order is a tall box, a row-ordered buffer a wide one): consecutive LDS words
    const bool wide = bw > bh;
    if (tiled) {
        for (int q = tid; q < bw * bh; q += SF_SPLAT_NT) {
            tile[0][q] = SF_PRED_EMPTY;
            tile[1][q] = SF_PRED_EMPTY;
        }
        __syncthreads();
    }
    if (valid)
        for (int i = i0; i <= i1; i++)       // key images and tile are column-major: neighbouring lanes (vertical
            for (int j = j0; j <= j1; j++) {  // neighbours) and the inner loop touch neighbouring words
                float z, depth;
                if (!surfel_fragment(a, h, n, rad, i, j, z, depth)) continue;
                const unsigned long long key = ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned)s;
                if (tiled) {
                    const int t = wide ? (j - bj0) * bw + (i - bi0) : (i - bi0) * bh + (j - bj0);
                    atomicMin(&tile[0][t], key);
                    if (high) atomicMin(&tile[1][t], key);
                } else {
                    const int o = j + i * a.rows;
                    __hip_atomic_fetch_min(as_global(a.key_low) + o, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (high) __hip_atomic_fetch_min(as_global(a.key_high) + o, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
    if (!tiled) return;
    __syncthreads();
    for (int q = tid; q < bw * bh; q += SF_SPLAT_NT) {
        const int i = bi0 + (wide ? q % bw : q / bh), j = bj0 + (wide ? q / bw : q % bh);
        const int o = j + i * a.rows;
        const unsigned long long kl = tile[0][q], kh = tile[1][q];
        if (kl != SF_PRED_EMPTY) __hip_atomic_fetch_min(as_global(a.key_low) + o, kl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (kh != SF_PRED_EMPTY) __hip_atomic_fetch_min(as_global(a.key_high) + o, kh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// z and colour bytes of the fragment that won pixel (i, j) in one target (0 / black where nothing was drawn)
__device__ __forceinline__ void resolve_pixel(const PredictArgs &a, unsigned long long key, int i, int j, float &z, int &r, int &g, int &b) {
    z = 0.f;
    r = g = b = 0;
    if (key == SF_PRED_EMPTY) return;
    const int s = (int)(unsigned)(key & 0xffffffffull);
    PV3 h, n;
    float rad, depth;
    surfel_to_camera(a, s, -1.0e30f, h, n, rad);
    surfel_fragment(a, h, n, rad, i, j, z, depth);
    const int c = (int)as_global(a.surfels)[(size_t)s * 12 + 4];  // color.glsl decodeColor; the RGBA8 target holds the bytes
    r = (c >> 16) & 0xFF;
    g = (c >> 8) & 0xFF;
    b = c & 0xFF;
}

// Resize to (cols/40) x (rows/40) + denseEnough: counts the sampled low-confidence pixels with all channels > 0
__global__ __launch_bounds__(64) void sf_predict_dense_kernel(const PredictArgs *tab) {
    const PredictArgs &a = tab[blockIdx.x];  // one wave per map
    const int rw = a.cols / 40, rh = a.rows / 40;
    int sum = 0;
    for (int q = threadIdx.x; q < rw * rh; q += 64) {
        const int j = q / rw, i = q - j * rw;
        const int sx = min(a.cols - 1, (int)(((float(i) + 0.5f) / float(rw)) * float(a.cols)));
        const int sy = min(a.rows - 1, (int)(((float(j) + 0.5f) / float(rh)) * float(a.rows)));
        float z;
        int r, g, b;
        resolve_pixel(a, a.key_low[sy + sx * a.rows], sx, sy, z, r, g, b);
        sum += (r > 0 && g > 0 && b > 0) ? 1 : 0;
    }
    sum = wave_sum_i32(sum);
    if (threadIdx.x == 0) *a.dense_count = sum;
}

// per pixel: resolve both targets, fill-in, extract depth, intensity (lanes along y: the outputs are column-major)
__global__ __launch_bounds__(256) void sf_predict_resolve_kernel(const PredictArgs *tab) {
    const PredictArgs &a = tab[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;  // y + x * rows
    if (idx >= a.rows * a.cols) return;
    const int x = idx / a.rows, y = idx - x * a.rows;
    const int o = y * a.cols + x;
    const int rw = a.cols / 40, rh = a.rows / 40;
    const bool dense = (rw * rh > 0) && (float(*a.dense_count) / float(rh * rw) > 0.25f);
    float zl, zh;
    int rl, gl, bl, rh_, gh, bh;
    resolve_pixel(a, a.key_low[idx], x, y, zl, rl, gl, bl);  // the key images are column-major like the outputs
    resolve_pixel(a, a.key_high[idx], x, y, zh, rh_, gh, bh);
    const bool high_empty = (rh_ + gh + bh) == 0, low_empty = (rl + gl + bl) == 0;
    float z;
    int r, g, b;
    if (!dense) {
        float z1 = zl;
        if (z1 == 0.f) {
            const float zr = float(a.filtered_mm[o]) / 1000.0f;
            z1 = (a.b_img[idx] > 0.6f) ? zr : 0.0f;
        }
        z = (zh == 0.f) ? z1 : zh;
        int r1 = rl, g1 = gl, b1 = bl;
        if (low_empty) {
            r1 = a.color[(size_t)o * 3];
            g1 = a.color[(size_t)o * 3 + 1];
            b1 = a.color[(size_t)o * 3 + 2];
        }
        r = high_empty ? r1 : rh_;
        g = high_empty ? g1 : gh;
        b = high_empty ? b1 : bh;
    } else {
        z = (zh == 0.f) ? zl : zh;
        r = high_empty ? rl : rh_;
        g = high_empty ? gl : gh;
        b = high_empty ? bl : bh;
    }
    a.depth_pred[idx] = (z > a.extract_max_depth || z <= 0.f) ? 0.f : z;
    const float norm_factor = 1.f / 255.f;
    const float fr = float(r) * norm_factor, fg = float(g) * norm_factor, fb = float(b) * norm_factor;
    a.inten_pred[idx] = 0.299f * fr + 0.587f * fg + 0.114f * fb;
}

// ---------------------------------------------------------------------------------------------------------------
//  GlobalModel::initialise (reference GlobalModel.cpp:200-258): the two vertex_feedback passes
//  (Reconstruction.cpp:205-216, Shaders/vertex_feedback.vert/.geom, geometry.glsl:19-41, surfels.glsl:19-35) and
//  init_unstable.vert. One 1024-thread workgroup walks the frame in the reference's point order (x outer, y inner =
//  this repository's column-major index); the two transform-feedback streams are ordered compactions: ballot ranks
//  per wave + an LDS prefix over the 16 waves + a running base per chunk.
// ---------------------------------------------------------------------------------------------------------------
struct InitModelArgs {
    const float *depth_metric;    // rows x cols row-major (DEPTH_METRIC, raw)
    const float *depth_filtered;  // column-major (depthCurrent after sf_filter_depth)
    const uint8_t *color;         // rows x cols x 3
    const float *b_img;           // column-major
    int rows, cols, time;
    float pose[16], cx, cy, fx, fy, max_depth;
    float *out;                   // rows*cols*12, zero-filled
    int *count;
};

struct FbVertex {
    PV3 pos, normal;
    float radius;
};
template <class Depth>
__device__ __forceinline__ bool feedback_vertex(const Depth &D, int i, int j, const InitModelArgs &a, FbVertex &out) {
    const float W = float(a.cols), H = float(a.rows);
    const float tx = float(double(float(i) / W) + 1.0 / double(2 * W));  // FeedbackBuffer.cpp:46-47
    const float ty = float(double(float(j) / H) + 1.0 / double(2 * H));
    const float x = tx * W, y = ty * H;
    const float camz = 1.0f / a.fx, camw = 1.0f / a.fy;
    auto vertex = [&](int ii, int jj, float xx, float yy) {
        const float z = D(min(max(ii, 0), a.cols - 1), min(max(jj, 0), a.rows - 1));
        return PV3{(xx - a.cx) * z * camz, (yy - a.cy) * z * camw, z};
    };
    const PV3 v = vertex(i, j, x, y);
    const PV3 xf = vertex(i + 1, j, x + 1.f, y), xb = vertex(i - 1, j, x - 1.f, y);
    const PV3 yf = vertex(i, j + 1, x, y + 1.f), yb = vertex(i, j - 1, x, y - 1.f);
    auto half_sum = [](PV3 p, PV3 q) { return PV3{(p.x + q.x) / 2.f, (p.y + q.y) / 2.f, (p.z + q.z) / 2.f}; };
    const PV3 del_x = psub(half_sum(xb, v), half_sum(xf, v));
    const PV3 del_y = psub(half_sum(yb, v), half_sum(yf, v));
    out.pos = v;
    out.normal = pnormalize(pcross(del_x, del_y));
    const float meanFocal = ((1.0f / fabsf(camz)) + (1.0f / fabsf(camw))) / 2.0f;
    const float radius = (v.z / meanFocal) * 1.41421356237f;
    out.radius = fminf(2.0f * radius, radius / fabsf(out.normal.z));
    return !(v.z <= 0.f || v.z > a.max_depth);
}

__global__ __launch_bounds__(1024) void sf_init_model_kernel(const InitModelArgs *tab) {
    __shared__ int wcount[2][16];
    __shared__ int base[2];
    const InitModelArgs &a = tab[blockIdx.x];  // one workgroup per map
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.rows * a.cols;
    if (tid < 2) base[tid] = 0;
    __syncthreads();
    const float *P = a.pose;
    auto Draw = [&](int i, int j) { return a.depth_metric[(size_t)j * a.cols + i]; };
    auto Dfil = [&](int i, int j) { return a.depth_filtered[j + (size_t)i * a.rows]; };
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int idx = c0 + tid;  // = j + i * rows: the reference's point order
        const bool in = idx < n;
        const int i = in ? idx / a.rows : 0, j = in ? idx - i * a.rows : 0;
        FbVertex vr, vf;
        const bool okr = feedback_vertex(Draw, i, j, a, vr) && in;
        const bool okf = feedback_vertex(Dfil, i, j, a, vf) && in;
        const unsigned long long mr = __ballot(okr), mf = __ballot(okf);
        if (lane == 0) {
            wcount[0][wave] = __popcll(mr);
            wcount[1][wave] = __popcll(mf);
        }
        __syncthreads();
        int offr = base[0], offf = base[1];
        for (int w = 0; w < wave; w++) {
            offr += wcount[0][w];
            offf += wcount[1][w];
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (okr) {  // RAW: position and colour (init_unstable.vert:36,43-45)
            float *s = a.out + (size_t)(offr + __popcll(mr & lt)) * 12;
            s[0] = P[0] * vr.pos.x + P[4] * vr.pos.y + P[8] * vr.pos.z + P[12];
            s[1] = P[1] * vr.pos.x + P[5] * vr.pos.y + P[9] * vr.pos.z + P[13];
            s[2] = P[2] * vr.pos.x + P[6] * vr.pos.y + P[10] * vr.pos.z + P[14];
            const uint8_t *c = a.color + ((size_t)j * a.cols + i) * 3;
            s[4] = float((int(c[0]) << 16) + (int(c[1]) << 8) + int(c[2]));
            s[5] = 1.0f;
            s[6] = 1.0f;
            s[7] = float(a.time);
        }
        if (okf) {  // FILTERED: normal, radius, b as confidence (init_unstable.vert:38-40,47)
            float *s = a.out + (size_t)(offf + __popcll(mf & lt)) * 12;
            const int k = (int)roundf(a.b_img[j + (size_t)i * a.rows] * 255.0f);
            s[3] = float(k & 0xFF) / 255.0f;
            s[8] = P[0] * vf.normal.x + P[4] * vf.normal.y + P[8] * vf.normal.z;
            s[9] = P[1] * vf.normal.x + P[5] * vf.normal.y + P[9] * vf.normal.z;
            s[10] = P[2] * vf.normal.x + P[6] * vf.normal.y + P[10] * vf.normal.z;
            s[11] = vf.radius;
        }
        __syncthreads();
        if (tid == 0) {
            int tr = 0, tf = 0;
            for (int w = 0; w < 16; w++) {
                tr += wcount[0][w];
                tf += wcount[1][w];
            }
            base[0] += tr;
            base[1] += tf;
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.count[0] = base[0];
        a.count[1] = base[1];
    }
}

__global__ __launch_bounds__(256) void sf_init_model_zero_kernel(const InitModelArgs *tab) {
    const InitModelArgs &a = tab[blockIdx.y];
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q < (size_t)a.rows * a.cols * 12) a.out[q] = 0.f;
}
// the draw call is sized by the RAW buffer: slots from count_raw on are not part of the model (zeroed)
__global__ __launch_bounds__(256) void sf_init_model_trim_kernel(const InitModelArgs *tab) {
    const InitModelArgs &a = tab[blockIdx.y];
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q < (size_t)a.rows * a.cols * 12 && q >= (size_t)a.count[0] * 12) a.out[q] = 0.f;
}
