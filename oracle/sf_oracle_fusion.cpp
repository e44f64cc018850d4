// sf_oracle_fusion.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE) for the fusion half of
// Reconstruction::fuseFrame (reference Reconstruction.cpp:235-325), a scalar restatement
// with the OpenGL pipeline written out:
//   IndexMap::predictIndices   IndexMap.cpp:117-184     Shaders/index_map.vert, index_map.frag
//   GlobalModel::fuse          GlobalModel.cpp:322-492  Shaders/data.vert, data.geom, data.frag (association),
//                                                       update.vert (merge); surfels.glsl, geometry.glsl, color.glsl
//   GlobalModel::clean         GlobalModel.cpp:494-601  Shaders/copy_unstable.vert, copy_unstable.geom
// PARITY UNPINNED: no reference vectors exist and the reference runs on a GL driver. Choices where GL leaves room, in
// addition to those of sf_oracle_predict.cpp:
//   * a size-1 GL point produces the fragment of the pixel that contains its window position (floor); a point whose
//     centre is outside the clip volume is dropped; GL_DEPTH_TEST / GL_LESS is on (Utils/GUI.h:67-69) and compares the
//     float window depth; primitives are processed in buffer order, so the first of two equal depths wins;
//   * textureLod on the GL_NEAREST textures (Utils/GPUTexture.cpp:39, draw = false) reads texel
//     clamp(floor(u * size), 0, size - 1) with the product in binary32. The association windows step their texture
//     coordinate in half texels by repeated float addition (data.vert:133-135, copy_unstable.vert:62-64): the loops
//     below repeat exactly that float sequence;
//   * exp() and log() of update.vert:58-60 and surfels.glsl:45 are sf_exp_det / sf_log_det / sf_exp_neg
//     (include/sf_detmath.h); acos(c) < 0.5 of data.vert:149 is evaluated as cos(0.5) < c <= 1 (acos is undefined
//     above 1: NaN, comparison false); min / max are the GLSL definitions (y < x ? y : x, x < y ? y : x);
//   * two frame pixels that pick the same model surfel both write texel `best` of the update maps; with the depth test
//     on and gl_Position.z = 0 for both, the first in the uv order (x outer, y inner: GlobalModel.cpp:77-84) stays;
//   * transform feedback stops when its buffer is full: the map is truncated at `capacity` surfels.
#include "sf_oracle_fusion.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "../include/sf_detmath.h"
#include "sf_oracle.hpp"

namespace sfo {
namespace {
struct V3 { float x, y, z; };
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { const float n = std::sqrt(dot(a, a)); return {a.x / n, a.y / n, a.z / n}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float gl_min(float x, float y) { return y < x ? y : x; }
inline float gl_max(float x, float y) { return x < y ? y : x; }
inline V3 xform(const float *T, V3 v) {  // (T * vec4(v, 1)).xyz, column-major T
    return {T[0] * v.x + T[4] * v.y + T[8] * v.z + T[12], T[1] * v.x + T[5] * v.y + T[9] * v.z + T[13], T[2] * v.x + T[6] * v.y + T[10] * v.z + T[14]};
}
inline V3 rotate(const float *T, V3 v) {  // mat3(T) * v
    return {T[0] * v.x + T[4] * v.y + T[8] * v.z, T[1] * v.x + T[5] * v.y + T[9] * v.z, T[2] * v.x + T[6] * v.y + T[10] * v.z};
}
inline int texel(float u, int size) {  // GL_NEAREST, clamp to edge
    const float t = std::floor(u * float(size));
    return t < 0.f ? 0 : (t > float(size - 1) ? size - 1 : int(t));
}
inline float encode_color(float r, float g, float b) {  // color.glsl:19-25
    int rgb = int(std::round(r * 255.0f));
    rgb = (rgb << 8) + int(std::round(g * 255.0f));
    rgb = (rgb << 8) + int(std::round(b * 255.0f));
    return float(rgb);
}
inline V3 decode_color(float c) {  // color.glsl:27-34
    const int k = int(c);
    return {float((k >> 16) & 0xFF) / 255.0f, float((k >> 8) & 0xFF) / 255.0f, float(k & 0xFF) / 255.0f};
}
void invert(const float pose[16], float t_inv[16]) {  // pose.inverse(): [C5] double Gauss-Jordan, rounded to float
    double A[16], Ai[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) A[r * 4 + c] = double(pose[r + 4 * c]);
    inverse_double(A, Ai, 4);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) t_inv[r + 4 * c] = float(Ai[r * 4 + c]);
}

// What the index map's four textures hold at a texel whose index is `idx` (index_map.vert:56-59, index_map.frag):
// recomputed from the surfel instead of stored -- the same operations on the same inputs.
struct IndexTexel {
    V3 pos;  // vPosHome
    float conf;
    float color, hist, t_init, t_last;
    V3 normal;
    float radius;
};
inline IndexTexel index_texel(const float *surfels, uint32_t idx, const float *t_inv) {
    const float *q = surfels + size_t(idx) * 12;
    IndexTexel t;
    t.pos = xform(t_inv, V3{q[0], q[1], q[2]});
    t.conf = q[3];
    t.color = q[4]; t.hist = q[5]; t.t_init = q[6]; t.t_last = q[7];
    t.normal = normalize(rotate(t_inv, V3{q[8], q[9], q[10]}));
    t.radius = q[11];
    return t;
}

struct DataRecord {  // the three varyings data.geom emits, and the association
    float v[12];
    int update_id;   // 1: merge into `best`, 2: new unstable surfel
    uint32_t best;
};

// data.vert for the frame pixel (i, j); false when nothing is emitted (updateId 0)
bool data_vertex(const FrameImages &f, int i, int j, const float *pose, const float *t_inv, const ModelParams &p, int time, float weighting,
                 const float *surfels, const uint32_t *index_map, DataRecord &out) {
    const int rows = f.rows, cols = f.cols;
    const float W = float(cols), H = float(rows);
    const float tx = float(double(float(i) / W) + 1.0 / double(2 * W));  // GlobalModel.cpp:81-82
    const float ty = float(double(float(j) / H) + 1.0 / double(2 * H));
    const float x = tx * W, y = ty * H;                                   // data.vert:79-80
    const float camz = float(1.0 / double(p.fx)), camw = float(1.0 / double(p.fy));  // GlobalModel.cpp:365-368
    auto Draw = [&](int ii, int jj) { return f.depth_metric[size_t(std::min(std::max(jj, 0), rows - 1)) * cols + std::min(std::max(ii, 0), cols - 1)]; };
    auto Dfil = [&](int ii, int jj) { return f.depth_filtered[std::min(std::max(jj, 0), rows - 1) + size_t(std::min(std::max(ii, 0), cols - 1)) * rows]; };
    auto vertex = [&](float z, float xx, float yy) { return V3{(xx - p.cx) * z * camz, (yy - p.cy) * z * camw, z}; };  // geometry.glsl:21-25
    const V3 vPosLocal = vertex(Draw(i, j), x, y);                       // :83
    const V3 world = xform(pose, vPosLocal);                              // :84
    const V3 vf = vertex(Dfil(i, j), x, y);                               // :87
    const float probIsStatic = f.b_img[j + size_t(i) * rows];             // :89
    const uint8_t *c = f.color + (size_t(j) * cols + i) * 3;              // :92-94
    const float color = encode_color(float(c[0]) / 255.0f, float(c[1]) / 255.0f, float(c[2]) / 255.0f);
    // :97 getNormal on the filtered depth (geometry.glsl:28-40)
    const V3 xf = vertex(Dfil(i + 1, j), x + 1.f, y), xb = vertex(Dfil(i - 1, j), x - 1.f, y);
    const V3 yf = vertex(Dfil(i, j + 1), x, y + 1.f), yb = vertex(Dfil(i, j - 1), x, y - 1.f);
    auto half_sum = [](V3 a, V3 b) { return V3{(a.x + b.x) / 2.f, (a.y + b.y) / 2.f, (a.z + b.z) / 2.f}; };
    const V3 vNormLocal = normalize(cross(sub(half_sum(xb, vf), half_sum(xf, vf)), sub(half_sum(yb, vf), half_sum(yf, vf))));
    // :98 getRadius (surfels.glsl:19-35)
    const float meanFocal = ((1.0f / std::fabs(camz)) + (1.0f / std::fabs(camw))) / 2.0f;
    const float radius0 = (vf.z / meanFocal) * 1.41421356237f;
    const float radius = gl_min(2.0f * radius0, radius0 / std::fabs(vNormLocal.z));
    const V3 nWorld = rotate(pose, vNormLocal);
    // :101 confidence (surfels.glsl:37-47)
    const float pcx = x - p.cx, pcy = y - p.cy;
    const float radialDist = std::sqrt(pcx * pcx + pcy * pcy) / 200.0f;
    const float radialConf = sf_exp_neg((radialDist * radialDist) / (2.0f * 0.72f));
    float conf = gl_min(probIsStatic, gl_min(weighting, radialConf));    // :102
    float t_last = 0.f;                                                  // :107
    out.update_id = 0;
    out.best = 0;
    const float ftime = float(time);
    // :114-116
    const bool parity = (int(x) % 2 == int(ftime) % 2) && (int(y) % 2 == int(ftime) % 2);
    const bool neighbours = !(Draw(i - 1, j) == 0.f) && !(Draw(i, j - 1) == 0.f) && !(Draw(i + 1, j) == 0.f) && !(Draw(i, j + 1) == 0.f);  // :52-71
    if (parity && neighbours && vPosLocal.z > 0.f && vPosLocal.z <= p.max_depth) {
        int counter = 0;
        const float scale = 4.0f;  // IndexMap::FACTOR
        const float indexXStep = (1.0f / (W * scale)) * 0.5f;  // :121-122
        const float indexYStep = (1.0f / (H * scale)) * 0.5f;
        float bestDist = 1000.f;
        const float windowMultiplier = 2.f;
        const float xl = (x - p.cx) * camz, yl = (y - p.cy) * camw;  // :128-129
        const float lambda = std::sqrt(xl * xl + yl * yl + 1.f);
        const V3 ray{xl, yl, 1.f};
        const int W4 = cols * 4, H4 = rows * 4;
        for (float u = tx - (scale * indexXStep * windowMultiplier); u < tx + (scale * indexXStep * windowMultiplier); u += indexXStep)
            for (float v = ty - (scale * indexYStep * windowMultiplier); v < ty + (scale * indexYStep * windowMultiplier); v += indexYStep) {
                const uint32_t current = index_map[size_t(texel(v, H4)) * W4 + texel(u, W4)];  // :138
                if (current > 0U) {
                    const IndexTexel t = index_texel(surfels, current, t_inv);
                    if (std::fabs((t.pos.z * lambda) - (vPosLocal.z * lambda)) < 0.05f) {  // :144
                        const float dist = length(cross(ray, t.pos)) / length(ray);         // :146
                        bool angle_ok = std::fabs(t.normal.z) < 0.75f;                       // :150
                        if (!angle_ok) {
                            const float cs = dot(t.normal, vNormLocal) / (length(t.normal) * length(vNormLocal));  // :73-76
                            angle_ok = cs > 0.87758256189f && cs <= 1.0f;                    // acos(cs) < 0.5
                        }
                        if (dist < bestDist && angle_ok) {
                            counter++;
                            bestDist = dist;
                            out.best = current;
                        }
                    }
                }
            }
        if (counter > 0) {  // :163-167
            out.update_id = 1;
            t_last = -1.f;
        } else {            // :169-180
            out.update_id = 2;
            t_last = -2.f;
            conf = 0.f;
            if (probIsStatic > 0.5f) conf = 0.08f;
        }
    }
    float *o = out.v;
    o[0] = world.x; o[1] = world.y; o[2] = world.z; o[3] = conf;
    o[4] = color; o[5] = 1.0f; o[6] = ftime; o[7] = t_last;  // :93-95,103,107
    o[8] = nWorld.x; o[9] = nWorld.y; o[10] = nWorld.z; o[11] = radius;
    return out.update_id > 0;  // data.geom:35
}

// update.vert for model surfel `q` merged with the data record `d`
void merge_surfel(const float *q, const float *d, int time, float *o) {
    float c_k = q[3];                        // :52
    const V3 v_k{q[0], q[1], q[2]};
    float a = d[3];                          // :55
    const V3 v_g{d[0], d[1], d[2]};
    const float hist = q[5];                 // :58
    const float max_val = 0.99f, min_val = 0.01f;
    a = gl_max(min_val, gl_min(0.53f, 2.f * a * a));   // :63
    c_k = gl_max(min_val, gl_min(c_k, max_val));       // :64
    float ltm = sf_log_det(1.0f / (1.0f - c_k) - 1.0f);  // :66
    ltm = ltm + sf_log_det(a / (1.0f - a));               // :67
    const float c_k1 = 1.0f - (1.0f / (1.0f + sf_exp_det(ltm)));  // :68
    if (d[11] < (1.0f + 0.5f) * q[11]) {     // :70
        const float w = hist * c_k, den = hist * c_k + a;
        o[0] = ((w * v_k.x) + (a * v_g.x)) / den;  // :72
        o[1] = ((w * v_k.y) + (a * v_g.y)) / den;
        o[2] = ((w * v_k.z) + (a * v_g.z)) / den;
        o[3] = c_k1;
        const V3 oldCol = decode_color(q[4]), newCol = decode_color(d[4]);  // :74-75
        o[4] = encode_color(((w * oldCol.x) + (a * newCol.x)) / den, ((w * oldCol.y) + (a * newCol.y)) / den,
                            ((w * oldCol.z) + (a * newCol.z)) / den);       // :77-79
        o[5] = hist + 1.0f;
        o[6] = q[6];
        o[7] = float(time);
        const V3 n = normalize(V3{((w * q[8]) + (a * d[8])) / den, ((w * q[9]) + (a * d[9])) / den, ((w * q[10]) + (a * d[10])) / den});  // :81-83
        o[8] = n.x; o[9] = n.y; o[10] = n.z;
        o[11] = ((w * q[11]) + (a * d[11])) / den;
    } else {                                  // :85-98
        std::memcpy(o, q, 12 * sizeof(float));
        o[3] = c_k1;
        o[5] = hist + 1.0f;
        o[7] = float(time);
    }
}

// copy_unstable.vert for one vertex (a model surfel or an entry of the new-unstable buffer); true = kept
bool clean_vertex(const float *q, const float *t_inv, const ModelParams &p, int rows, int cols, int time, const float *surfels,
                  const uint32_t *index_map, float *o) {
    std::memcpy(o, q, 12 * sizeof(float));
    int test = 1;
    const V3 localPos = xform(t_inv, V3{q[0], q[1], q[2]});               // :46
    const float W = float(cols), H = float(rows);
    const float x = ((p.fx * localPos.x) / localPos.z) + p.cx;            // :48-49
    const float y = ((p.fy * localPos.y) / localPos.z) + p.cy;
    const float scale = 4.0f;
    const float indexXStep = (1.0f / (W * scale)) * 0.5f;                 // :53-54
    const float indexYStep = (1.0f / (H * scale)) * 0.5f;
    const float windowMultiplier = 2.f;
    int count = 0, zCount = 0;
    const float ftime = float(time), fdelta = float(p.time_delta);
    if (ftime - q[7] < fdelta && localPos.z > 0.f && x > 0.f && y > 0.f && x < W && y < H) {  // :61
        const int W4 = cols * 4, H4 = rows * 4;
        for (float u = x / W - (scale * indexXStep * windowMultiplier); u < x / W + (scale * indexXStep * windowMultiplier); u += indexXStep)
            for (float v = y / H - (scale * indexYStep * windowMultiplier); v < y / H + (scale * indexYStep * windowMultiplier); v += indexYStep) {
                const uint32_t current = index_map[size_t(texel(v, H4)) * W4 + texel(u, W4)];
                if (current > 0U) {
                    const IndexTexel t = index_texel(surfels, current, t_inv);
                    const float dx = t.pos.x - localPos.x, dy = t.pos.y - localPos.y;
                    if (t.t_init < q[6] && t.conf > p.conf_high && t.pos.z > localPos.z && t.pos.z - localPos.z < 0.01f &&
                        std::sqrt(dx * dx + dy * dy) < q[11] * 1.4f)                          // :74-78
                        count++;
                    if (t.t_last == ftime && t.conf > 0.4f * p.conf_high && t.pos.z > localPos.z && t.pos.z - localPos.z > 0.01f)  // :83-86
                        zCount++;
                }
            }
    }
    if (count > 6 || zCount > 5) test = 0;                                // :95-98
    if (o[7] == -2.f) o[7] = ftime;                                       // :101-104
    if ((o[7] == -1.f || ((ftime - o[7]) > 10.f && o[3] < 0.5f)) || (o[3] == 0.0f)) test = 0;  // :108-111
    if (o[7] > 0.f && ftime - o[7] > fdelta) test = 1;                    // :113-116
    return test > 0;
}
}  // namespace

void predict_indices(const float *surfels, int count, const float t_inv[16], const ModelParams &p, int rows, int cols, int time,
                     uint32_t *index_map) {
    const int W4 = cols * 4, H4 = rows * 4;
    std::vector<float> zbuf(size_t(W4) * H4, 1.0f);
    std::fill(index_map, index_map + size_t(W4) * H4, 0u);
    const float camx = p.cx * 4.f, camy = p.cy * 4.f, camz = p.fx * 4.f, camw = p.fy * 4.f;  // IndexMap.cpp:136-139
    const float fcols = float(cols) * 4.f, frows = float(rows) * 4.f;                          // :144-145
    for (int s = 0; s < count; s++) {
        const float *q = surfels + size_t(s) * 12;
        const V3 h = xform(t_inv, V3{q[0], q[1], q[2]});                                      // index_map.vert:38
        if (h.z > p.max_depth || h.z < 0.f || float(time) - q[7] > float(p.time_delta)) continue;  // :43-48
        const float ndc_x = ((((camz * h.x) / h.z) + camx) - (fcols * 0.5f)) / (fcols * 0.5f);  // :51-52
        const float ndc_y = ((((camw * h.y) / h.z) + camy) - (frows * 0.5f)) / (frows * 0.5f);
        const float ndc_z = h.z / p.max_depth;                                                  // :57
        if (!(ndc_x >= -1.f && ndc_x <= 1.f && ndc_y >= -1.f && ndc_y <= 1.f && ndc_z >= -1.f && ndc_z <= 1.f)) continue;
        const float xw = (ndc_x + 1.f) * (fcols * 0.5f), yw = (ndc_y + 1.f) * (frows * 0.5f);
        const float fx_ = std::floor(xw), fy_ = std::floor(yw);
        if (!(fx_ >= 0.f && fx_ < float(W4) && fy_ >= 0.f && fy_ < float(H4))) continue;
        const size_t o = size_t(int(fy_)) * W4 + int(fx_);
        const float depth = ndc_z * 0.5f + 0.5f;
        if (!(depth < zbuf[o])) continue;  // GL_LESS
        zbuf[o] = depth;
        index_map[o] = uint32_t(s);        // index_map.frag:36 (vertexId; surfel 0 reads as "empty")
    }
}

int fuse_frame(SurfelMap &m, const FrameImages &f, const float *in_pose, float weight_multiplier, const ModelParams &p) {
    const int rows = f.rows, cols = f.cols;
    const size_t npx = size_t(rows) * cols;
    auto compose = [&](const float *a, const float *b, float *out) {  // Eigen::Matrix4f product, column-major
        float r[16];
        for (int c = 0; c < 4; c++)
            for (int rr = 0; rr < 4; rr++) {
                float acc = a[rr] * b[4 * c];
                for (int k = 1; k < 4; k++) acc = acc + a[rr + 4 * k] * b[k + 4 * c];
                r[rr + 4 * c] = acc;
            }
        std::memcpy(out, r, sizeof r);
    };
    int overflow = 0;
    m.index_map.assign(size_t(rows) * 4 * cols * 4, 0u);
    if (m.tick == 1) {  // Reconstruction.cpp:255-262
        if (in_pose) compose(m.pose, in_pose, m.pose);
        std::vector<float> init(npx * 12);
        int n = init_model_from_frame(f.depth_metric, f.depth_filtered, f.color, f.b_img, rows, cols, m.pose, p, m.tick, init.data());
        if (n > m.capacity) { n = m.capacity; overflow = 1; }
        m.surfels.assign(init.begin(), init.begin() + size_t(n) * 12);
        m.count = n;
        m.stats[0] = m.stats[1] = m.stats[2] = 0;
        m.stats[3] = n;
        m.tick++;
        return overflow;
    }
    float last_pose[16];
    std::memcpy(last_pose, m.pose, sizeof last_pose);
    compose(m.pose, in_pose, m.pose);                                                   // :268
    const float weighting = sf_fusion_weighting(last_pose, m.pose, weight_multiplier);  // :270-282
    float t_inv[16];
    invert(m.pose, t_inv);
    const int time = m.tick;
    predict_indices(m.surfels.data(), m.count, t_inv, p, rows, cols, time, m.index_map.data());  // :284

    // ---- GlobalModel::fuse, first half: data association (data.vert / .geom / .frag) ----
    std::vector<DataRecord> emitted;
    std::vector<uint32_t> winner(size_t(m.count), 0xffffffffu);  // update-map texel -> emission index of the record it holds
    int merged = 0;
    for (int i = 0; i < cols; i++)
        for (int j = 0; j < rows; j++) {
            DataRecord d;
            if (!data_vertex(f, i, j, m.pose, t_inv, p, time, weighting, m.surfels.data(), m.index_map.data(), d)) continue;
            if (d.update_id == 1) {
                merged++;
                if (winner[d.best] == 0xffffffffu) winner[d.best] = uint32_t(emitted.size());  // first fragment passes GL_LESS
            }
            emitted.push_back(d);
        }
    // ---- second half: merge (update.vert), the whole model is rewritten ----
    std::vector<float> updated(size_t(m.count) * 12);
    int n_updated = 0;
    for (int s = 0; s < m.count; s++) {
        const float *q = m.surfels.data() + size_t(s) * 12;
        if (winner[s] != 0xffffffffu) {  // newColor.w == -1 (:45)
            merge_surfel(q, emitted[winner[s]].v, time, updated.data() + size_t(s) * 12);
            n_updated++;
        } else
            std::memcpy(updated.data() + size_t(s) * 12, q, 12 * sizeof(float));  // :100-106
    }
    predict_indices(updated.data(), m.count, t_inv, p, rows, cols, time, m.index_map.data());  // :300

    // ---- GlobalModel::clean: the model, then the new-unstable buffer, through copy_unstable ----
    std::vector<float> out;
    out.reserve((size_t(m.count) + emitted.size()) * 12);
    int n_out = 0;
    float o[12];
    auto push = [&](const float *v) {
        if (n_out >= m.capacity) { overflow = 1; return; }
        out.insert(out.end(), v, v + 12);
        n_out++;
    };
    for (int s = 0; s < m.count; s++)
        if (clean_vertex(updated.data() + size_t(s) * 12, t_inv, p, rows, cols, time, updated.data(), m.index_map.data(), o)) push(o);
    for (const DataRecord &d : emitted)
        if (clean_vertex(d.v, t_inv, p, rows, cols, time, updated.data(), m.index_map.data(), o)) push(o);
    m.surfels.swap(out);
    m.count = n_out;
    m.stats[0] = int(emitted.size());
    m.stats[1] = merged;
    m.stats[2] = n_updated;
    m.stats[3] = n_out;
    m.tick++;
    return overflow;
}
}  // namespace sfo
