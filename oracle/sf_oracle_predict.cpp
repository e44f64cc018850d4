// sf_oracle_predict.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE) for the frame-to-model
// prediction: a scalar restatement of Reconstruction::getPredictedImages
// (reference Reconstruction.cpp:628-720) with the OpenGL pipeline written out:
//   IndexMap::combinedPredict  IndexMap.cpp:221-300   Shaders/splat.vert, Shaders/combo_splat.frag, color.glsl
//   Resize::image + denseEnough  Shaders/Resize.cpp, resize.frag, Reconstruction.cpp:218-233
//   FillIn passes              Shaders/FillIn.cpp, fill_vertex.frag, fill_vertex_from_texture.frag, fill_rgb.frag
//   extractDepthFromPrediction Shaders/FillIn.cpp:263-295, extract_depth.frag
// PARITY UNPINNED: no reference vectors exist and the reference runs on a GL driver. Where GL leaves room,
// this file fixes a choice: a point sprite covers the pixels whose centres lie within
// gl_PointSize/2 of the projected centre in both axes; points whose centre is outside the clip volume are
// dropped; the depth test is GL_LESS on the float gl_FragDepth (no 24-bit quantisation), surfels are
// rasterised in buffer order, so the first of two equal depths wins; RGBA8 targets store round(255 c);
// nearest-texel fetches; normalize(v) = v / sqrt(dot(v, v)); float arithmetic without contraction.
#include "sf_oracle_predict.hpp"

#include <cmath>
#include <limits>
#include <vector>

namespace sfo {
namespace {
struct V3 { float x, y, z; };
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 normalize(V3 a) { const float n = std::sqrt(dot(a, a)); return {a.x / n, a.y / n, a.z / n}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float min4(float a, float b, float c, float d) { return std::fmin(a, std::fmin(b, std::fmin(c, d))); }
inline float max4(float a, float b, float c, float d) { return std::fmax(a, std::fmax(b, std::fmax(c, d))); }

struct Target {  // one render target set of IndexMap (vertex.z + the RGBA8 image), cleared to 0 / depth 1
    std::vector<float> z, zbuf;
    std::vector<uint8_t> rgb;
    Target(int n) : z(n, 0.f), zbuf(n, 1.f), rgb(size_t(n) * 3, 0) {}
};

void splat_pass(const float *surfels, int count, const float *t_inv, const ModelParams &p, int rows, int cols, float conf_threshold,
                Target &t) {
    auto T = [&](int r, int c) { return t_inv[r + 4 * c]; };
    const float fcols = float(cols), frows = float(rows);
    for (int s = 0; s < count; s++) {
        const float *q = surfels + size_t(s) * 12;
        const V3 vp{q[0], q[1], q[2]};
        const float conf = q[3], col = q[4], tlast = q[7];
        // splat.vert:55  vPosHome = t_inv * vec4(vPosition.xyz, 1.0)
        const V3 h{T(0, 0) * vp.x + T(0, 1) * vp.y + T(0, 2) * vp.z + T(0, 3), T(1, 0) * vp.x + T(1, 1) * vp.y + T(1, 2) * vp.z + T(1, 3),
                   T(2, 0) * vp.x + T(2, 1) * vp.y + T(2, 2) * vp.z + T(2, 3)};
        // :57 cull
        if (h.z > p.max_depth || h.z < 0.4f || conf < conf_threshold || float(p.time) - tlast > float(p.time_delta) || tlast > float(p.max_time)) continue;
        // :64 projectPoint -> normalised device coordinates; the viewport maps them back to window coordinates
        const float ndc_x = ((((p.fx * h.x) / h.z) + p.cx) - (fcols * 0.5f)) / (fcols * 0.5f);
        const float ndc_y = ((((p.fy * h.y) / h.z) + p.cy) - (frows * 0.5f)) / (frows * 0.5f);
        if (!(ndc_x >= -1.f && ndc_x <= 1.f && ndc_y >= -1.f && ndc_y <= 1.f)) continue;  // point clipped by its centre
        const float xw = (ndc_x + 1.f) * (fcols * 0.5f), yw = (ndc_y + 1.f) * (frows * 0.5f);
        // :68 normal into the camera frame, radius
        const V3 nin{q[8], q[9], q[10]};
        const V3 n = normalize(V3{T(0, 0) * nin.x + T(0, 1) * nin.y + T(0, 2) * nin.z, T(1, 0) * nin.x + T(1, 1) * nin.y + T(1, 2) * nin.z,
                                  T(2, 0) * nin.x + T(2, 1) * nin.y + T(2, 2) * nin.z});
        const float rad = q[11];
        // :70-72 two tangent vectors spanning the disc's bounding diamond
        const V3 x1 = mul(mul(normalize(V3{n.y - n.z, -n.x, n.x}), rad), 1.41421356f);
        const V3 y1 = cross(n, x1);
        auto proj = [&](V3 a, float &px, float &py) { px = ((p.fx * a.x) / a.z) + p.cx; py = ((p.fy * a.y) / a.z) + p.cy; };  // :39-44
        float p1x, p1y, p2x, p2y, p3x, p3y, p4x, p4y;
        proj(add(h, x1), p1x, p1y);
        proj(add(h, y1), p2x, p2y);
        proj(sub(h, y1), p3x, p3y);
        proj(sub(h, x1), p4x, p4y);
        const float xDiff = std::fabs(max4(p1x, p2x, p3x, p4x) - min4(p1x, p2x, p3x, p4x));  // :79-83
        const float yDiff = std::fabs(max4(p1y, p2y, p3y, p4y) - min4(p1y, p2y, p3y, p4y));
        const float size = std::fmax(0.f, std::fmax(xDiff, yDiff));                              // :85
        if (!(size > 0.f)) continue;
        const float half = size * 0.5f;
        const int i0 = std::max(0, int(std::ceil(xw - half - 0.5f))), i1 = std::min(cols - 1, int(std::floor(xw + half - 0.5f)));
        const int j0 = std::max(0, int(std::ceil(yw - half - 0.5f))), j1 = std::min(rows - 1, int(std::floor(yw + half - 0.5f)));
        const float sqrRad = rad * rad;  // combo_splat.frag:42
        const float pn = dot(h, n);
        for (int j = j0; j <= j1; j++)
            for (int i = i0; i <= i1; i++) {
                // combo_splat.frag:37-49
                const float fx_ = float(i) + 0.5f, fy_ = float(j) + 0.5f;  // gl_FragCoord
                const V3 l = normalize(V3{(fx_ - p.cx) / p.fx, (fy_ - p.cy) / p.fy, 1.0f});
                const V3 corrected = mul(l, pn / dot(l, n));
                const V3 diff = sub(corrected, h);
                if (dot(diff, diff) > sqrRad) continue;  // discard
                const float depth = (corrected.z / (2.f * p.max_depth)) + 0.5f;  // :63 gl_FragDepth
                const int o = j * cols + i;
                if (!(depth >= 0.f && depth <= 1.f) || !(depth < t.zbuf[o])) continue;  // depth range clip, GL_LESS
                t.zbuf[o] = depth;
                t.z[o] = corrected.z;  // :55-57 vertex.z
                const int c = int(col);  // color.glsl:30-37 decodeColor; the RGBA8 target stores the bytes back
                t.rgb[size_t(o) * 3 + 0] = uint8_t((c >> 16) & 0xFF);
                t.rgb[size_t(o) * 3 + 1] = uint8_t((c >> 8) & 0xFF);
                t.rgb[size_t(o) * 3 + 2] = uint8_t(c & 0xFF);
            }
    }
}
}  // namespace

bool predict_from_model(const float *surfels, int count, const float t_inv[16], const ModelParams &p, int rows, int cols,
                        const uint16_t *filtered_mm, const uint8_t *color, const float *b_img, float *depth_pred, float *inten_pred) {
    const int n = rows * cols;
    Target low(n), high(n);
    splat_pass(surfels, count, t_inv, p, rows, cols, p.conf_low, low);    // Reconstruction.cpp:638-645
    splat_pass(surfels, count, t_inv, p, rows, cols, p.conf_high, high);  // :648-655
    // resize.image(imageTexLowConf, imageBuff) to (cols/40) x (rows/40), nearest texel at the target pixel's centre (:658)
    const int rw = cols / 40, rh = rows / 40;
    int sum = 0;
    for (int j = 0; j < rh; j++)
        for (int i = 0; i < rw; i++) {
            const int sx = std::min(cols - 1, int(((float(i) + 0.5f) / float(rw)) * float(cols)));
            const int sy = std::min(rows - 1, int(((float(j) + 0.5f) / float(rh)) * float(rows)));
            const uint8_t *c = &low.rgb[(size_t(sy) * cols + sx) * 3];
            sum += (c[0] > 0 && c[1] > 0 && c[2] > 0) ? 1 : 0;  // denseEnough :226-228
        }
    const bool dense = (rw * rh > 0) && (float(sum) / float(rh * rw) > 0.25f);  // :232
    const float norm_factor = 1.f / 255.f;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const int o = y * cols + x;
            float z;
            const uint8_t *c;
            const bool high_empty_rgb = (int(high.rgb[size_t(o) * 3]) + int(high.rgb[size_t(o) * 3 + 1]) + int(high.rgb[size_t(o) * 3 + 2])) == 0;
            const bool low_empty_rgb = (int(low.rgb[size_t(o) * 3]) + int(low.rgb[size_t(o) * 3 + 1]) + int(low.rgb[size_t(o) * 3 + 2])) == 0;
            if (!dense) {  // :663-669
                float z1 = low.z[o];
                if (z1 == 0.f) {  // fill_vertex.frag:47-57: raw filtered depth where the pixel is believed static
                    const float zr = float(filtered_mm[o]) / 1000.0f;
                    z1 = (b_img[y + size_t(x) * rows] > 0.6f) ? zr : 0.0f;
                }
                z = (high.z[o] == 0.f) ? z1 : high.z[o];                          // fill_vertex_from_texture.frag:41-48
                const uint8_t *c1 = low_empty_rgb ? &color[size_t(o) * 3] : &low.rgb[size_t(o) * 3];  // fill_rgb.frag:33-36
                c = high_empty_rgb ? c1 : &high.rgb[size_t(o) * 3];
            } else {       // :696-700
                z = (high.z[o] == 0.f) ? low.z[o] : high.z[o];
                c = high_empty_rgb ? &low.rgb[size_t(o) * 3] : &high.rgb[size_t(o) * 3];
            }
            depth_pred[y + size_t(x) * rows] = (z > p.extract_max_depth || z <= 0.f) ? 0.f : z;  // extract_depth.frag:32-38 (discard -> cleared 0)
            const float r = float(c[0]) * norm_factor, g = float(c[1]) * norm_factor, b = float(c[2]) * norm_factor;  // :686-690
            inten_pred[y + size_t(x) * rows] = 0.299f * r + 0.587f * g + 0.114f * b;                                    // :692
        }
    return dense;
}
}  // namespace sfo

// ------------------------------------------------------------------------------------------------
//  GlobalModel::initialise (GlobalModel.cpp:200-258): the two vertex_feedback passes (Reconstruction.cpp:205-216,
//  Shaders/vertex_feedback.vert, vertex_feedback.geom, geometry.glsl:19-41, surfels.glsl:19-35, color.glsl:19-25)
//  and init_unstable.vert. Choices beyond the ones above: texel fetches clamp to the edge; GLSL round() = roundf.
// ------------------------------------------------------------------------------------------------
#include "../include/sf_detmath.h"
namespace sfo {
namespace {
struct FeedbackVertex {
    V3 pos, normal;
    float radius;
};
// vertex_feedback.vert:40-52 for point (i, j) on a float depth image accessed through D(i, j)
template <class Depth>
bool feedback_vertex(const Depth &D, int i, int j, int rows, int cols, const ModelParams &p, FeedbackVertex &out) {
    const float W = float(cols), H = float(rows);
    const float tx = float(double(float(i) / W) + 1.0 / double(2 * W));  // FeedbackBuffer.cpp:46-47
    const float ty = float(double(float(j) / H) + 1.0 / double(2 * H));
    const float x = tx * W, y = ty * H;                                   // vertex_feedback.vert:40-41
    const float camz = 1.0f / p.fx, camw = 1.0f / p.fy;                   // FeedbackBuffer.cpp:89-92
    auto vertex = [&](int ii, int jj, float xx, float yy) {               // geometry.glsl:21-25
        const float z = D(std::min(std::max(ii, 0), cols - 1), std::min(std::max(jj, 0), rows - 1));
        return V3{(xx - p.cx) * z * camz, (yy - p.cy) * z * camw, z};
    };
    const V3 v = vertex(i, j, x, y);
    const V3 xf = vertex(i + 1, j, x + 1.f, y), xb = vertex(i - 1, j, x - 1.f, y);  // geometry.glsl:30-34
    const V3 yf = vertex(i, j + 1, x, y + 1.f), yb = vertex(i, j - 1, x, y - 1.f);
    auto half_sum = [](V3 a, V3 b) { return V3{(a.x + b.x) / 2.f, (a.y + b.y) / 2.f, (a.z + b.z) / 2.f}; };
    const V3 del_x = sub(half_sum(xb, v), half_sum(xf, v));               // :36-37
    const V3 del_y = sub(half_sum(yb, v), half_sum(yf, v));
    out.pos = v;
    out.normal = normalize(cross(del_x, del_y));                          // :39
    const float meanFocal = ((1.0f / std::fabs(camz)) + (1.0f / std::fabs(camw))) / 2.0f;  // surfels.glsl:21
    const float radius = (v.z / meanFocal) * 1.41421356237f;              // :25
    out.radius = std::fmin(2.0f * radius, radius / std::fabs(out.normal.z));  // :27-31
    return !(v.z <= 0.f || v.z > p.max_depth);                            // vertex_feedback.vert:49-56, .geom:34
}
}  // namespace

int init_model_from_frame(const float *depth_metric, const float *depth_filtered, const uint8_t *color, const float *b_img, int rows, int cols,
                          const float pose[16], const ModelParams &p, int time, float *surfels) {
    auto P = [&](int r, int c) { return pose[r + 4 * c]; };
    auto Draw = [&](int i, int j) { return depth_metric[size_t(j) * cols + i]; };
    auto Dfil = [&](int i, int j) { return depth_filtered[j + size_t(i) * rows]; };
    const size_t cap = size_t(rows) * cols;
    for (size_t q = 0; q < cap * 12; q++) surfels[q] = 0.f;  // the feedback buffers start zero-filled (FeedbackBuffer.cpp:30-32)
    int n_raw = 0, n_fil = 0;
    for (int i = 0; i < cols; i++)        // FeedbackBuffer.cpp:41-49: x outer, y inner
        for (int j = 0; j < rows; j++) {
            FeedbackVertex v;
            if (feedback_vertex(Draw, i, j, rows, cols, p, v)) {  // RAW: position and colour (init_unstable.vert:36,43-45)
                float *s = surfels + size_t(n_raw++) * 12;
                s[0] = P(0, 0) * v.pos.x + P(0, 1) * v.pos.y + P(0, 2) * v.pos.z + P(0, 3);
                s[1] = P(1, 0) * v.pos.x + P(1, 1) * v.pos.y + P(1, 2) * v.pos.z + P(1, 3);
                s[2] = P(2, 0) * v.pos.x + P(2, 1) * v.pos.y + P(2, 2) * v.pos.z + P(2, 3);
                const uint8_t *c = color + (size_t(j) * cols + i) * 3;
                s[4] = float((int(c[0]) << 16) + (int(c[1]) << 8) + int(c[2]));  // encodeColor of the bytes
                s[5] = 1.0f;         // hist weight
                s[6] = 1.0f;         // initialisation time
                s[7] = float(time);  // vertex_feedback.vert:64
            }
            if (feedback_vertex(Dfil, i, j, rows, cols, p, v)) {  // FILTERED: normal, radius, and b as its "colour"
                float *s = surfels + size_t(n_fil++) * 12;
                const int k = int(std::round(b_img[j + size_t(i) * rows] * 255.0f));  // encodeColor -> decodeColor(...).x
                s[3] = float(k & 0xFF) / 255.0f;                                        // init_unstable.vert:38-40
                s[8] = P(0, 0) * v.normal.x + P(0, 1) * v.normal.y + P(0, 2) * v.normal.z;  // :47
                s[9] = P(1, 0) * v.normal.x + P(1, 1) * v.normal.y + P(1, 2) * v.normal.z;
                s[10] = P(2, 0) * v.normal.x + P(2, 1) * v.normal.y + P(2, 2) * v.normal.z;
                s[11] = v.radius;
            }
        }
    // the draw call is sized by the RAW buffer (GlobalModel.cpp:236-237); slots beyond it are not part of the model
    for (size_t q = size_t(n_raw) * 12; q < cap * 12; q++) surfels[q] = 0.f;
    return n_raw;
}
}  // namespace sfo
