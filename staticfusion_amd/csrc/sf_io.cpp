// sf_io.cpp — libsf_io.so: the on-disk formats either side of the solver path (include/sf_io.h,
// SURVEY.md §8(f) rank 2). Host-only C++ + zlib; OpenCV / libpng / MRPT are not needed.
//   association file  reference FrontEnd.cpp:183-214 (StaticFusion::loadAssoc)
//   PNG frames        what cv::imread hands to FrontEnd.cpp:220,240 (PNG: ISO/IEC 15948)
//   trajectory file   reference Utils/Datasets.cpp:252-265, Reconstruction.cpp:53-81
#include "../../include/sf_io.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
extern "C" const char *sf_io_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------------
//  association file
// ------------------------------------------------------------------------------------------------
struct sf_io_assoc {
    std::vector<double> timestamps;
    std::vector<std::string> filesDepth, filesColor;
};

extern "C" int sf_io_assoc_load(const char *dir, const char *assoc_file, sf_io_assoc **out) {
    if (!dir || !assoc_file || !out) return fail(SF_IO_ERR_ARG, "null argument");
    const std::string d(dir), assocPath = d + assoc_file;  // :186
    std::ifstream assocIn(assocPath.c_str());
    if (!assocIn.is_open()) return fail(SF_IO_ERR_FILE, "cannot open " + assocPath);  // :193-194 (the reference returns false)
    sf_io_assoc *a = new sf_io_assoc;
    std::string line;
    while (std::getline(assocIn, line)) {  // :197
        if (line.empty() || line.compare(0, 1, "#") == 0) continue;  // :199-200
        std::istringstream iss(line);
        double timestampDepth, timestampColor;
        std::string fileDepth, fileColor;
        if (!(iss >> timestampColor >> fileColor >> timestampDepth >> fileDepth)) break;  // :204-205
        a->timestamps.push_back(timestampDepth);    // :207
        a->filesDepth.push_back(d + fileDepth);     // :208
        a->filesColor.push_back(d + fileColor);     // :209
    }
    *out = a;
    return SF_IO_OK;
}
extern "C" int sf_io_assoc_count(const sf_io_assoc *a) { return a ? (int)a->timestamps.size() : 0; }
extern "C" int sf_io_assoc_entry(const sf_io_assoc *a, int i, double *timestamp, const char **depth_path, const char **color_path) {
    if (!a || i < 0 || i >= (int)a->timestamps.size()) return fail(SF_IO_ERR_ARG, "index out of range");
    if (timestamp) *timestamp = a->timestamps[i];
    if (depth_path) *depth_path = a->filesDepth[i].c_str();
    if (color_path) *color_path = a->filesColor[i].c_str();
    return SF_IO_OK;
}
extern "C" void sf_io_assoc_free(sf_io_assoc *a) { delete a; }

// ------------------------------------------------------------------------------------------------
//  PNG decoder: chunks -> zlib inflate -> scanline unfilter -> samples
// ------------------------------------------------------------------------------------------------
namespace {
struct Png {
    int width = 0, height = 0, bit_depth = 0, color_type = 0, channels = 0;
    std::vector<uint8_t> palette;  // RGB triples
    std::vector<uint8_t> pix;      // height x stride, unfiltered, big-endian samples
    size_t stride = 0;
};

inline uint32_t be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

int parse_png(const uint8_t *buf, size_t size, Png &png) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (!buf || size < 8 || std::memcmp(buf, sig, 8) != 0) return fail(SF_IO_ERR_FORMAT, "not a PNG file");
    size_t pos = 8;
    std::vector<uint8_t> idat;
    bool have_ihdr = false, have_iend = false;
    int interlace = 0;
    while (pos + 12 <= size && !have_iend) {
        const uint32_t len = be32(buf + pos);
        const uint8_t *type = buf + pos + 4, *data = buf + pos + 8;
        if (len > size || pos + 12 + len > size) return fail(SF_IO_ERR_FORMAT, "truncated PNG chunk");
        const uint32_t crc = be32(data + len);
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != crc) return fail(SF_IO_ERR_FORMAT, "PNG chunk CRC mismatch");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return fail(SF_IO_ERR_FORMAT, "bad IHDR");
            png.width = (int)be32(data);
            png.height = (int)be32(data + 4);
            png.bit_depth = data[8];
            png.color_type = data[9];
            if (data[10] != 0 || data[11] != 0) return fail(SF_IO_ERR_FORMAT, "unknown PNG compression / filter method");
            interlace = data[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            png.palette.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            have_iend = true;
        }  // ancillary chunks (tRNS, gAMA, pHYs, tEXt, ...) do not change what cv::imread returns here
        pos += 12 + len;
    }
    if (!have_ihdr || !have_iend) return fail(SF_IO_ERR_FORMAT, "PNG without IHDR / IEND");
    if (png.width <= 0 || png.height <= 0 || png.width > (1 << 16) || png.height > (1 << 16)) return fail(SF_IO_ERR_FORMAT, "bad PNG size");
    if (interlace != 0) return fail(SF_IO_ERR_UNSUPPORTED, "Adam7-interlaced PNG");
    switch (png.color_type) {
        case 0: png.channels = 1; break;
        case 2: png.channels = 3; break;
        case 3: png.channels = 1; break;
        case 4: png.channels = 2; break;
        case 6: png.channels = 4; break;
        default: return fail(SF_IO_ERR_FORMAT, "bad PNG colour type");
    }
    const int bd = png.bit_depth;
    const bool bd_ok = (png.color_type == 0 && (bd == 1 || bd == 2 || bd == 4 || bd == 8 || bd == 16)) ||
                       (png.color_type == 3 && (bd == 1 || bd == 2 || bd == 4 || bd == 8)) ||
                       ((png.color_type == 2 || png.color_type == 4 || png.color_type == 6) && (bd == 8 || bd == 16));
    if (!bd_ok) return fail(SF_IO_ERR_FORMAT, "bad PNG bit depth");
    if (png.color_type == 3 && (png.palette.empty() || png.palette.size() % 3)) return fail(SF_IO_ERR_FORMAT, "palette PNG without PLTE");
    const size_t bits_pp = size_t(png.channels) * bd;
    png.stride = (size_t(png.width) * bits_pp + 7) / 8;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;  // filter distance in bytes
    std::vector<uint8_t> raw((png.stride + 1) * png.height);
    uLongf raw_len = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || raw_len != raw.size()) return fail(SF_IO_ERR_FORMAT, "PNG image data does not inflate to the declared size");
    png.pix.assign(png.stride * png.height, 0);
    std::vector<uint8_t> zero(png.stride, 0);
    for (int y = 0; y < png.height; y++) {
        const uint8_t ft = raw[(png.stride + 1) * y];
        const uint8_t *src = &raw[(png.stride + 1) * y + 1];
        uint8_t *cur = &png.pix[png.stride * y];
        const uint8_t *up = y ? &png.pix[png.stride * (y - 1)] : zero.data();
        const size_t n = png.stride, lead = bpp < n ? bpp : n;  // the first pixel has no left neighbour
        switch (ft) {
            case 0: std::memcpy(cur, src, n); break;
            case 1:
                for (size_t x = 0; x < lead; x++) cur[x] = src[x];
                for (size_t x = lead; x < n; x++) cur[x] = uint8_t(src[x] + cur[x - bpp]);
                break;
            case 2:
                for (size_t x = 0; x < n; x++) cur[x] = uint8_t(src[x] + up[x]);
                break;
            case 3:
                for (size_t x = 0; x < lead; x++) cur[x] = uint8_t(src[x] + (up[x] >> 1));
                for (size_t x = lead; x < n; x++) cur[x] = uint8_t(src[x] + ((cur[x - bpp] + up[x]) >> 1));
                break;
            case 4:  // Paeth
                for (size_t x = 0; x < lead; x++) cur[x] = uint8_t(src[x] + up[x]);  // a = c = 0 -> predictor b
                for (size_t x = lead; x < n; x++) {
                    const int a = cur[x - bpp], b = up[x], cc = up[x - bpp];
                    const int p = a + b - cc, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - cc);
                    cur[x] = uint8_t(src[x] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : cc)));
                }
                break;
            default: return fail(SF_IO_ERR_FORMAT, "bad PNG filter type");
        }
    }
    return SF_IO_OK;
}

// sample s (channel ch) of pixel x in an unfiltered row; < 8-bit samples are returned as stored (not scaled)
inline unsigned sample(const Png &p, const uint8_t *row, int x, int ch) {
    const int bd = p.bit_depth;
    if (bd == 8) return row[size_t(x) * p.channels + ch];
    if (bd == 16) {
        const uint8_t *q = row + (size_t(x) * p.channels + ch) * 2;
        return (unsigned(q[0]) << 8) | q[1];
    }
    const int per_byte = 8 / bd, idx = x;  // 1 channel only
    const uint8_t byte = row[idx / per_byte];
    const int shift = 8 - bd * (idx % per_byte + 1);
    return (byte >> shift) & ((1u << bd) - 1u);
}

int read_file(const char *path, std::vector<uint8_t> &buf) {
    if (!path) return fail(SF_IO_ERR_ARG, "null path");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(SF_IO_ERR_FILE, std::string("cannot open ") + path);
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? size_t(n) : 0);
    const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size()) return fail(SF_IO_ERR_FILE, std::string("short read on ") + path);
    return SF_IO_OK;
}
}  // namespace

extern "C" int sf_io_decode_color(const uint8_t *buf, size_t size, uint8_t **bgr, int *rows, int *cols) {
    if (!bgr || !rows || !cols) return fail(SF_IO_ERR_ARG, "null output");
    Png p;
    if (int e = parse_png(buf, size, p)) return e;
    uint8_t *out = (uint8_t *)std::malloc(size_t(p.width) * p.height * 3);
    if (!out) return fail(SF_IO_ERR_ARG, "out of memory");
    const unsigned maxv = (1u << (p.bit_depth < 8 ? p.bit_depth : 8)) - 1u;
    for (int y = 0; y < p.height; y++) {
        const uint8_t *row = &p.pix[p.stride * y];
        uint8_t *o = out + size_t(y) * p.width * 3;
        for (int x = 0; x < p.width; x++, o += 3) {
            auto s8 = [&](int ch) -> uint8_t {  // 16-bit samples keep the high byte, < 8-bit grey is scaled to 0..255
                const unsigned v = sample(p, row, x, ch);
                if (p.bit_depth == 16) return uint8_t(v >> 8);
                if (p.bit_depth < 8 && p.color_type == 0) return uint8_t(v * 255u / maxv);
                return uint8_t(v);
            };
            switch (p.color_type) {
                case 0:
                case 4: o[0] = o[1] = o[2] = s8(0); break;
                case 2:
                case 6: o[0] = s8(2); o[1] = s8(1); o[2] = s8(0); break;  // R G B -> B G R
                case 3: {
                    const unsigned idx = sample(p, row, x, 0);
                    if (idx * 3 + 2 >= p.palette.size()) {
                        std::free(out);
                        return fail(SF_IO_ERR_FORMAT, "palette index out of range");
                    }
                    o[0] = p.palette[idx * 3 + 2]; o[1] = p.palette[idx * 3 + 1]; o[2] = p.palette[idx * 3];
                    break;
                }
            }
        }
    }
    *bgr = out;
    *rows = p.height;
    *cols = p.width;
    return SF_IO_OK;
}

extern "C" int sf_io_decode_depth16(const uint8_t *buf, size_t size, uint16_t **depth, int *rows, int *cols) {
    if (!depth || !rows || !cols) return fail(SF_IO_ERR_ARG, "null output");
    Png p;
    if (int e = parse_png(buf, size, p)) return e;
    if (p.color_type != 0 || (p.bit_depth != 16 && p.bit_depth != 8))
        return fail(SF_IO_ERR_UNSUPPORTED, "depth image must be 8- or 16-bit monochrome (reference README.md:84-87)");
    uint16_t *out = (uint16_t *)std::malloc(size_t(p.width) * p.height * 2);
    if (!out) return fail(SF_IO_ERR_ARG, "out of memory");
    for (int y = 0; y < p.height; y++) {
        const uint8_t *row = &p.pix[p.stride * y];
        for (int x = 0; x < p.width; x++) out[size_t(y) * p.width + x] = (uint16_t)sample(p, row, x, 0);
    }
    *depth = out;
    *rows = p.height;
    *cols = p.width;
    return SF_IO_OK;
}

extern "C" int sf_io_imread_color(const char *path, uint8_t **bgr, int *rows, int *cols) {
    std::vector<uint8_t> buf;
    if (int e = read_file(path, buf)) return e;
    return sf_io_decode_color(buf.data(), buf.size(), bgr, rows, cols);
}
extern "C" int sf_io_imread_depth16(const char *path, uint16_t **depth, int *rows, int *cols) {
    std::vector<uint8_t> buf;
    if (int e = read_file(path, buf)) return e;
    return sf_io_decode_depth16(buf.data(), buf.size(), depth, rows, cols);
}
extern "C" void sf_io_free(void *p) { std::free(p); }

// ------------------------------------------------------------------------------------------------
//  poses and the trajectory file
// ------------------------------------------------------------------------------------------------
extern "C" void sf_io_pose_compose(const float A[16], const float B[16], float out[16]) {
    float r[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            float s = A[i] * B[4 * j];
            for (int k = 1; k < 4; k++) s += A[i + 4 * k] * B[k + 4 * j];
            r[i + 4 * j] = s;
        }
    std::memcpy(out, r, sizeof r);
}

// Eigen::Quaternionf(Matrix3f) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other,3,3>); m is column-major 4x4
static void quat_from_rotation(const float *m, float q[4] /* x y z w */) {
    auto M = [&](int r, int c) { return m[r + 4 * c]; };
    float t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0.f) {
        t = std::sqrt(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (M(2, 1) - M(1, 2)) * t;
        q[1] = (M(0, 2) - M(2, 0)) * t;
        q[2] = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (M(k, j) - M(j, k)) * t;
        q[j] = (M(j, i) + M(i, j)) * t;
        q[k] = (M(k, i) + M(i, k)) * t;
    }
}

extern "C" int sf_io_trajectory_line(double timestamp, const float pose[16], int rotate_by_z, char *buf, size_t buf_size) {
    if (!pose || !buf) return fail(SF_IO_ERR_ARG, "null argument");
    float P[16];
    char ts[64];
    if (rotate_by_z) {
        // rotateByZ = AngleAxisf(M_PI, UnitZ()).toRotationMatrix() (Datasets.cpp:57-59): in float, sin(pi) = -8.74e-8
        const float s = std::sin(float(M_PI)), c = std::cos(float(M_PI));
        float Rz[16] = {0};
        Rz[0] = c;       Rz[1] = s;   // column 0: ( c,  s, 0)
        Rz[4] = -s;      Rz[5] = c;   // column 1: (-s,  c, 0)
        Rz[10] = (1.f - c) * 1.f + c;  // (1 - c) * z * z + c
        Rz[15] = 1.f;
        sf_io_pose_compose(pose, Rz, P);                   // convertedPose = currPose * rotateByZ (:256)
        std::snprintf(ts, sizeof ts, "%.04f", timestamp);  // :261
    } else {
        std::memcpy(P, pose, sizeof P);
        std::snprintf(ts, sizeof ts, "%.6f", timestamp);   // Reconstruction.cpp:66 setprecision(6) << fixed
    }
    float q[4];
    quat_from_rotation(P, q);
    std::ostringstream os;  // default float formatting of std::ostream, as `f_res << float` in the reference
    os << ts << " " << P[12] << " " << P[13] << " " << P[14] << " " << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << "\n";
    const std::string line = os.str();
    if (line.size() + 1 > buf_size) return fail(SF_IO_ERR_ARG, "buffer too small");
    std::memcpy(buf, line.c_str(), line.size() + 1);
    return (int)line.size();
}

extern "C" int sf_io_save_ply(const char *path, const float *surfels, int count, float conf_threshold) {
    if (!path || (!surfels && count > 0) || count < 0) return fail(SF_IO_ERR_ARG, "bad argument");
    int valid = 0;
    for (int i = 0; i < count; i++)
        if (surfels[size_t(i) * 12 + 3] > conf_threshold) valid++;  // Reconstruction.cpp:371-379
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return fail(SF_IO_ERR_FILE, std::string("cannot open ") + path);
    std::ostringstream hd;  // :382-401
    hd << "ply" << "\nformat " << "binary_little_endian" << " 1.0" << "\nelement vertex " << valid
       << "\nproperty float x\nproperty float y\nproperty float z"
       << "\nproperty uchar red\nproperty uchar green\nproperty uchar blue"
       << "\nproperty float nx\nproperty float ny\nproperty float nz"
       << "\nproperty float radius" << "\nend_header\n";
    const std::string h = hd.str();
    bool ok = std::fwrite(h.data(), 1, h.size(), f) == h.size();
    for (int i = 0; i < count && ok; i++) {
        const float *s = surfels + size_t(i) * 12;
        if (!(s[3] > conf_threshold)) continue;  // :413
        unsigned char rec[3 * 4 + 3 + 4 * 4];
        std::memcpy(rec, s, 12);  // x y z
        const int col = int(s[4]);  // :432-434
        rec[12] = (unsigned char)(col >> 16 & 0xFF);
        rec[13] = (unsigned char)(col >> 8 & 0xFF);
        rec[14] = (unsigned char)(col & 0xFF);
        const float n[4] = {s[8] * -1.f, s[9] * -1.f, s[10] * -1.f, s[11]};  // :418-420, radius unchanged
        std::memcpy(rec + 15, n, 16);
        ok = std::fwrite(rec, 1, sizeof rec, f) == sizeof rec;
    }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) return fail(SF_IO_ERR_FILE, std::string("write failed: ") + path);
    return valid;
}
