// sf_smallmath.h — the small dense algebra of the solver, on device.
//
//  * pivoted LDL^T (float) of the 6x6 normal equations (reference FrontEnd.cpp:642) and of the
//    24x24 segmentation system (reference SegmentationBackground.cpp:168): one wave, one matrix
//    row per lane, matrix in LDS.  Every element is updated with the same operation sequence
//    as a scalar unblocked left-looking LDL^T with diagonal pivoting (Eigen's algorithm), so the
//    result does not depend on the lane count.
//  * 6x6 inverse, symmetric 6x6 eigen-decomposition (cyclic Jacobi), SE(3) exp / log, 4x4 and
//    3x3 inverse in double on one lane (reference FrontEnd.cpp:689,719-771,800,1139-1144) — a few
//    thousand flops per outer iteration, off the streaming path.
#pragma once

#include "sf_device_common.h"

// ---------------------------------------------------------------------------------------------
//  wave-parallel pivoted LDL^T.  M is [N][N+1] floats in LDS (lower triangle used).
//  Must be called by all 64 lanes of ONE wave.
// ---------------------------------------------------------------------------------------------
// DPP move that leaves lanes without a source (row edges, masked rows) at the identity of a max over keys (-1)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_i64_keep(long long b) {
    const int lo = __builtin_amdgcn_update_dpp(-1, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(-1, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ long long sf_op_max64(long long a, long long b) {  // unsigned order: -1 is the identity's opposite, see below
    return ((unsigned long long)(a + 1) > (unsigned long long)(b + 1)) ? a : b;     // keys + 1: the identity -1 becomes 0, the smallest
}

__device__ __forceinline__ float readlane_f32(float x, int l) {  // the builtin is int -> int: a float argument would be CONVERTED
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}

// Lane i keeps ROW i of the factor in registers for the whole factorisation -- m[k] = L(row i, elimination step k) -- and
// the rows never move: the symmetric row / column exchanges of the pivoting are book-keeping (`pos`, the position a row
// currently has in the permuted order, decides ties exactly like the in-place algorithm: the first maximum wins), the
// column of the pivot is read from the untouched input matrix, and the factor is stored in its permuted layout at the
// end. Per step: one DPP maximum, k broadcasts (v_readlane) of the pivot row, 2 k multiply-adds per lane. Every number is
// produced by the operation sequence of the unblocked left-looking algorithm (acc += L(i,j) (D_j L(k,j)) for ascending j;
// L(i,k) = (A(i,k) - acc) / D_k), so the factor has the same bits as Eigen's / the oracle's.
template <int N>
__device__ inline bool ldlt_factor_wave(LDS volatile float *Mv, LDS volatile float *temp, LDS volatile int *transp, int lane) {
    (void)temp;
    constexpr int LD = N + 1;
    LDS float *M = (LDS float *)Mv;
    const int row = (lane < N) ? lane : 0;
    float m[N], dstep[N];
#pragma unroll
    for (int j = 0; j < N; j++) m[j] = 0.f;
    float diag = M[row * LD + row];
    int pos = lane;
    bool live = lane < N;
    bool all_zero = false;
#pragma unroll
    for (int k = 0; k < N; k++) {
        // largest |diagonal| among the rows not yet eliminated; the first POSITION wins. One 64-bit key per lane -- |a|
        // (non-negative floats order like their bit patterns) above the complemented position -- and a DPP maximum.
        const float a = live ? fabsf(diag) : 0.f;
        // a NaN never replaces the running maximum of the scalar algorithm (x > best is false), unless it is the first element
        const unsigned abits = (a != a) ? ((pos == k) ? 0xffffffffu : 0u) : __float_as_uint(a);
        long long key = live ? (long long)(((unsigned long long)abits << 32) | (unsigned)(63 - pos)) : -1ll;
        SF_DPP_REDUCE(key, dpp_i64_keep, sf_op_max64)
        const int big = 63 - (int)(__builtin_amdgcn_readlane((int)(key & 0xffffffffll), 63) & 63);  // position of the pivot
        const int p = __ffsll((long long)__ballot(live && pos == big)) - 1;                        // ... and its lane
        if (lane == 0) transp[k] = big;
        if (live && pos == k) pos = big;  // the row at position k changes places with the pivot row
        if (lane == p) pos = k;
        const float x = M[row * LD + p];  // A(row, pivot) of the input matrix (symmetric: both triangles are filled)
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < k; j++) {
            const float tj = dstep[j] * readlane_f32(m[j], p);  // temp[j] = D_j L(k, j)
            acc += m[j] * tj;
        }
        float xm = x;
        if (k > 0) {
            xm = x - acc;
            if (lane == p) diag -= acc;
        }
        const float akk = readlane_f32(diag, p);
        dstep[k] = akk;
        const bool pivot_valid = fabsf(akk) > 0.f;
        if (k == 0 && !pivot_valid) {
            if (lane < N) transp[lane] = lane;
            all_zero = true;
            break;
        }
        m[k] = (live && lane != p) ? (pivot_valid ? xm / akk : xm) : 0.f;
        if (lane == p) live = false;
    }
    __builtin_amdgcn_wave_barrier();  // every read of the input matrix precedes the stores below
    if (!all_zero && lane < N) {      // the factor in its permuted layout: this row sits at position `pos`
#pragma unroll
        for (int j = 0; j < N; j++)
            if (j < pos) M[pos * LD + j] = m[j];
        M[pos * LD + pos] = diag;
    }
    __builtin_amdgcn_wave_barrier();
    return all_zero;
}

// Solves with the factors above. y is [N] floats in LDS holding b on entry and x on exit.
// Lane i keeps y_i, row i of L (forward substitution) and column i of L (backward substitution) in registers -- 2 N
// batched LDS reads -- and the substitutions run on v_readlane broadcasts: every y_i sees the operations of the scalar
// algorithm in the same order (y_i -= L_ij y_j for ascending j, then descending j), without an LDS round trip per step.
template <int N>
__device__ inline void ldlt_solve_wave(LDS volatile const float *Mv, LDS volatile const int *transpv, bool all_zero,
                                       LDS volatile float *y, int lane) {
    constexpr int LD = N + 1;
    const LDS float *M = (const LDS float *)Mv;
    const LDS int *transp = (const LDS int *)transpv;
    const int row = (lane < N) ? lane : 0;
    float lrow[N], lcol[N];
    int tr[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        lrow[j] = M[row * LD + j];  // L(lane, j), j < lane
        lcol[j] = M[j * LD + row];  // L(j, lane), j > lane
        tr[j] = transp[j];
    }
    float yi = y[row];
    const float di = M[row * LD + row];
    // P b
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int t = __builtin_amdgcn_readfirstlane(tr[k]);
        if (t != k) {
            const float yk = readlane_f32(yi, k), yt = readlane_f32(yi, t);
            yi = (lane == k) ? yt : ((lane == t) ? yk : yi);
        }
    }
    if (!all_zero) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            const float yj = readlane_f32(yi, j);
            if (lane > j) yi -= lrow[j] * yj;
        }
    }
    {
        const float d = all_zero ? 0.f : di;
        const float tol = 1.17549435e-38f;  // std::numeric_limits<float>::min(), Eigen's LDLT tolerance
        yi = (fabsf(d) > tol) ? (yi / d) : 0.f;
    }
    if (!all_zero) {
#pragma unroll
        for (int j = N - 1; j >= 0; j--) {
            const float yj = readlane_f32(yi, j);
            if (lane < j) yi -= lcol[j] * yj;
        }
    }
    // P^T x
#pragma unroll
    for (int k = N - 1; k >= 0; k--) {
        const int t = __builtin_amdgcn_readfirstlane(tr[k]);
        if (t != k) {
            const float yk = readlane_f32(yi, k), yt = readlane_f32(yi, t);
            yi = (lane == k) ? yt : ((lane == t) ? yk : yi);
        }
    }
    if (lane < N) y[lane] = yi;
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
//  single-lane double-precision helpers (arrays live in LDS; row-major)
// ---------------------------------------------------------------------------------------------
__device__ inline void inverse_double_lds(LDS double *A, LDS double *Ainv, int n) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ainv[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int c = 0; c < n; c++) {
        int p = c;
        double pv = fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++)
            if (fabs(A[r * n + c]) > pv) {
                pv = fabs(A[r * n + c]);
                p = r;
            }
        if (p != c)
            for (int j = 0; j < n; j++) {
                double t = A[c * n + j];
                A[c * n + j] = A[p * n + j];
                A[p * n + j] = t;
                t = Ainv[c * n + j];
                Ainv[c * n + j] = Ainv[p * n + j];
                Ainv[p * n + j] = t;
            }
        const double inv = 1.0 / A[c * n + c];
        for (int j = 0; j < n; j++) {
            A[c * n + j] *= inv;
            Ainv[c * n + j] *= inv;
        }
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            const double f = A[r * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < n; j++) {
                A[r * n + j] -= f * A[c * n + j];
                Ainv[r * n + j] -= f * Ainv[c * n + j];
            }
        }
    }
}

// 6 x 6 Gauss-Jordan inverse and cyclic Jacobi (symmetric 6 x 6: A is destroyed, diagonal = eigenvalues, V columns =
// eigenvectors) with the n independent element updates of every step spread over lanes 0..n-1 of one wave (called by
// the whole wave; every element sees exactly the operations a one-lane loop -- inverse_double_lds above, the oracle's
// jacobi -- performs, so the results are bit-identical). LDS accesses of one wave execute in program order; the wave barriers keep the compiler
// from moving them across the phases.
__device__ inline void inverse_double_wave6(LDS volatile double *A, LDS volatile double *Ainv, int lane) {
    const int n = 6;
    if (lane < 36) Ainv[lane] = (lane / n == lane % n) ? 1.0 : 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int c = 0; c < n; c++) {
        int p = c;  // pivot search: uniform, every lane reads the same column
        double pv = fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++) {
            const double v = fabs(A[r * n + c]);
            if (v > pv) {
                pv = v;
                p = r;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (p != c && lane < n) {  // lane j swaps column j of rows c and p
            double t = A[c * n + lane];
            A[c * n + lane] = A[p * n + lane];
            A[p * n + lane] = t;
            t = Ainv[c * n + lane];
            Ainv[c * n + lane] = Ainv[p * n + lane];
            Ainv[p * n + lane] = t;
        }
        __builtin_amdgcn_wave_barrier();
        const double inv = 1.0 / A[c * n + c];
        __builtin_amdgcn_wave_barrier();
        if (lane < n) {
            A[c * n + lane] *= inv;
            Ainv[c * n + lane] *= inv;
        }
        __builtin_amdgcn_wave_barrier();
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            const double f = A[r * n + c];  // read by every lane BEFORE lane c overwrites it below
            __builtin_amdgcn_wave_barrier();
            if (f != 0.0 && lane < n) {
                A[r * n + lane] -= f * A[c * n + lane];
                Ainv[r * n + lane] -= f * Ainv[c * n + lane];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__device__ inline void jacobi_eig6_wave(LDS volatile double *A, LDS volatile double *V, int lane) {
    const int n = 6;
    if (lane < 36) V[lane] = (lane / n == lane % n) ? 1.0 : 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, diag = 0.0;  // uniform: every lane sums the same elements in the same order
        for (int i = 0; i < n; i++) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-60 || off <= 1e-34 * diag) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                __builtin_amdgcn_wave_barrier();
                if (lane < n) {  // columns p, q of row `lane`; the eigenvector update is independent of A
                    const double akp = A[lane * n + p], akq = A[lane * n + q];
                    A[lane * n + p] = c * akp - s * akq;
                    A[lane * n + q] = s * akp + c * akq;
                    const double vkp = V[lane * n + p], vkq = V[lane * n + q];
                    V[lane * n + p] = c * vkp - s * vkq;
                    V[lane * n + q] = s * vkp + c * vkq;
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < n) {  // rows p, q of column `lane`
                    const double apk = A[p * n + lane], aqk = A[q * n + lane];
                    A[p * n + lane] = c * apk - s * aqk;
                    A[q * n + lane] = s * apk + c * aqk;
                }
                __builtin_amdgcn_wave_barrier();
            }
    }
}

// ---------------------------------------------------------------------------------------------
//  6 x 6 double-precision algebra with ONE MATRIX ELEMENT PER LANE (lane 6 i + j holds element (i, j) in a register,
//  36 lanes of one wave): a step that touches whole rows / columns is one instruction, partners are fetched with
//  ds_bpermute, nothing goes through LDS memory.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double shfl_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// Gauss-Jordan inverse with partial pivoting: a = A(i, j) on entry, ai = A^-1(i, j) on exit. Every element sees exactly
// the operations of inverse_double_lds (the oracle's inverse_double), so the result is bit-identical to it.
__device__ inline void inverse6_lanes(double &a, double &ai, int lane) {
    const int l = (lane < 36) ? lane : 0, i = l / 6, j = l - 6 * i;
    ai = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        int p = c;  // pivot search: uniform, every lane looks at the same column in the same order
        double pv = fabs(shfl_f64(a, 6 * c + c));
#pragma unroll
        for (int r = c + 1; r < 6; r++) {
            const double v = fabs(shfl_f64(a, 6 * r + c));
            if (v > pv) {
                pv = v;
                p = r;
            }
        }
        const int prow = (i == c) ? p : ((i == p) ? c : i);  // rows c and p change places
        a = shfl_f64(a, 6 * prow + j);
        ai = shfl_f64(ai, 6 * prow + j);
        const double inv = 1.0 / shfl_f64(a, 6 * c + c);
        if (i == c) {
            a *= inv;
            ai *= inv;
        }
        const double f = shfl_f64(a, 6 * i + c);  // A(i, c) before this column is eliminated
        const double pc = shfl_f64(a, 6 * c + j), pic = shfl_f64(ai, 6 * c + j);
        if (i != c && f != 0.0) {
            a -= f * pc;
            ai -= f * pic;
        }
    }
}

// Jacobi eigen-decomposition of a symmetric 6 x 6: a = A(i, j) on entry; on exit the diagonal lanes hold the eigenvalues and
// v = V(i, j) the eigenvectors (columns). Round-robin ordering: the 15 index pairs of a sweep in 5 rounds of 3 DISJOINT
// pairs, whose rotations commute and are applied together -- every lane computes the rotation of its column's pair itself
// (three partners' elements, the formulas of the cyclic algorithm), so a round costs what one rotation cost. The result is
// an eigen-decomposition to the same accuracy as the cyclic order's (the oracle's), not the same bits: eigenvalues come out
// in another order, vectors may change sign; the motion filter that consumes them (FrontEnd.cpp:726-760) is invariant to both.
__device__ inline void jacobi6_lanes(double &a, double &v, int lane) {
    const int l = (lane < 36) ? lane : 0, i = l / 6, j = l - 6 * i;
    const bool in = lane < 36;
    v = (i == j) ? 1.0 : 0.0;
    // partner of index k in round r (tournament schedule): packed 3 bits per index
    const unsigned sched[5] = {05 | 04 << 3 | 03 << 6 | 02 << 9 | 01 << 12 | 00 << 15, 04 | 02 << 3 | 01 << 6 | 05 << 9 | 00 << 12 | 03 << 15,
                               03 | 05 << 3 | 04 << 6 | 00 << 9 | 02 << 12 | 01 << 15, 02 | 03 << 3 | 00 << 6 | 01 << 9 | 05 << 12 | 04 << 15,
                               01 | 00 << 3 | 05 << 6 | 04 << 9 | 03 << 12 | 02 << 15};
    for (int sweep = 0; sweep < 60; sweep++) {
        const double sq = a * a;
        const double off = wave_sum_f64((in && i < j) ? sq : 0.0), diag = wave_sum_f64((in && i == j) ? sq : 0.0);
        if (off <= 1e-60 || off <= 1e-34 * diag) break;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const int jp = (int)((sched[r] >> (3 * j)) & 7u), ip = (int)((sched[r] >> (3 * i)) & 7u);
            const int p = min(j, jp), q = max(j, jp);  // the pair this lane's COLUMN belongs to
            const double app = shfl_f64(a, 6 * p + p), aqq = shfl_f64(a, 6 * q + q), apq = shfl_f64(a, 6 * p + q);
            double c = 1.0, sn = 0.0;
            const bool rot = apq != 0.0;
            if (rot) {
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                c = 1.0 / sqrt(t * t + 1.0);
                sn = t * c;
            }
            // columns p, q of every row (A and V)
            const double ap = shfl_f64(a, 6 * i + jp), vp = shfl_f64(v, 6 * i + jp);
            if (rot) {
                a = (j == p) ? (c * a - sn * ap) : (sn * ap + c * a);
                v = (j == p) ? (c * v - sn * vp) : (sn * vp + c * v);
            }
            // rows of the pair this lane's ROW belongs to: its rotation is the one lane `i` (row 0, column i) computed
            const double ci = shfl_f64(c, i), si = shfl_f64(sn, i);
            const bool roti = __shfl((int)rot, i, 64) != 0;
            const double ar = shfl_f64(a, 6 * ip + j);
            if (roti) a = (i < ip) ? (ci * a - si * ar) : (si * ar + ci * a);
        }
    }
}

__device__ inline void skew_sq_d(const double w[3], double K[9], double K2[9]) {
    K[0] = 0;     K[1] = -w[2]; K[2] = w[1];
    K[3] = w[2];  K[4] = 0;     K[5] = -w[0];
    K[6] = -w[1]; K[7] = w[0];  K[8] = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += K[i * 3 + k] * K[k * 3 + j];
            K2[i * 3 + j] = s;
        }
}

// SE(3) exponential, twist (v, w) -> row-major 4x4
__device__ inline void se3_exp_d(const double xi[6], double T[16]) {
    const double *v = xi, *w = xi + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th = sqrt(th2);
    double a, b, c;
    if (th < 1e-5) {
        a = 1.0 - th2 / 6.0;
        b = 0.5 - th2 / 24.0;
        c = 1.0 / 6.0 - th2 / 120.0;
    } else {
        a = sin(th) / th;
        b = (1.0 - cos(th)) / th2;
        c = (th - sin(th)) / (th2 * th);
    }
    double K[9], K2[9];
    skew_sq_d(w, K, K2);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double t = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double I = (i == j) ? 1.0 : 0.0;
            T[i * 4 + j] = I + a * K[i * 3 + j] + b * K2[i * 3 + j];
            t += (I + b * K[i * 3 + j] + c * K2[i * 3 + j]) * v[j];
        }
        T[i * 4 + 3] = t;
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}

// SE(3) logarithm of a rigid row-major 4x4 -> twist (v, w)
__device__ inline void se3_log_d(const double T[16], double xi[6]) {
    const double rx = 0.5 * (T[2 * 4 + 1] - T[1 * 4 + 2]);
    const double ry = 0.5 * (T[0 * 4 + 2] - T[2 * 4 + 0]);
    const double rz = 0.5 * (T[1 * 4 + 0] - T[0 * 4 + 1]);
    const double s = sqrt(rx * rx + ry * ry + rz * rz);
    const double cth = 0.5 * (T[0] + T[5] + T[10] - 1.0);
    const double th = atan2(s, cth);
    double w[3];
    if (s < 1e-9) {
        const double k = 1.0 + th * th / 6.0;
        w[0] = k * rx; w[1] = k * ry; w[2] = k * rz;
    } else {
        const double k = th / s;
        w[0] = k * rx; w[1] = k * ry; w[2] = k * rz;
    }
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double tha = sqrt(th2);
    double d;
    if (tha < 1e-4)
        d = 1.0 / 12.0 + th2 / 720.0;
    else
        d = (1.0 - (tha * sin(tha)) / (2.0 * (1.0 - cos(tha)))) / th2;
    double K[9], K2[9];
    skew_sq_d(w, K, K2);
    const double t[3] = {T[3], T[7], T[11]};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double vv = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double I = (i == j) ? 1.0 : 0.0;
            vv += (I - 0.5 * K[i * 3 + j] + d * K2[i * 3 + j]) * t[j];
        }
        xi[i] = vv;
    }
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

// column-major float 4x4 helpers --------------------------------------------------------------
// twist = vee(log(T)) for a column-major float T
template <class P>
__device__ inline void log_twist_cm(P Tcm, float out[6]) {
    double Td[16], xi[6];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) Td[r * 4 + c] = (double)Tcm[r + 4 * c];
    se3_log_d(Td, xi);
#pragma unroll
    for (int i = 0; i < 6; i++) out[i] = (float)xi[i];
}

// C = A * B, column-major float, inner sum left to right (reference FrontEnd.cpp:766)
template <class PB>
__device__ inline void mul4_cm(const float *A, PB B, float *C) {
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float s = A[r + 0] * B[0 + 4 * c];
            s += A[r + 4] * B[1 + 4 * c];
            s += A[r + 8] * B[2 + 4 * c];
            s += A[r + 12] * B[3 + 4 * c];
            C[r + 4 * c] = s;
        }
}

// inverse of a column-major float 4x4 through double Gauss-Jordan (scratch: 32 doubles in LDS)
template <class PT, class PO>
__device__ inline void inverse4_cm(PT Tcm, PO out, LDS double *scratch) {
    LDS double *A = scratch, *Ai = scratch + 16;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) A[r * 4 + c] = (double)Tcm[r + 4 * c];
    inverse_double_lds(A, Ai, 4);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[r + 4 * c] = (float)Ai[r * 4 + c];
}
