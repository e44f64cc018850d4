// Standalone check of ordered_tile_splat against ro_splat (sf_reforder.h): one workgroup, a synthetic level, a rigid warp.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I staticfusion_amd/csrc -o tools/micro/ordered_splat_check tools/micro/ordered_splat_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "sf_device_common.h"

struct Src {
    gptr<const float> d, i;
    LevelCoord lc;
    __device__ __forceinline__ bool load(int v, int u, int idx, float &z, float &xr, float &yr, float &iw) const {
        z = gld(d, idx);
        iw = gld(i, idx);
        xr = coord_x(lc, u, z);
        yr = coord_y(lc, v, z);
        return z != 0.f;
    }
};

__global__ __launch_bounds__(SF_NT) void k(const float *d, const float *in, int rows, int cols, SplatGeom g, LevelCoord lc, long long *acc_d, long long *acc_i,
                                           int *list, int fast, int *ok) {
    __shared__ SplatWin win;
    const int tid = threadIdx.x;
    Src src{as_global(d), as_global(in), lc};
    if (fast) {
        const bool r = ordered_tile_splat(g, lc, rows, cols, src, as_global(acc_d), as_global(acc_i), *(LDS SplatWin *)&win, tid);
        if (tid == 0) *ok = r ? 1 : 0;
        // fast == 2: what ordered_splat() does when the tiles give up in the middle of a level -- the lists, over the cells the tiles left
        if (fast == 2 && !r) ro_splat(g, lc, rows * cols, src, as_global(acc_d), as_global(acc_i), as_global(list), tid);
    } else {
        ro_splat(g, lc, rows * cols, src, as_global(acc_d), as_global(acc_i), as_global(list), tid);
        if (tid == 0) *ok = 1;
    }
}

int main() {
    int bad_total = 0;
    // {rows, cols, roll in milliradians about the optical axis}: the last three leave the tile windows (the fall-back of ordered_splat)
    const int sizes[][3] = {{15, 20, 0}, {30, 40, 0}, {60, 80, 0}, {50, 66, 0}, {37, 51, 0}, {64, 100, 0}, {8, 8, 0}, {30, 40, 300}, {60, 80, 300}, {64, 100, 500}};
    for (auto &sz : sizes) {
        const int rows = sz[0], cols = sz[1], n = rows * cols;
        const float roll = 0.001f * sz[2];
        std::vector<float> d(n), in(n);
        for (int u = 0; u < cols; u++)
            for (int v = 0; v < rows; v++) {
                d[v + u * rows] = (u % 7 == 3 && v % 5 == 1) ? 0.f : 1.5f + 0.3f * std::sin(0.2f * u) + 0.2f * std::cos(0.3f * v);
                in[v + u * rows] = 0.5f + 0.4f * std::sin(0.5f * u + 0.7f * v);
            }
        const float tanh_ = std::tan(0.5f * 1.0908f);
        LevelCoord lc{2.f * tanh_ / float(cols), 0.5f * (cols - 1), 0.5f * (rows - 1), 1.f / float(rows), rows};
        SplatGeom g;
        const float T[12] = {0.9995f, 0.01f, -0.02f, 0.03f, -0.01f, 0.9998f, 0.015f, -0.02f, 0.02f, -0.015f, 0.9996f, 0.04f};
        const float Tr[12] = {std::cos(roll), -std::sin(roll), 0.f, 0.02f, std::sin(roll), std::cos(roll), 0.f, 0.01f, 0.f, 0.f, 1.f, -0.01f};  // rows x, y, depth
        for (int q = 0; q < 12; q++) g.T[q] = roll != 0.f ? Tr[q] : T[q];
        g.f = float(cols) / (2.f * tanh_);
        g.disp_u_i = 0.5f * (cols - 1);
        g.disp_v_i = 0.5f * (rows - 1);
        g.cols_lim = 100 * (cols - 1);
        g.rows_lim = 100 * (rows - 1);
        g.rows_i = rows;
        float *dd, *di;
        long long *ad[2], *ai[2];
        int *list, *ok;
        hipMalloc(&dd, n * 4); hipMalloc(&di, n * 4); hipMalloc(&list, n * 32 * 4); hipMalloc(&ok, 8);
        hipMemcpy(dd, d.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(di, in.data(), n * 4, hipMemcpyHostToDevice);
        std::vector<long long> hd[2], hi[2];
        int okh[2] = {0, 0};
        for (int f = 0; f < 2; f++) {
            hipMalloc(&ad[f], n * 8); hipMalloc(&ai[f], n * 8);
            hipMemset(ad[f], 0xff, n * 8); hipMemset(ai[f], 0xff, n * 8);
            hipLaunchKernelGGL(k, dim3(1), dim3(SF_NT), 0, 0, dd, di, rows, cols, g, lc, ad[f], ai[f], list, f ? (roll != 0.f ? 2 : 1) : 0, ok);
            hipDeviceSynchronize();
            hipMemcpy(&okh[f], ok, 4, hipMemcpyDeviceToHost);
            hd[f].resize(n); hi[f].resize(n);
            hipMemcpy(hd[f].data(), ad[f], n * 8, hipMemcpyDeviceToHost); hipMemcpy(hi[f].data(), ai[f], n * 8, hipMemcpyDeviceToHost);
        }
        int touched[2] = {0, 0}, bad = 0;
        for (int q = 0; q < n; q++) {
            touched[0] += hi[0][q] != 0; touched[1] += hi[1][q] != 0;
            if ((hi[0][q] != 0) != (hi[1][q] != 0) || (hi[0][q] != 0 && hd[0][q] != hd[1][q])) bad++;
        }
        printf("%d x %d roll %.1f: lists touched %d, tiles%s touched %d (tiles returned %d), differing cells %d\n", rows, cols, roll, touched[0],
               roll != 0.f ? " + fall-back" : "", touched[1], okh[1], bad);
        bad_total += bad;
    }
    printf(bad_total ? "FAIL\n" : "OK\n");
    return bad_total != 0;
}
