// Microbenchmark: what the linearisation's record stores cost on MI355X (write-through L2), pattern by pattern.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/strip_stores tools/micro/strip_stores.hip && tools/micro/strip_stores
// A "stream" is a column-major float image of 240 x 320 (column pitch 960 B), PLANES planes each. A wave owns one strip of rows
// and sweeps the columns, storing one float per lane, plane and column -- like solve_linearise_strips:
//   mode 0: strips of 62 rows at row 62 k, lanes 1 .. 62 store (the tree: 248 B runs that start at multiples of 248 B)
//   mode 1: strips of 64 rows at row 64 k, every lane stores (the last strip of a column 48 rows): 256 B runs on 64 B boundaries
//   mode 2: strips of 62 rows, but all 64 lanes store (rows 62 k - 1 .. 62 k + 62: overlapping runs, 256 B, misaligned)
//   mode 3: strips of 64 rows at row 64 k, lanes 1 .. 62 store (aligned start, two lanes masked)
// Reported: useful GB/s (bytes of the rows a strip owns).
#include <hip/hip_runtime.h>
#include <cstdio>
#define ROWS 240
#define COLS 320
#define PLANES 6
template <int MODE>
__global__ __launch_bounds__(256) void k(float *base, int streams) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int own = (MODE == 1 || MODE == 3) ? 64 : 62;
    const int strips = (ROWS + own - 1) / own;  // 4
    for (int s = blockIdx.x; s < streams; s += gridDim.x) {
        float *img = base + (size_t)s * PLANES * ROWS * COLS;
        for (int strip = wave; strip < strips; strip += 4) {
            int v;
            bool st;
            if (MODE == 0) { v = strip * 62 - 1 + lane; st = lane >= 1 && lane <= 62 && v < ROWS; }
            if (MODE == 1) { v = strip * 64 + lane; st = v < ROWS; }
            if (MODE == 2) { v = strip * 62 - 1 + lane; st = v >= 0 && v < ROWS; }
            if (MODE == 3) { v = strip * 64 + lane; st = lane >= 1 && lane <= 62 && v < ROWS; }
            for (int u = 0; u < COLS; u++) {
                if (st) {
#pragma unroll
                    for (int p = 0; p < PLANES; p++) img[(size_t)p * ROWS * COLS + v + u * ROWS] = (float)(u + p);
                }
            }
        }
    }
}
template <int MODE>
void run(const char *name, float *buf, int streams) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<1280, 256>>>(buf, streams); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < 3; r++) k<MODE><<<1280, 256>>>(buf, streams); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    printf("%-70s %7.1f GB/s\n", name, (double)streams * PLANES * ROWS * COLS * 4 / ms / 1e6);
}
int main() {
    const int streams = 2560;  // 2560 x 6 x 307200 B = 4.7 GB
    float *buf; hipMalloc(&buf, (size_t)streams * PLANES * ROWS * COLS * 4);
    run<0>("62-row strips, lanes 1..62 store (the tree)", buf, streams);
    run<1>("64-row strips on 64-row boundaries, every lane stores", buf, streams);
    run<2>("62-row strips, all 64 lanes store (overlapping, misaligned)", buf, streams);
    run<3>("64-row strips on 64-row boundaries, lanes 1..62 store", buf, streams);
    run<0>("62-row strips, lanes 1..62 store (again)", buf, streams);
    run<1>("64-row strips, every lane stores (again)", buf, streams);
    return 0;
}
