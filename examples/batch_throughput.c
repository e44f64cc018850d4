/* batch_throughput.c — plain C against include/sf.h: many independent RGB-D streams on one MI355X, frames fed from
 * host memory with the upload of step k+1 overlapping the solve of step k (frame-to-frame prediction on the device).
 *
 *   gcc -std=c11 -O2 -Iinclude examples/batch_throughput.c -o batch_throughput
 *       -Lstaticfusion_amd/csrc -lsf_hip -Wl,-rpath,$PWD/staticfusion_amd/csrc -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lm   (one line)
 *   ./batch_throughput [streams] [steps]
 *
 * The images are a synthetic ramp (this example shows the call sequence, not a dataset). */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "sf.h"

#define CHECK(call)                                                       \
    do {                                                                  \
        int rc_ = (call);                                                 \
        if (rc_ != SF_OK) {                                               \
            fprintf(stderr, "%s failed: %d: %s\n", #call, rc_, sf_last_error()); \
            return 1;                                                     \
        }                                                                 \
    } while (0)

static void fill(float *depth, float *inten, int batch, int rows, int cols, int step) {
    for (int b = 0; b < batch; b++)
        for (int u = 0; u < cols; u++)
            for (int v = 0; v < rows; v++) {
                const size_t o = ((size_t)b * cols + u) * rows + v; /* column-major images, [batch][cols][rows] */
                const float x = (u + 0.35f * step) / cols, y = (float)v / rows;
                depth[o] = 1.5f + 0.8f * x + 0.3f * y + 0.1f * sinf(9.f * x + b);
                inten[o] = 0.5f + 0.25f * sinf(23.f * x + 7.f * y) + 0.2f * cosf(17.f * y - 5.f * x);
            }
}

int main(int argc, char **argv) {
    const int batch = argc > 1 ? atoi(argv[1]) : 1024, steps = argc > 2 ? atoi(argv[2]) : 6;
    const int rows = 240, cols = 320;
    sf_params p;
    sf_default_params(&p); /* the reference drivers' parameter block */
    p.kb = 1.05f;
    sf_handle *h = NULL;
    CHECK(sf_create(&p, rows, cols, batch, 0, &h));
    printf("backend %s, %d streams\n", sf_backend(), batch);

    const size_t n = (size_t)batch * rows * cols;
    float *host[2][2]; /* two page-locked frame sets: one being uploaded, one being refilled */
    for (int s = 0; s < 2; s++)
        for (int c = 0; c < 2; c++) CHECK(sf_alloc_pinned(n * sizeof(float), (void **)&host[s][c]));

    fill(host[0][0], host[0][1], batch, rows, cols, 0);
    CHECK(sf_upload_current_async(h, host[0][0], host[0][1]));
    CHECK(sf_commit_upload(h));
    CHECK(sf_current_to_prediction(h)); /* bootstrap: the first frame is the prediction of the second */
    CHECK(sf_push_history(h, 0));
    fill(host[1][0], host[1][1], batch, rows, cols, 1);
    CHECK(sf_upload_current_async(h, host[1][0], host[1][1]));

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 1; k <= steps; k++) {
        CHECK(sf_commit_upload(h));          /* frame k is in place once its copy has landed (device-side wait) */
        CHECK(sf_process_frame(h, k));       /* pyramid + solver (+ 5-frame residuals) + b image + ring push, one launch */
        CHECK(sf_current_to_prediction(h));  /* frame-to-frame mode */
        fill(host[(k + 1) & 1][0], host[(k + 1) & 1][1], batch, rows, cols, k + 1); /* CPU work overlaps the GPU */
        CHECK(sf_upload_current_async(h, host[(k + 1) & 1][0], host[(k + 1) & 1][1]));
    }
    CHECK(sf_synchronize(h));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double sec = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    float T[16];
    CHECK(sf_get_T(h, 0, T));
    int64_t frames, irls, outer, pix;
    CHECK(sf_get_counters(h, &frames, &irls, &outer, &pix));
    printf("%d steps: %.1f ms/step, %.0f frames/s incl. host fill and PCIe; stream 0 moved by (%.4f %.4f %.4f) m; %lld IRLS iterations\n",
           steps, 1e3 * sec / steps, batch * steps / sec, T[12], T[13], T[14], (long long)irls);
    CHECK(sf_commit_upload(h));
    CHECK(sf_synchronize(h));
    for (int s = 0; s < 2; s++)
        for (int c = 0; c < 2; c++) sf_free_pinned(host[s][c]);
    sf_destroy(h);
    return 0;
}
