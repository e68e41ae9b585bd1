// Diagnostic: HBM read rate of the fused-backward access pattern (384 rows, 128-byte pieces per tile) with
// (a) contiguous tile ranges per workgroup and (b) tiles dealt round-robin over the workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void rd(const float* __restrict__ base, float* out, int rows, int P, int NP, int mode, int piece) {
    // tile = (p, chunk of `piece` floats); rows x piece floats per tile; thread -> (row group, 16-byte part)
    const int nfc = NP / piece;
    const long total = (long)P * nfc;
    const int lpr = piece / 4;                 // lanes per row
    const int rpi = 512 / lpr;                 // rows per pass
    f32x4 acc = {0, 0, 0, 0};
    long t_begin = blockIdx.x * total / gridDim.x, t_end = (blockIdx.x + 1) * total / gridDim.x;
    for (long k = 0;; ++k) {
        long t = mode == 0 ? t_begin + k : blockIdx.x + k * gridDim.x;
        if (mode == 0 ? t >= t_end : t >= total) break;
        const int p = t / nfc, ch = t % nfc;
        for (int r = threadIdx.x / lpr; r < rows; r += rpi) {
            const f32x4 v = *(const f32x4*)(base + ((size_t)r * P + p) * NP + (size_t)ch * piece + 4 * (threadIdx.x % lpr));
            acc += v;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = 1.f;
}
int main() {
    const int rows = 384, P = 128, NP = 32256;
    size_t n = (size_t)rows * P * NP;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 4);
    hipMemset(d, 0, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int piece : {32, 64, 256})
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL(rd, dim3(256), dim3(512), 0, 0, d, o, rows, P, NP, mode, piece);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep == 2) printf("piece %4d B  mode %s: %.3f ms  %.2f TB/s\n", piece * 4, mode ? "round-robin" : "contiguous ", ms, n * 4 / ms / 1e9);
            }
        }
    return 0;
}
