// Store-only micro-benchmark, part 2: does it matter WHICH wave completes a 128-byte line?  Output (32, 192, 128, 512) fp32.
// A workgroup = 64 columns x 8 rows of one image (the first-stage kernels' tile).
//   A: wave w writes columns [16w, 16w+16) of every channel (64-byte pieces; a line is completed by two waves)
//   B: wave w writes columns [32(w&1), +32) of half the channels (two back-to-back 64-byte pieces = one line per channel)
//   C: like A, but each wave's pieces of a row go out channel-tile by channel-tile with the other waves' interleaved in time
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sp2 tools/store_pattern2.hip && /tmp/sp2
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* dst, int W, int H, int C) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    unsigned b = blockIdx.x;
    b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
    const int tiles = W / 64, hb = H / 8;
    const int w0 = (b % tiles) * 64; b /= tiles;
    const int h0 = (b % hb) * 8; const int n = b / hb;
    const size_t S = (size_t)H * W;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)tid);
    float* img = dst + (size_t)n * C * S;
    for (int t = 0; t < 3; ++t)
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {
                for (int q = 0; q < 4; ++q) {
                    const int ch = (t * 4 + q) * 16 + fr;
                    *reinterpret_cast<float4*>(img + (size_t)ch * S + (size_t)(h0 + r) * W + w0 + wave * 16 + fk * 4) = v;
                }
            } else {
                for (int q = 0; q < 2; ++q) {
                    const int ch = (t * 4 + (wave >> 1) * 2 + q) * 16 + fr;
                    float* p = img + (size_t)ch * S + (size_t)(h0 + r) * W + w0 + (wave & 1) * 32 + fk * 4;
                    *reinterpret_cast<float4*>(p) = v;
                    *reinterpret_cast<float4*>(p + 16) = v;
                }
            }
            if (MODE == 2) __syncthreads();
        }
}

template <int MODE>
static void run(float* d, const char* name) {
    const int N = 32, C = 192, H = 128, W = 512;
    const int grid = N * (H / 8) * (W / 64);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, W, H, C);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, W, H, C);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)N * C * H * W * 4;
    printf("%s: %.1f us, %.2f TB/s\n", name, ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e12);
}

int main() {
    float* d;
    hipMalloc(&d, (size_t)32 * 192 * 128 * 512 * 4);
    run<0>(d, "A 64-byte pieces per wave ");
    run<1>(d, "B 128-byte lines per wave ");
    run<2>(d, "B + barrier per row       ");
    return 0;
}
