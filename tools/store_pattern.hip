// Store-only micro-benchmark for the first layer's output pattern: 32 images x 192 channels x 65536 positions fp32
// (1.61 GB).  A persistent 4-wave workgroup walks position tiles of TP positions and writes, per tile, TP*4 bytes into
// each of the 192 channel rows (row stride 256 KB).  Shows what the HBM write path gives for 256-byte runs per
// (workgroup, channel) against longer runs, without any compute.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern tools/store_pattern.hip && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int TP>   // positions per tile: 64, 128, 256
__global__ __launch_bounds__(256) void store_kernel(float* dst, long long ntiles, int dstS, int C) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)tid);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long pos0 = tile * TP;
        const long long img = pos0 / dstS, rem = pos0 - img * dstS;
        float* base = dst + (size_t)img * C * dstS + rem;
        // wave w covers channels [w*48, w*48+48): 3 groups of 16 channels (lane fr), each TP positions:
        // lane fk writes 16-byte pieces at fk*4, +16 positions, ...
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int ch = (wave * 3 + t) * 16 + fr;
#pragma unroll
            for (int s = 0; s < TP / 16; ++s)
                *reinterpret_cast<float4*>(base + (size_t)ch * dstS + s * 16 + fk * 4) = v;
        }
    }
}

template <int TP>
static void run(float* d, int grid) {
    const int dstS = 65536, C = 192, N = 32;
    const long long ntiles = (long long)N * dstS / TP;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<TP>, dim3(grid), dim3(256), 0, 0, d, ntiles, dstS, C);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(store_kernel<TP>, dim3(grid), dim3(256), 0, 0, d, ntiles, dstS, C);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)N * C * dstS * 4;
    printf("TP %3d grid %4d: %.1f us, %.2f TB/s\n", TP, grid, ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e12);
}

int main() {
    float* d;
    hipMalloc(&d, (size_t)32 * 192 * 65536 * 4);
    for (int grid : {512, 1024, 2048}) { run<64>(d, grid); run<128>(d, grid); run<256>(d, grid); }
    return 0;
}
