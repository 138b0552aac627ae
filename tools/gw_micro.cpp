// Stand-alone timing harness of csrc/hcq_wgrad_grp.hip with parts of the kernel switched off at compile time
// (-DGW_DBG=mask, wrong results): which phase of a step costs what.  Built by tools/gw_micro_build.sh for several masks;
// usage: gw_micro_<mask> <family 0|1|2> [batch]
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../sound-event-localization-and-detection_amd/csrc/hcq_wgrad_grp.hip"

namespace seld {
thread_local int g_last_hip_error = 0;
const SeldEnv& env() { static SeldEnv e; return e; }
}

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((int)(h & 0xFFFF) - 32768) * (1.0f / 32768.0f);
    }
}

int main(int argc, char** argv) {
    const int fam = argc > 1 ? atoi(argv[1]) : 0;
    const int B = argc > 2 ? atoi(argv[2]) : 32;
    struct Shape { int cin, cout, H, W, kh, kw; int count; };
    const Shape shapes[3] = {{192, 384, 1, 512, 1, 3, 20}, {384, 192, 1, 512, 1, 1, 19}, {192, 192, 16, 512, 3, 3, 1}};
    const Shape s = shapes[fam];
    const int dil[10] = {1, 1, 2, 3, 5, 8, 13, 21, 34, 55};
    std::vector<seld_wgrad_job> jobs(s.count);
    for (int i = 0; i < s.count; ++i) {
        seld_wgrad_job& j = jobs[i];
        memset(&j, 0, sizeof(j));
        seld_conv_desc& d = j.desc;
        d.algebra = 8; d.ndim = s.H > 1 ? 2 : 1; d.N = B; d.Cin = s.cin; d.Cout = s.cout; d.groups = 1;
        d.in[0] = s.H; d.in[1] = s.W; d.k[0] = s.kh; d.k[1] = s.kw; d.stride[0] = d.stride[1] = 1;
        const int dl = (fam == 0) ? dil[(i / 2) % 10] : 1;
        d.dil[0] = 1; d.dil[1] = dl;
        d.pad[0] = (s.kh - 1) / 2; d.pad[1] = (s.kw - 1) / 2 * dl;
        const size_t nx = (size_t)B * s.cin * s.H * s.W, ny = (size_t)B * s.cout * s.H * s.W;
        float *x, *dy;
        hipMalloc(&x, nx * 4); hipMalloc(&dy, ny * 4);
        fill_kernel<<<2048, 256>>>(x, nx, 17u * i + 1);
        fill_kernel<<<2048, 256>>>(dy, ny, 31u * i + 7);
        j.x = x; j.dy = dy;
        for (int q = 0; q < 8; ++q) {
            float* w;
            const size_t nw = (size_t)(s.cout / 8) * (s.cin / 8) * s.kh * s.kw;
            hipMalloc(&w, nw * 4);
            hipMemset(w, 0, nw * 4);
            j.dw[q] = w;
        }
    }
    const size_t wsb = seld_hcq_wgrad_group_workspace(jobs.data(), (int)jobs.size());
    if (!wsb) { printf("shape not taken\n"); return 1; }
    void* ws;
    hipMalloc(&ws, wsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0, 0);
        const int rc = seld_hcq_wgrad_group(jobs.data(), (int)jobs.size(), ws, wsb, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (rc) { printf("launch failed %d\n", rc); return 1; }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    printf("GW_DBG=%d family %d batch %d: %.1f us\n", GW_DBG, fam, B, best * 1e3f);
    return 0;
}
