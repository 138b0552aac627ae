// Shared helpers for the gfx950 kernels of libseld_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/seld_hip.h"

namespace seld {

extern thread_local int g_last_hip_error;

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return SELD_ELAUNCH;
    }
    return SELD_OK;
}

typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---- Hamilton product block structure --------------------------------------
// Real matrix of the left Hamilton product, rows = output component p, cols = input component q:
//   component index  comp(p, q) = p ^ q           (r=0, i=1, j=2, k=3)
//   sign negative at (0,1) (0,2) (0,3) (1,2) (2,3) (3,1)      -> bit p*4+q of 0x284E
// Dual quaternion [[Q, 0], [Q2, Q]]: halves hp = p>>2, hq = q>>2; block is zero for hp=0,hq=1,
// uses the second weight set (+4) for hp=1,hq=0.
__host__ __device__ __forceinline__ int quat_comp(int p, int q) { return (p ^ q) & 3; }
__host__ __device__ __forceinline__ float quat_sign(int p, int q) {
    return ((0x284Eu >> (((p & 3) << 2) | (q & 3))) & 1u) ? -1.0f : 1.0f;
}
// returns component index in [0, A) or -1 for a structural zero; *sign receives +-1
__host__ __device__ __forceinline__ int block_comp(int algebra, int p, int q, float* sign) {
    if (algebra == 1) { *sign = 1.0f; return 0; }
    *sign = quat_sign(p, q);
    int c = quat_comp(p, q);
    if (algebra == 4) return c;
    int hp = p >> 2, hq = q >> 2;
    if (hp == 0 && hq == 1) return -1;
    return (hp == 1 && hq == 0) ? c + 4 : c;
}

struct WPtrs {
    const float* p[8];
};
struct WPtrsMut {
    float* p[8];
};

// Philox-4x32-10 (counter, key) -> four 32-bit draws; u01: the top 24 bits as a float in [0, 1)
__device__ __forceinline__ uint4 philox4x32_10(uint64_t counter, uint64_t key) {
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// a Dropout to replay on a gradient while it is loaded (p == 0: none)
struct DropP {
    float p, scale;
    uint64_t seed, offset;
    const uint64_t* state;      // nullable: the device-resident step state, state[0] is added to offset
};

// four dropout factors (0 or scale) of the 128-bit group `group`: the mask seld_dropout_fwd draws for the same stream position
__device__ __forceinline__ float4 dropout_mask4(uint64_t group, uint64_t seed, float p, float scale) {
    const uint4 r = philox4x32_10(group, seed);
    return make_float4(u01(r.x) >= p ? scale : 0.f, u01(r.y) >= p ? scale : 0.f, u01(r.z) >= p ? scale : 0.f, u01(r.w) >= p ? scale : 0.f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace seld
