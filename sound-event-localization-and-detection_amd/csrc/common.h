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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace seld
