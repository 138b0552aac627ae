// Shared declarations of the hypercomplex convolution kernels (hc_conv_fwd.hip, hc_wgrad.hip).
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
#include "env.h"

namespace seld {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

struct ConvP {
    int algebra, mode;
    int Csrc, Cdst;            // channels of the streamed operand / of the result
    int srcH, srcW, dstH, dstW;
    int KH, KW;
    int SMh, OFFh, KDh, SDh;   // src index = d*SM + OFF + k*KD, then (if SD > 1) must divide by SD
    int SMw, OFFw, KDw, SDw;
    int Ktot;                  // Csrc * KH * KW
    int OA, IA;                // Cout/A, Cin/A of the convolution
    int srcS, dstS;
    long long Ptot;            // N * dstS
    long long src_elems;       // N * Csrc * srcS
    int pairing;               // allow primal/dual paired channel tiles (work balance)
    int wt;                    // dgrad: component tensors are transposed to [c][o][k]
    int skip_mode;             // 0 none, 1 (fwd DQ): low-half channels x high-half K is zero, 2 (dgrad DQ): high x low
    int epilogue;
    WPtrs w;
    const float* wmin;         // lowest component pointer (hc_conv_vec_kernel addresses the others as 32-bit offsets from it)
    unsigned wspan;            // bytes from wmin to the end of the highest component tensor
    const float* src;
    const float* bias;
    float* dst;
    const float* addend;
    float* stats;
    // A second convolution of the SAME geometry in the same launch (hc_conv_vec_kernel only; nslots = 2).
    //   forward : same src, blockIdx.z selects {w, bias, dst, addend, stats, epilogue} or the *2 set
    //             (filter | gate and skip | residual of a residual block read the same tensor, model.py:121-132)
    //   dgrad   : dst = dgrad(src, w) + dgrad(src2, w2): the K loop runs over both sources in turn
    int nslots;
    int epilogue2;
    WPtrs w2;
    const float* src2;
    const float* bias2;
    float* dst2;
    const float* addend2;
    float* stats2;
};

struct WgradP {
    int algebra;
    int N, Cin, Cout;
    int inH, inW, outH, outW;
    int KH, KW;
    int sh, sw, ph, pw, dh, dw;
    int Ktot;          // Cin*KK  (columns)
    int OA, IA;
    int inS, outS;
    long long Ptot;    // N*outS  (reduction length)
    int nsplit;
    long long split_len;   // positions per split (multiple of 16)
    const float* x;
    const float* dy;
    WPtrsMut gw;       // component gradients (accumulated into)
    // a second weight gradient with the same x in the same launch (hc_wgrad_row_kernel only): blockIdx.y selects
    // {dy, gw} or {dy2, gw2}
    int nslots;
    const float* dy2;
    WPtrsMut gw2;
    // Fused BatchNorm+ReLU+MaxPool(ph, 1) backward (hc_wgrad_row_kernel<..., 1>): `dy` is the CONV OUTPUT y and the
    // gradient is formed while staging:  dy = y*c1[co] + dz*a[co] + c0[co],  dz = dpooled at the arg-max row of a
    // window whose pooled value is > 0, else 0.  coef = [c1 | a | c0] (3*Cout).
    const float* pooled;
    const float* dpooled;
    const unsigned char* pidx;
    const float* coef;
    int poolh;         // pooling window height
    DropP drop;        // FUSED: dpooled is the gradient behind the stage's Dropout; replay its mask (p == 0: none)
    int mz, nact, nt;  // tile enumeration without the zero quadrant: the first mz row tiles have nact column tiles, the rest nt
    int dbg;           // SELD_WGRAD_DBG: timing experiments (wrong results): 1 = no loads in the loop, 2 = no LDS stores
};

// blockIdx.z -> (row tile, column tile), skipping the tiles that lie wholly in the dual-quaternion zero quadrant.
// The position split is blockIdx.x, the FAST dispatch index: workgroup ids go round-robin to the 8 XCDs, so with a
// split count that is a multiple of 8 all tiles of one split -- which read the same dy rows / x columns -- run on
// the same XCD and share its L2 (PMC: the 3x3 layer fetched 3.4x its operands with the tile index fastest).
__device__ __forceinline__ void wgrad_tile(const WgradP& p, int* mt, int* nt) {
    const int t = blockIdx.z, head = p.mz * p.nact;
    if (t < head) { *mt = t / p.nact; *nt = t - *mt * p.nact; }
    else { const int u = t - head; const int m = u / p.nt; *mt = p.mz + m; *nt = u - m * p.nt; }
}

// Block (p, q) of the Hamilton matrix with one index per lane and the other wave-uniform:
// returns the component, *zero for the structural zero quadrant, *neg for a negative sign.
__device__ __forceinline__ int hc_comp(int algebra, int pp, int qq, bool* zero, bool* neg) {
    *zero = false;
    *neg = false;
    if (algebra == 1) return 0;
    *neg = (0x284Eu >> (((pp & 3) << 2) | (qq & 3))) & 1u;
    int c = (pp ^ qq) & 3;
    if (algebra == 8) {
        const int hp = pp >> 2, hq = qq >> 2;
        *zero = (hp == 0 && hq == 1);
        if (hp == 1 && hq == 0) c += 4;
    }
    return c;
}

inline int hc_validate(const seld_conv_desc* d) {
    if (!d) return SELD_EINVAL;
    if (d->algebra != 1 && d->algebra != 4 && d->algebra != 8) return SELD_EINVAL;
    if (d->ndim != 1 && d->ndim != 2) return SELD_EINVAL;
    if (d->groups != 1) return SELD_EUNSUPPORTED;
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return SELD_EINVAL;
    if (d->Cin % d->algebra || d->Cout % d->algebra) return SELD_EINVAL;
    for (int i = 0; i < 2; ++i)
        if (d->in[i] <= 0 || d->k[i] <= 0 || d->stride[i] <= 0 || d->dil[i] <= 0 || d->pad[i] < 0) return SELD_EINVAL;
    if (d->k[0] * d->k[1] > 255) return SELD_EUNSUPPORTED;
    // one image of either operand is addressed with 32-bit byte offsets
    if ((long long)d->Cin * d->in[0] * d->in[1] >= (1LL << 28)) return SELD_EUNSUPPORTED;
    return SELD_OK;
}

inline void hc_out_shape(const seld_conv_desc* d, int out[2]) {
    for (int i = 0; i < 2; ++i)
        out[i] = (d->in[i] + 2 * d->pad[i] - d->dil[i] * (d->k[i] - 1) - 1) / d->stride[i] + 1;
}

}  // namespace seld
