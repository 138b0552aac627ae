// Hypercomplex (real / quaternion / dual-quaternion) convolution for gfx950 (MI355X).
//
// Forward and data-gradient are ONE implicit-GEMM kernel on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32, exact fp32):
//
//     D[pos][ch] = sum_kk  X(pos, kk) * Wfull(ch, kk)
//
//   forward : ch = co, kk = ci*KK + kidx, X = im2col(x)            (quaternion_ops.py:147)
//   dgrad   : ch = ci, kk = co*KK + kidx, X = im2col of dy with the mirrored index map
//
// Wfull is the real block matrix of the Hamilton product (quaternion_ops.py:131-135,
// dual_quaternion_ops.py:122-140).  It is never materialised: the weight stager reads the
// 1/4/8 COMPONENT tensors and applies comp(p,q) / sign(p,q) on the way into LDS, and the MFMA
// loop skips the structurally-zero quadrant of the dual-quaternion matrix (25 % of the flops).
//
// Tiling (wave64): a workgroup of 4 waves owns BC = 16*CT channels x BP = 64*PT positions; the
// waves split the positions.  Positions sit on the MFMA row index so that each lane ends up
// with 4 consecutive positions of one channel -> 16-byte coalesced stores along T / W.
// LDS images are [k/4][row][4] so one ds_read_b128 feeds four MFMAs (conflict-free, see
// DESIGN.md), staging is register double-buffered with one barrier per 16-deep K step.
#include <stdio.h>
#include "common.h"

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
    int skip_mode;             // 0 none, 1 (fwd DQ): low-half channels x high-half K is zero, 2 (dgrad DQ): high x low
    int epilogue;
    WPtrs w;
    const float* src;
    const float* bias;
    float* dst;
    const float* addend;
    float* stats;
};

// element (ch, kk) of the expanded weight matrix, read from the component tensors
template <int KK_T>
__device__ __forceinline__ float wfull_elem(const ConvP& p, int ch, int kk) {
    const int KK = KK_T ? KK_T : p.KH * p.KW;
    int co, ci, kidx;
    if (p.mode == MODE_FWD) {
        co = ch;
        ci = kk / KK;
        kidx = kk - ci * KK;
    } else {
        ci = ch;
        co = kk / KK;
        kidx = kk - co * KK;
    }
    int pp = co / p.OA, o = co - pp * p.OA;
    int qq = ci / p.IA, c = ci - qq * p.IA;
    float sign;
    int comp = block_comp(p.algebra, pp, qq, &sign);
    if (comp < 0) return 0.0f;
    return sign * p.w.p[comp][((size_t)o * p.IA + c) * KK + kidx];
}

template <int CT, int PT, int KH_T, int KW_T>
__global__ __launch_bounds__(256) void hc_conv_kernel(const ConvP p) {
    constexpr int BC = CT * 16;
    constexpr int BP = PT * 64;
    constexpr int KK_T = KH_T * KW_T;
    constexpr int XV = (4 * BP) / 256;             // float4 items of the X stage per thread
    constexpr int WV = (4 * BC + 255) / 256;       // float4 items of the W stage per thread
    static_assert(256 % BP == 0 || BP % 256 == 0, "BP must divide or be a multiple of 256");

    __shared__ __attribute__((aligned(16))) float Xs[2][4][BP][4];
    __shared__ __attribute__((aligned(16))) float Ws[2][4][BC][4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c0 = blockIdx.y * BC;
    const long long p0 = (long long)blockIdx.x * BP;

    const int KH = KH_T ? KH_T : p.KH;
    const int KW = KW_T ? KW_T : p.KW;
    const int KK = KH * KW;

    // ---- per-thread position decode for the X stage (one position per thread) --------------
    const int xpos = tid % BP;
    const int xg0 = tid / BP;                      // first k-group this thread stages
    constexpr int XGSTEP = (256 / BP) > 0 ? (256 / BP) : 1;
    const long long pg = p0 + xpos;
    const bool pvalid = pg < p.Ptot;
    int base_h = 0, base_w = 0;
    size_t src_img = 0;
    if (pvalid) {
        long long img = pg / p.dstS;
        int rem = (int)(pg - img * p.dstS);
        int oh = rem / p.dstW;
        int ow = rem - oh * p.dstW;
        base_h = oh * p.SMh + p.OFFh;
        base_w = ow * p.SMw + p.OFFw;
        src_img = (size_t)img * p.Csrc * p.srcS;
    }

    // ---- K range of this workgroup (dual-quaternion zero quadrant) --------------------------
    const int half_c = p.Cdst >> 1;
    const int half_k = p.Ktot >> 1;
    const bool halves_aligned = (p.skip_mode != 0) && (half_k % 16 == 0) && (half_c % 16 == 0);
    int kbeg = 0, kend = p.Ktot;
    if (halves_aligned) {
        if (p.skip_mode == 1 && c0 + BC <= half_c) kend = half_k;      // all channels primal
        if (p.skip_mode == 2 && c0 >= half_c) kbeg = half_k;           // all channels dual
    }
    const int nchunks = (kend - kbeg + 15) >> 4;

    float xr[XV][4];
    float wr[WV][4];

    auto load_chunk = [&](int chunk) {
        const int kk0 = kbeg + chunk * 16;
        // X operand: im2col gather, coalesced along positions
#pragma unroll
        for (int j = 0; j < XV; ++j) {
            const int g = xg0 + j * XGSTEP;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kk = kk0 + g * 4 + s;
                float v = 0.0f;
                if (pvalid && kk < kend) {
                    int chan = kk / KK;
                    int kidx = kk - chan * KK;
                    int kh = kidx / KW;
                    int kw = kidx - kh * KW;
                    int ih = base_h + kh * p.KDh;
                    int iw = base_w + kw * p.KDw;
                    bool ok = true;
                    if (p.SDh > 1) { ok = ok && (ih % p.SDh == 0); ih /= p.SDh; }
                    if (p.SDw > 1) { ok = ok && (iw % p.SDw == 0); iw /= p.SDw; }
                    ok = ok && ih >= 0 && ih < p.srcH && iw >= 0 && iw < p.srcW;
                    if (ok) v = p.src[src_img + (size_t)chan * p.srcS + (size_t)ih * p.srcW + iw];
                }
                xr[j][s] = v;
            }
        }
        // W operand: signed gather from the component tensors
#pragma unroll
        for (int j = 0; j < WV; ++j) {
            const int item = tid + j * 256;
            const int ch = item % BC;
            const int g = item / BC;
            const int chg = c0 + ch;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kk = kk0 + g * 4 + s;
                float v = 0.0f;
                if (item < 4 * BC && chg < p.Cdst && kk < kend) v = wfull_elem<KK_T>(p, chg, kk);
                wr[j][s] = v;
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < XV; ++j) {
            const int g = xg0 + j * XGSTEP;
            *reinterpret_cast<float4*>(&Xs[buf][g][xpos][0]) = make_float4(xr[j][0], xr[j][1], xr[j][2], xr[j][3]);
        }
#pragma unroll
        for (int j = 0; j < WV; ++j) {
            const int item = tid + j * 256;
            if (item < 4 * BC) {
                const int ch = item % BC;
                const int g = item / BC;
                *reinterpret_cast<float4*>(&Ws[buf][g][ch][0]) = make_float4(wr[j][0], wr[j][1], wr[j][2], wr[j][3]);
            }
        }
    };

    floatx4 acc[PT][CT];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15;      // row / col inside a 16x16 tile
    const int fk = lane >> 4;      // k group of this lane

    if (nchunks > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunks) load_chunk(chunk + 1);

        const int kk0 = kbeg + chunk * 16;
        const bool chunk_hi = halves_aligned && (kk0 >= half_k);
        const bool chunk_lo = halves_aligned && (kk0 + 16 <= half_k);

        float4 a[PT], b[CT];
#pragma unroll
        for (int i = 0; i < PT; ++i)
            a[i] = *reinterpret_cast<const float4*>(&Xs[buf][fk][wave * (PT * 16) + i * 16 + fr][0]);
#pragma unroll
        for (int j = 0; j < CT; ++j) b[j] = *reinterpret_cast<const float4*>(&Ws[buf][fk][j * 16 + fr][0]);

#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int ctile0 = c0 + j * 16;
            bool skip = false;
            if (p.skip_mode == 1) skip = chunk_hi && (ctile0 + 16 <= half_c);
            if (p.skip_mode == 2) skip = chunk_lo && (ctile0 >= half_c);
            if (skip) continue;
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
            }
        }

        if (chunk + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds positions prow..prow+3 (regs) of channel ch -------------------
    const bool vec_ok = (p.dstS % 4 == 0);
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int ch = c0 + j * 16 + fr;
        const bool chok = ch < p.Cdst;
        const float bv = (chok && p.bias) ? p.bias[ch] : 0.0f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const long long pos = p0 + wave * (PT * 16) + i * 16 + fk * 4;
            floatx4 v = acc[i][j];
            if (chok && pos < p.Ptot) {
                if (vec_ok) {
                    long long img = pos / p.dstS;
                    int rem = (int)(pos - img * p.dstS);
                    size_t off = ((size_t)img * p.Cdst + ch) * p.dstS + rem;
                    float4 o = make_float4(v[0] + bv, v[1] + bv, v[2] + bv, v[3] + bv);
                    if (p.epilogue & SELD_EPI_ADD) {
                        float4 ad = *reinterpret_cast<const float4*>(p.addend + off);
                        o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                    }
                    if (p.epilogue & SELD_EPI_ACCUMULATE) {
                        float4 old = *reinterpret_cast<const float4*>(p.dst + off);
                        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                    }
                    *reinterpret_cast<float4*>(p.dst + off) = o;
                    s1 += o.x + o.y + o.z + o.w;
                    s2 += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        long long pr = pos + r;
                        if (pr < p.Ptot) {
                            long long img = pr / p.dstS;
                            int rem = (int)(pr - img * p.dstS);
                            size_t off = ((size_t)img * p.Cdst + ch) * p.dstS + rem;
                            float o = v[r] + bv;
                            if (p.epilogue & SELD_EPI_ADD) o += p.addend[off];
                            if (p.epilogue & SELD_EPI_ACCUMULATE) o += p.dst[off];
                            p.dst[off] = o;
                            s1 += o;
                            s2 += o * o;
                        }
                    }
                }
            }
        }
        if (p.epilogue & SELD_EPI_STATS) {
            // lanes fr, fr+16, fr+32, fr+48 hold the same channel
            s1 += __shfl_xor(s1, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (fk == 0 && chok) {
                atomicAdd(p.stats + ch, s1);
                atomicAdd(p.stats + p.Cdst + ch, s2);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient:  dWfull[co][ci*KK + kidx] = sum_{img,pos} dy[img][co][pos] * x[img][ci][src(pos,kidx)]
// One implicit GEMM with the reduction over positions, split over workgroups; partial slabs go
// to the workspace and `hc_wgrad_fold_kernel` sums the slabs and folds the signed blocks that
// share a component (the transpose of the assembly at quaternion_ops.py:131-135).
// ------------------------------------------------------------------------------------------
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
    float* partial;    // [nsplit][Cout][Ktot]
};

template <int RT /*row (co) tiles per wave*/, int CTL /*col tiles per wave*/, int KH_T, int KW_T>
__global__ __launch_bounds__(256) void hc_wgrad_kernel(const WgradP p) {
    // workgroup tile: rows BM = 2*RT*16 (2 waves along rows), cols BN = 2*CTL*16 (2 waves along cols)
    constexpr int BM = 2 * RT * 16;
    constexpr int BN = 2 * CTL * 16;
    constexpr int AV = (4 * BM + 255) / 256;
    constexpr int BV = (4 * BN + 255) / 256;
    __shared__ __attribute__((aligned(16))) float As[2][4][BM][4];   // dy   [k-group][co][4 positions]
    __shared__ __attribute__((aligned(16))) float Bs[2][4][BN][4];   // xcol [k-group][col][4 positions]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr_ = wave >> 1, wc_ = wave & 1;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int split = blockIdx.z;
    const int KH = KH_T ? KH_T : p.KH;
    const int KW = KW_T ? KW_T : p.KW;
    const int KK = KH * KW;

    // structural zero block of the dual-quaternion matrix: rows primal (co < Cout/2), cols dual (ci >= Cin/2)
    if (p.algebra == 8) {
        if (m0 + BM <= (p.Cout >> 1) && n0 >= (p.Ktot >> 1)) return;
    }

    const long long pbeg = (long long)split * p.split_len;
    long long pend = pbeg + p.split_len;
    if (pend > p.Ptot) pend = p.Ptot;
    const int nchunks = pbeg < pend ? (int)((pend - pbeg + 15) >> 4) : 0;

    float ar[AV][4], br[BV][4];

    auto load_chunk = [&](int chunk) {
        const long long pos0 = pbeg + (long long)chunk * 16;
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int item = tid + j * 256;
            const int row = item % BM, g = item / BM;
            const int co = m0 + row;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const long long pos = pos0 + g * 4 + s;
                float v = 0.f;
                if (item < 4 * BM && co < p.Cout && pos < pend) {
                    long long img = pos / p.outS;
                    int rem = (int)(pos - img * p.outS);
                    v = p.dy[((size_t)img * p.Cout + co) * p.outS + rem];
                }
                ar[j][s] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int item = tid + j * 256;
            const int col = item % BN, g = item / BN;
            const int kk = n0 + col;
            const int ci = kk / KK;
            const int kidx = kk - ci * KK;
            const int kh = kidx / KW, kw = kidx - kh * KW;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const long long pos = pos0 + g * 4 + s;
                float v = 0.f;
                if (item < 4 * BN && kk < p.Ktot && pos < pend) {
                    long long img = pos / p.outS;
                    int rem = (int)(pos - img * p.outS);
                    int oh = rem / p.outW, ow = rem - oh * p.outW;
                    int ih = oh * p.sh - p.ph + kh * p.dh;
                    int iw = ow * p.sw - p.pw + kw * p.dw;
                    if (ih >= 0 && ih < p.inH && iw >= 0 && iw < p.inW)
                        v = p.x[((size_t)img * p.Cin + ci) * p.inS + (size_t)ih * p.inW + iw];
                }
                br[j][s] = v;
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AV; ++j) {
            const int item = tid + j * 256;
            if (item < 4 * BM)
                *reinterpret_cast<float4*>(&As[buf][item / BM][item % BM][0]) = make_float4(ar[j][0], ar[j][1], ar[j][2], ar[j][3]);
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int item = tid + j * 256;
            if (item < 4 * BN)
                *reinterpret_cast<float4*>(&Bs[buf][item / BN][item % BN][0]) = make_float4(br[j][0], br[j][1], br[j][2], br[j][3]);
        }
    };

    floatx4 acc[RT][CTL];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CTL; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fk = lane >> 4;
    if (nchunks > 0) { load_chunk(0); store_chunk(0); }
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunks) load_chunk(chunk + 1);
        float4 a[RT], b[CTL];
#pragma unroll
        for (int i = 0; i < RT; ++i) a[i] = *reinterpret_cast<const float4*>(&As[buf][fk][wr_ * (RT * 16) + i * 16 + fr][0]);
#pragma unroll
        for (int j = 0; j < CTL; ++j) b[j] = *reinterpret_cast<const float4*>(&Bs[buf][fk][wc_ * (CTL * 16) + j * 16 + fr][0]);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CTL; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
            }
        if (chunk + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // D layout: col = lane&15 (B index = column kk), row = (lane>>4)*4 + r (A index = co)
    float* out = p.partial + (size_t)split * p.Cout * p.Ktot;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CTL; ++j) {
            const int kk = n0 + wc_ * (CTL * 16) + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + wr_ * (RT * 16) + i * 16 + fk * 4 + r;
                if (co < p.Cout && kk < p.Ktot) out[(size_t)co * p.Ktot + kk] = acc[i][j][r];
            }
        }
}

// dw[comp][o][c][kidx] = sum_split sum_{(p,q) -> comp} sign * partial[split][p*OA+o][(q*IA+c)*KK+kidx]
__global__ void hc_wgrad_fold_kernel(int algebra, int OA, int IA, int KK, int Cout, int Ktot, int nsplit,
                                     const float* __restrict__ partial, WPtrsMut dw) {
    const int per = OA * IA * KK;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * algebra) return;
    const int comp = idx / per;
    const int rem = idx - comp * per;
    const int o = rem / (IA * KK);
    const int ck = rem - o * (IA * KK);
    float total = 0.f;
    for (int pp = 0; pp < algebra; ++pp)
        for (int qq = 0; qq < algebra; ++qq) {
            float sign;
            int cc = block_comp(algebra, pp, qq, &sign);
            if (cc != comp) continue;
            const size_t off = (size_t)(pp * OA + o) * Ktot + (size_t)qq * IA * KK + ck;
            float s = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) s += partial[(size_t)sp * Cout * Ktot + off];
            total += sign * s;
        }
    dw.p[comp][rem] = total;
}

// per-channel sum over (N, S): dbias
__global__ void channel_sum_kernel(const float* __restrict__ x, int N, int C, int S, float* __restrict__ out) {
    const int c = blockIdx.x;
    float s = 0.f;
    const long long total = (long long)N * S;
    for (long long i = threadIdx.x; i < total; i += blockDim.x) {
        long long n = i / S;
        int r = (int)(i - n * S);
        s += x[((size_t)n * C + c) * S + r];
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int validate(const seld_conv_desc* d) {
    if (!d) return SELD_EINVAL;
    if (d->algebra != 1 && d->algebra != 4 && d->algebra != 8) return SELD_EINVAL;
    if (d->ndim != 1 && d->ndim != 2) return SELD_EINVAL;
    if (d->groups != 1) return SELD_EUNSUPPORTED;
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return SELD_EINVAL;
    if (d->Cin % d->algebra || d->Cout % d->algebra) return SELD_EINVAL;
    for (int i = 0; i < 2; ++i)
        if (d->in[i] <= 0 || d->k[i] <= 0 || d->stride[i] <= 0 || d->dil[i] <= 0 || d->pad[i] < 0) return SELD_EINVAL;
    if (d->k[0] * d->k[1] > 255) return SELD_EUNSUPPORTED;
    return SELD_OK;
}

static void out_shape(const seld_conv_desc* d, int out[2]) {
    for (int i = 0; i < 2; ++i)
        out[i] = (d->in[i] + 2 * d->pad[i] - d->dil[i] * (d->k[i] - 1) - 1) / d->stride[i] + 1;
}

template <int CT, int PT>
static void launch_conv(const ConvP& p, hipStream_t st) {
    constexpr int BC = CT * 16, BP = PT * 64;
    dim3 grid((unsigned)((p.Ptot + BP - 1) / BP), (unsigned)((p.Cdst + BC - 1) / BC), 1);
    if (p.KH == 1 && p.KW == 1) hipLaunchKernelGGL((hc_conv_kernel<CT, PT, 1, 1>), grid, dim3(256), 0, st, p);
    else if (p.KH == 1 && p.KW == 3) hipLaunchKernelGGL((hc_conv_kernel<CT, PT, 1, 3>), grid, dim3(256), 0, st, p);
    else if (p.KH == 3 && p.KW == 3) hipLaunchKernelGGL((hc_conv_kernel<CT, PT, 3, 3>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((hc_conv_kernel<CT, PT, 0, 0>), grid, dim3(256), 0, st, p);
}

// tile selection shared by the launcher and by seld_hc_conv_kernel_label
static void pick_tiles(int C, int* ct, int* pt) {
    *pt = 4;
    if (C <= 16) *ct = 1;
    else if (C <= 32) *ct = 2;
    else if (C % 96 == 0 && C % 64 != 0) *ct = 6;
    else *ct = 4;
}

static int run_conv(ConvP& p, hipStream_t st) {
    int ct, pt;
    pick_tiles(p.Cdst, &ct, &pt);
    if (ct == 1) launch_conv<1, 4>(p, st);
    else if (ct == 2) launch_conv<2, 4>(p, st);
    else if (ct == 6) launch_conv<6, 4>(p, st);
    else launch_conv<4, 4>(p, st);
    return check_launch();
}

}  // namespace seld

using namespace seld;

extern "C" int seld_hc_conv_out_shape(const seld_conv_desc* d, int32_t out[2]) {
    int rc = validate(d);
    if (rc) return rc;
    out_shape(d, out);
    return (out[0] > 0 && out[1] > 0) ? SELD_OK : SELD_EINVAL;
}

extern "C" int seld_hc_conv_fwd_ex(const seld_conv_desc* d, const float* x, const float* const w[8],
                                   const float* bias, float* y, int32_t epilogue, const float* addend,
                                   float* stats, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    int o[2];
    out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0 || !x || !w || !y) return SELD_EINVAL;
    if ((epilogue & SELD_EPI_ADD) && !addend) return SELD_EINVAL;
    if ((epilogue & SELD_EPI_STATS) && !stats) return SELD_EINVAL;
    ConvP p{};
    p.algebra = d->algebra; p.mode = MODE_FWD;
    p.Csrc = d->Cin; p.Cdst = d->Cout;
    p.srcH = d->in[0]; p.srcW = d->in[1]; p.dstH = o[0]; p.dstW = o[1];
    p.KH = d->k[0]; p.KW = d->k[1];
    p.SMh = d->stride[0]; p.OFFh = -d->pad[0]; p.KDh = d->dil[0]; p.SDh = 1;
    p.SMw = d->stride[1]; p.OFFw = -d->pad[1]; p.KDw = d->dil[1]; p.SDw = 1;
    p.Ktot = d->Cin * p.KH * p.KW;
    p.OA = d->Cout / d->algebra; p.IA = d->Cin / d->algebra;
    p.srcS = p.srcH * p.srcW; p.dstS = p.dstH * p.dstW;
    p.Ptot = (long long)d->N * p.dstS;
    p.skip_mode = (d->algebra == 8) ? 1 : 0;
    p.epilogue = epilogue;
    for (int i = 0; i < 8; ++i) p.w.p[i] = (i < d->algebra) ? w[i] : nullptr;
    p.src = x; p.bias = bias; p.dst = y; p.addend = addend; p.stats = stats;
    return run_conv(p, (hipStream_t)stream);
}

extern "C" int seld_hc_conv_fwd(const seld_conv_desc* d, const float* x, const float* const w[8],
                                const float* bias, float* y, void* stream) {
    return seld_hc_conv_fwd_ex(d, x, w, bias, y, SELD_EPI_NONE, nullptr, nullptr, stream);
}

extern "C" int seld_hc_conv_bwd_data(const seld_conv_desc* d, const float* dy, const float* const w[8],
                                     float* dx, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    int o[2];
    out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0 || !dy || !w || !dx) return SELD_EINVAL;
    ConvP p{};
    p.algebra = d->algebra; p.mode = MODE_DGRAD;
    p.Csrc = d->Cout; p.Cdst = d->Cin;
    p.srcH = o[0]; p.srcW = o[1]; p.dstH = d->in[0]; p.dstW = d->in[1];
    p.KH = d->k[0]; p.KW = d->k[1];
    // oh = (ih + pad - kh*dil) / stride
    p.SMh = 1; p.OFFh = d->pad[0]; p.KDh = -d->dil[0]; p.SDh = d->stride[0];
    p.SMw = 1; p.OFFw = d->pad[1]; p.KDw = -d->dil[1]; p.SDw = d->stride[1];
    p.Ktot = d->Cout * p.KH * p.KW;
    p.OA = d->Cout / d->algebra; p.IA = d->Cin / d->algebra;
    p.srcS = p.srcH * p.srcW; p.dstS = p.dstH * p.dstW;
    p.Ptot = (long long)d->N * p.dstS;
    p.skip_mode = (d->algebra == 8) ? 2 : 0;
    p.epilogue = 0;
    for (int i = 0; i < 8; ++i) p.w.p[i] = (i < d->algebra) ? w[i] : nullptr;
    p.src = dy; p.bias = nullptr; p.dst = dx;
    return run_conv(p, (hipStream_t)stream);
}

static int wgrad_splits(const seld_conv_desc* d, int o[2], long long* split_len) {
    long long Ptot = (long long)d->N * o[0] * o[1];
    long long Ktot = (long long)d->Cin * d->k[0] * d->k[1];
    long long tiles = ((d->Cout + 63) / 64) * ((Ktot + 63) / 64);
    long long want = (1024 + tiles - 1) / tiles;          // ~4 workgroups per CU
    long long maxs = (Ptot + 255) / 256;                  // at least 256 positions per split
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    long long len = (Ptot + want - 1) / want;
    len = (len + 15) / 16 * 16;
    int ns = (int)((Ptot + len - 1) / len);
    *split_len = len;
    return ns < 1 ? 1 : ns;
}

extern "C" size_t seld_hc_conv_bwd_weight_workspace(const seld_conv_desc* d) {
    if (validate(d)) return 0;
    int o[2];
    out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0) return 0;
    long long sl;
    int ns = wgrad_splits(d, o, &sl);
    return (size_t)ns * d->Cout * d->Cin * d->k[0] * d->k[1] * sizeof(float);
}

extern "C" int seld_hc_conv_bwd_weight(const seld_conv_desc* d, const float* x, const float* dy,
                                       float* const dw[8], float* dbias, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    int rc = validate(d);
    if (rc) return rc;
    int o[2];
    out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0 || !x || !dy || !dw) return SELD_EINVAL;
    if (workspace_bytes < seld_hc_conv_bwd_weight_workspace(d) || !workspace) return SELD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    WgradP p{};
    p.algebra = d->algebra; p.N = d->N; p.Cin = d->Cin; p.Cout = d->Cout;
    p.inH = d->in[0]; p.inW = d->in[1]; p.outH = o[0]; p.outW = o[1];
    p.KH = d->k[0]; p.KW = d->k[1];
    p.sh = d->stride[0]; p.sw = d->stride[1]; p.ph = d->pad[0]; p.pw = d->pad[1]; p.dh = d->dil[0]; p.dw = d->dil[1];
    p.Ktot = d->Cin * p.KH * p.KW;
    p.OA = d->Cout / d->algebra; p.IA = d->Cin / d->algebra;
    p.inS = p.inH * p.inW; p.outS = p.outH * p.outW;
    p.Ptot = (long long)d->N * p.outS;
    p.nsplit = wgrad_splits(d, o, &p.split_len);
    p.x = x; p.dy = dy; p.partial = (float*)workspace;
    // the zero quadrant is skipped by the GEMM kernel, so the fold must not read garbage there:
    // it never does (block_comp returns -1 for it).
    dim3 grid((p.Ktot + 63) / 64, (p.Cout + 63) / 64, p.nsplit);
    if (p.KH == 1 && p.KW == 1) hipLaunchKernelGGL((hc_wgrad_kernel<2, 2, 1, 1>), grid, dim3(256), 0, st, p);
    else if (p.KH == 1 && p.KW == 3) hipLaunchKernelGGL((hc_wgrad_kernel<2, 2, 1, 3>), grid, dim3(256), 0, st, p);
    else if (p.KH == 3 && p.KW == 3) hipLaunchKernelGGL((hc_wgrad_kernel<2, 2, 3, 3>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((hc_wgrad_kernel<2, 2, 0, 0>), grid, dim3(256), 0, st, p);
    rc = check_launch();
    if (rc) return rc;
    WPtrsMut out{};
    for (int i = 0; i < 8; ++i) out.p[i] = (i < d->algebra) ? dw[i] : nullptr;
    const int KK = p.KH * p.KW;
    const int total = p.OA * p.IA * KK * d->algebra;
    hipLaunchKernelGGL(hc_wgrad_fold_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d->algebra, p.OA, p.IA, KK,
                       d->Cout, p.Ktot, p.nsplit, (const float*)workspace, out);
    rc = check_launch();
    if (rc) return rc;
    if (dbias) {
        hipLaunchKernelGGL(channel_sum_kernel, dim3(d->Cout), dim3(256), 0, st, dy, d->N, d->Cout, p.outS, dbias);
        rc = check_launch();
    }
    return rc;
}

// Label of the kernel symbol a call would launch ("hc_conv_kernel<CT,PT,KH,KW>" as rocprofv3 prints the
// template arguments); which = 0 forward, 1 data gradient, 2 weight gradient.
extern "C" int seld_hc_conv_kernel_label(const seld_conv_desc* d, int32_t which, char* buf, int32_t buflen) {
    int rc = validate(d);
    if (rc || !buf || buflen < 48) return SELD_EINVAL;
    int kh = d->k[0], kw = d->k[1];
    if (!((kh == 1 && kw == 1) || (kh == 1 && kw == 3) || (kh == 3 && kw == 3))) kh = kw = 0;
    if (which == 2) {
        snprintf(buf, buflen, "hc_wgrad_kernel<2, 2, %d, %d>", kh, kw);
    } else {
        int ct, pt;
        pick_tiles(which == 0 ? d->Cout : d->Cin, &ct, &pt);
        snprintf(buf, buflen, "hc_conv_kernel<%d, %d, %d, %d>", ct, pt, kh, kw);
    }
    return SELD_OK;
}
