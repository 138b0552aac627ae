// First CNN stage of the SELD networks WITHOUT its 1.6 GB convolution output (gfx950).
//
//     conv3x3 (8 real input channels -> Cout) -> BatchNorm2d (batch statistics) -> ReLU -> MaxPool2d(8, 1) -> Dropout
//
// (model.py:273-283 on the network input; quaternion_ops.py:125-147 / dual_quaternion_ops.py:111-153 for the convolution.)
// The convolution output y = W xcol is LINEAR in an 8-channel input, so nothing downstream needs y in memory:
//
//   * BatchNorm's batch statistics follow from the input's second moments: with xcol(pos) the 72 values under the 3x3
//     window (8 channels x 9 taps, zero outside the image), s = sum_pos xcol and G = sum_pos xcol xcol^T (72 x 72),
//         sum_pos y_c = w_c . s          sum_pos y_c^2 = w_c^T G w_c
//     for every output channel c, w_c being row c of the real 192 x 72 matrix the reference assembles from the weight
//     components.  fs_gram_kernel accumulates [G; s] on the matrix cores in one pass over the input (67 MB at batch 32,
//     15 tiles of 16 x 16 per 4 positions), fs_gram_fold_kernel adds the workgroups' partials in double precision in a fixed
//     order, fs_bn_from_gram_kernel evaluates the two forms above in double and finishes mean / invstd / running
//     statistics exactly as seld_bn_finalize_ex does.
//   * The pooling convolution (csrc/hcq_conv.hip, hcq_first_pool_kernel<.., FIN>) then runs with y and its statistics
//     switched off and finishes the stage in its epilogue: relu(a v + b) and the Dropout on the window value -- the stage's
//     output and the window row are all it writes (pooled size; the raw value only for channels with gamma == 0).
//   * Backward, the gradient w.r.t. y is dy = c1 y + a dz + c0 per channel (BatchNorm backward; dz = the pooled gradient at
//     the window's arg-max row where ReLU is open, 0 elsewhere), so the weight gradient of the real matrix is
//         dW_c = c1_c (W G)_c  +  c0_c s  +  a_c sum_pos dz_c(pos) xcol(pos)
//     -- the first two terms come from G and s again, the third (fs_wgrad_kernel) needs the pooled-size tensors and x only.
//     No y is read, and no atomics are used: workgroup partials + a fold in fixed order (reproducible).
#include <string.h>
#include <type_traits>
#include "hc_common.h"

namespace seld {

constexpr int FS_K = 72;                  // 8 input channels x 9 taps
constexpr int FS_KP = 80;                 // padded to 5 tiles of 16; row / column 72 is the constant 1 (gives s)
constexpr int FS_WEXT = 72, FS_DPAD = 4;  // staged columns: w0 - 4 .. w0 + 67
constexpr int FS_XR = 10;                 // staged rows: h0 - 1 .. h0 + 8
constexpr int FS_NTILE = 15;              // upper triangle of the 5 x 5 tile grid

typedef unsigned int uintx4f __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int fs_tri(int t, int u) { return t * 5 - t * (t - 1) / 2 + (u - t); }      // t <= u < 5

// x tile of one block (8 window rows x 64 columns of image n): [channel][10 rows][72 columns], zero outside the image.
// A thread's FS_XI 16-byte items are requested together (fs_load_x) and written to LDS later (fs_store_x): one load ->
// wait -> store per loop iteration exposed six memory latencies per block.
constexpr int FS_XI = (8 * FS_XR * (FS_WEXT / 4) + 255) / 256;
__device__ __forceinline__ void fs_load_x(const float* x, uintx4f (&v)[FS_XI], int n, int h0, int w0, int H, int W, int tid) {
    const unsigned S = (unsigned)(H * W);
    const unsigned OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)n * 8 * S), 0, 8u * S * 4u, 0x00020000);
    constexpr int qw = FS_WEXT / 4;
#pragma unroll
    for (int i = 0; i < FS_XI; ++i) {
        const int f = tid + 256 * i;
        const int row = f / qw, quad = f - row * qw;
        const int ch = row / FS_XR, xr = row - ch * FS_XR;
        const int hh = h0 - 1 + xr, ww = w0 - FS_DPAD + 4 * quad;
        const bool ok = f < 8 * FS_XR * qw && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
        const unsigned e = ((unsigned)ch * (unsigned)H + (unsigned)hh) * (unsigned)W + (unsigned)ww;
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? e * 4u : OOB, 0, 0);
    }
}
__device__ __forceinline__ void fs_store_x(float* lds, const uintx4f (&v)[FS_XI], int tid) {
#pragma unroll
    for (int i = 0; i < FS_XI; ++i) {
        const int f = tid + 256 * i;
        if (f < 8 * FS_XR * (FS_WEXT / 4))
            *reinterpret_cast<float4*>(lds + f * 4) =
                make_float4(__uint_as_float(v[i][0]), __uint_as_float(v[i][1]), __uint_as_float(v[i][2]), __uint_as_float(v[i][3]));
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// [G; s]: partial[wg][tile][256] accumulator fragments (tile (t, u), t <= u: rows 16t.., columns 16u..)
// ---------------------------------------------------------------------------------------------------------------------
struct FsGramP {
    const float* x;
    float* part;
    int N, H, W;
    int nblocks;            // N * (H / 8) * (W / 64)
};

__global__ __launch_bounds__(256, 2) void fs_gram_kernel(const FsGramP p) {
    __shared__ __attribute__((aligned(16))) float lds[8 * FS_XR * FS_WEXT > 3 * FS_NTILE * 256 ? 8 * FS_XR * FS_WEXT : 3 * FS_NTILE * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    // lane constant of k-tile t: offset of xcol[16t + fr] at (row 0, column 16 * wave + fk) of the block
    int goff[5];
    float gone[5];            // 1 where the lane's row is the constant-1 row (kk == 72), else 0; rows past it stay 0
    bool gld[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int kk = 16 * t + fr;
        const int kc = kk < FS_K ? kk : 0;
        const int ch = kc / 9, tap = kc - ch * 9, kh = tap / 3, kw = tap - kh * 3;
        goff[t] = (ch * FS_XR + kh) * FS_WEXT + FS_DPAD + 16 * wave + fk + (kw - 1);
        gld[t] = kk < FS_K;
        gone[t] = kk == FS_K ? 1.f : 0.f;
    }
    floatx4 acc[FS_NTILE];
#pragma unroll
    for (int i = 0; i < FS_NTILE; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const int wq = p.W >> 6, hq = p.H >> 3;
    // the next block's tile is requested into registers before this block's MFMAs and written to LDS after them
    uintx4f xv[FS_XI];
    auto request = [&](int b) __attribute__((always_inline)) {
        const int n = b / (hq * wq), rem = b - n * (hq * wq);
        fs_load_x(p.x, xv, n, (rem / wq) * 8, (rem % wq) * 64, p.H, p.W, tid);
    };
    if ((int)blockIdx.x < p.nblocks) request(blockIdx.x);
    for (int b = blockIdx.x; b < p.nblocks; b += gridDim.x) {
        __syncthreads();                                   // everybody is done with the previous block's tile
        fs_store_x(lds, xv, tid);
        __syncthreads();
        if (b + (int)gridDim.x < p.nblocks) request(b + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);                 // (the compiler would sink the requests to behind the MFMAs)
#pragma unroll 1
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                  // 4-column groups of this wave's 16 columns
                float v[5];
#pragma unroll
                for (int t = 0; t < 5; ++t) v[t] = gld[t] ? lds[goff[t] + r * FS_WEXT + 4 * j] : gone[t];
#pragma unroll
                for (int t = 0; t < 5; ++t)
#pragma unroll
                    for (int u = t; u < 5; ++u)
                        acc[fs_tri(t, u)] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[t], v[u], acc[fs_tri(t, u)], 0, 0, 0);
            }
        }
    }
    // ---- the four waves' tiles -> one partial per workgroup (waves 1..3 through LDS, added by wave 0 in wave order) ----
    __syncthreads();
    if (wave > 0) {
        float* dst = lds + (wave - 1) * FS_NTILE * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < FS_NTILE; ++i) *reinterpret_cast<floatx4*>(dst + i * 256) = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = p.part + (size_t)blockIdx.x * FS_NTILE * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < FS_NTILE; ++i) {
            floatx4 a = acc[i];
#pragma unroll
            for (int w = 0; w < 3; ++w) a += *reinterpret_cast<const floatx4*>(lds + w * FS_NTILE * 256 + i * 256 + lane * 4);
            *reinterpret_cast<floatx4*>(out + i * 256) = a;
        }
    }
}

// Ordered sum of `nparts` partial buffers of `n` floats each: red[e] = sum_w part[w][e] in double.  One workgroup per 64
// consecutive elements; its four waves take the partials w = wave, wave + 4, ... (coalesced 256-byte reads) and their four
// sums are added in wave order: the same order on every run.
__global__ __launch_bounds__(256) void fs_sum_parts_kernel(const float* __restrict__ part, int nparts, int n, double* __restrict__ red) {
    __shared__ double sh[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    double a = 0.0;
    if (e < n) {
        // eight loads in flight, added in index order (the chain of adds is what fixes the result, not the load order):
        // one dependent load per add made this a 35 us kernel for 3.9 MB
        int w = wave;
        for (; w + 28 < nparts; w += 32) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(w + 4 * k) * n + e];
#pragma unroll
            for (int k = 0; k < 8; ++k) a += (double)v[k];
        }
        for (; w < nparts; w += 4) a += (double)part[(size_t)w * n + e];
    }
    sh[wave][lane] = a;
    __syncthreads();
    if (wave == 0 && e < n) red[e] = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
}
// gram[i][j], 0 <= i, j < 80 (double, symmetric; row / column 72 = s, [72][72] = position count) from the summed tiles
__global__ __launch_bounds__(256) void fs_gram_fold_kernel(const double* __restrict__ tiles, double* __restrict__ gram) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= FS_KP * FS_KP) return;
    const int i = e / FS_KP, j = e - i * FS_KP;
    const int lo = i < j ? i : j, hi = i < j ? j : i;           // stored: row in the lower-index tile
    const int t = lo >> 4, u = hi >> 4;
    int row, col;
    if (t == u) { row = i & 15; col = j & 15; }                  // diagonal tile holds the full 16 x 16 block
    else { row = lo & 15; col = hi & 15; }
    const int ln = (row >> 2) * 16 + col;
    gram[e] = tiles[(size_t)fs_tri(t, u) * 256 + ln * 4 + (row & 3)];
}

// ---------------------------------------------------------------------------------------------------------------------
// row c of the real weight matrix, W[c][ch * 9 + tap] = sign * component weight (quaternion_ops.py:131-135,
// dual_quaternion_ops.py:122-140; algebra 1: the weight itself)
// ---------------------------------------------------------------------------------------------------------------------
struct FsW {
    const float* w[8];
    int A, OB, IB;          // algebra, Cout / A, 8 / A
};
__device__ __forceinline__ float fs_wfull(const FsW& f, int c, int kk) {
    const int ch = kk / 9, tap = kk - ch * 9;
    const int po = c / f.OB, ob = c - po * f.OB;
    const int qi = ch / f.IB, ib = ch - qi * f.IB;
    float sign;
    const int comp = block_comp(f.A, po, qi, &sign);
    if (comp < 0) return 0.f;
    return sign * f.w[comp][((size_t)ob * f.IB + ib) * 9 + tap];
}

struct FsBnP {
    FsW w;
    const double* gram;
    const float* bias;      // nullable
    float* mean;
    float* invstd;
    float* rmean;           // nullable
    float* rvar;
    long long* nbt;         // nullable
    float* wg;              // nullable: (Cout, 72) floats, row c = W_c G (kept for the backward pass)
    int C;
    float eps, momentum;
};
// one workgroup per channel: mean = w.s / P + bias, E[(y - bias)^2] = w^T G w / P, in double
__global__ __launch_bounds__(256) void fs_bn_from_gram_kernel(const FsBnP p) {
    __shared__ double wrow[FS_K];
    __shared__ double red1[4], red2[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    if (tid < FS_K) wrow[tid] = (double)fs_wfull(p.w, c, tid);
    __syncthreads();
    double m2 = 0.0, m1 = 0.0;
    if (tid < FS_K) {
        double row = 0.0;
        for (int k = 0; k < FS_K; ++k) row += wrow[k] * p.gram[tid * FS_KP + k];
        if (p.wg) p.wg[(size_t)c * FS_K + tid] = (float)row;
        m2 = wrow[tid] * row;
        m1 = wrow[tid] * p.gram[FS_K * FS_KP + tid];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m1 += __shfl_xor(m1, o, 64);
        m2 += __shfl_xor(m2, o, 64);
    }
    if ((tid & 63) == 0) { red1[tid >> 6] = m1; red2[tid >> 6] = m2; }
    __syncthreads();
    if (tid != 0) return;
    if (c == 0 && p.nbt) *p.nbt += 1;
    const double count = p.gram[FS_K * FS_KP + FS_K];
    const double t1 = (red1[0] + red1[1] + red1[2] + red1[3]) / count;       // mean of the convolution without bias
    double var = (red2[0] + red2[1] + red2[2] + red2[3]) / count - t1 * t1;
    if (var < 0.0) var = 0.0;
    const double m = t1 + (p.bias ? (double)p.bias[c] : 0.0);
    p.mean[c] = (float)m;
    p.invstd[c] = (float)(1.0 / sqrt(var + (double)p.eps));
    if (p.rmean) p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * (float)m;
    if (p.rvar) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * (float)unbiased;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward: per-channel reductions over the pooled-size tensors, reproducible (chunk partials + ordered sum)
//   dz = dout * scale * [out != 0]      xhat = (out / scale - beta) / gamma       v0 = sum dz xhat, v1 = sum dz
// (out = relu(a raw + b) * mask * scale is non-zero exactly where ReLU is open AND the Dropout kept the element: the
// stage's own output replays both decisions, no random numbers are redrawn)
// ---------------------------------------------------------------------------------------------------------------------
struct FsRedP {
    const float* dout;          // (N, C, PH, W): gradient w.r.t. the stage's output (behind the Dropout)
    const float* out;           // (N, C, PH, W): the stage's output, relu(a raw + b) * dropout mask * scale
    const float* raw;           // (N, C, PH, W): convolution output at the window's chosen row
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    float* part;                // [C][nchunk][2]
    int N, C, S;                // S = PH * W
    int nchunk;
    float scale;                // 1 / (1 - p) of the stage's Dropout (1: none)
};
__global__ __launch_bounds__(256) void fs_reduce_kernel(const FsRedP p) {
    const int c = blockIdx.y, chunk = blockIdx.x;
    const float mu = p.mean[c], is = p.invstd[c];
    // xhat from the stage's output: out / scale = gamma xhat + beta wherever out != 0; only a channel with gamma == 0 needs
    // the raw window value (the one case seld_hcq_first_pool_bn writes it for)
    const float ga = p.gamma[c], be = p.beta[c];
    const bool degenerate = ga == 0.f;
    const float inv_g = degenerate ? 0.f : 1.0f / ga, inv_scale = 1.0f / p.scale;
    const long long total = (long long)p.N * p.S;                 // elements of this channel, S % 4 == 0
    const long long per = ((total / 4 + p.nchunk - 1) / p.nchunk) * 4;
    const long long beg = (long long)chunk * per;
    long long end = beg + per;
    if (end > total) end = total;
    float v0 = 0.f, v1 = 0.f;
    for (long long k = beg + threadIdx.x * 4; k < end; k += 256 * 4) {
        const long long n = k / p.S;
        const size_t off = ((size_t)n * p.C + c) * p.S + (size_t)(k - n * p.S);
        float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (degenerate) r4 = *reinterpret_cast<const float4*>(p.raw + off);
        const float4 d4 = *reinterpret_cast<const float4*>(p.dout + off);
        const float4 o4 = *reinterpret_cast<const float4*>(p.out + off);
        const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w}, oo[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (oo[e] != 0.f) {
                const float dz = dd[e] * p.scale;
                v0 += dz * (degenerate ? (rr[e] - mu) * is : (oo[e] * inv_scale - be) * inv_g);
                v1 += dz;
            }
    }
    __shared__ float r0[4], r1[4];
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = v0; r1[threadIdx.x >> 6] = v1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = p.part + ((size_t)c * p.nchunk + chunk) * 2;
        o[0] = (r0[0] + r0[1]) + (r0[2] + r0[3]);
        o[1] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
    }
}
// chunks summed in order; dgamma / dbeta ADDED to their gradient slots; coef = [c1 | a | c0] with dy = c1 y + a dz + c0
__global__ void fs_coef_kernel(const float* __restrict__ part, int nchunk, const float* __restrict__ mean,
                               const float* __restrict__ invstd, const float* __restrict__ gamma, int C, double count,
                               float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < nchunk; ++k) {
        s0 += (double)part[((size_t)c * nchunk + k) * 2];
        s1 += (double)part[((size_t)c * nchunk + k) * 2 + 1];
    }
    dgamma[c] += (float)s0;
    dbeta[c] += (float)s1;
    const float mu = mean[c], is = invstd[c], a = gamma[c] * is;
    const float k1 = (float)(s1 / count), k2 = (float)(s0 / count);
    coef[c] = -a * is * k2;
    coef[C + c] = a;
    coef[2 * C + c] = a * (mu * is * k2 - k1);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward: sum_pos (a dz)(pos) xcol(pos)^T on the matrix cores -- rows = output channels, columns = the 72 (+8) xcol rows.
// A workgroup (8 waves) walks blocks of 8 window rows x 64 columns.  Wave (hc, cq) owns HALF of the channel tiles
// (hc: for the dual quaternion half of the primal tiles + half of the dual ones, so both halves issue the same number of
// MFMAs -- primal channels do not see the dual input: column tiles 3, 4 are skipped for them) and a QUARTER of the columns
// (cq): a k-step of 4 positions is 5 LDS reads of xcol for 24 MFMAs (with every wave on all positions and a sixth of the
// channels it was 5 reads per 12 and LDS-bound on bank conflicts: 695 us, r3l).  The A operand of window row r is the
// pooled-size value masked by (arg-max row == r): loaded once per 4 columns, one group ahead, used for 8 rows.
// ---------------------------------------------------------------------------------------------------------------------
struct FsWgP {
    const float* x;
    const float* dout;
    const float* out;           // the stage's output: non-zero where ReLU is open and the Dropout kept the element
    const unsigned char* idx;
    const float* invstd;
    const float* gamma;
    float* part;                // [nwg][ctiles][5][256]
    int N, C, H, W;
    int nblocks;
    float scale;                // 1 / (1 - p)
};

// NTW channel tiles per wave (C / 32); DQ: the first NTW / 2 of them are primal (narrow)
template <int NTW, bool DQ>
__global__ __launch_bounds__(512, 2) void fs_wgrad_kernel(const FsWgP p) {
    constexpr int NARROW = DQ ? NTW / 2 : 0;
    constexpr int NACC = NARROW * 3 + (NTW - NARROW) * 5;
    constexpr int XT = 8 * FS_XR * FS_WEXT;
    constexpr int RED = 2 * NACC * 256;                       // one reduction round: both channel halves
    __shared__ __attribute__((aligned(16))) float lds[(2 * XT > RED ? 2 * XT : RED) + 192];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hc = wave >> 2, cq = wave & 3;
    const int fr = lane & 15, fk = lane >> 4;
    int goff[5];
    bool gld[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int kk = 16 * t + fr;
        const int kc = kk < FS_K ? kk : 0;
        const int ch = kc / 9, tap = kc - ch * 9, kh = tap / 3, kw = tap - kh * 3;
        goff[t] = (ch * FS_XR + kh) * FS_WEXT + FS_DPAD + 16 * cq + fk + (kw - 1);
        gld[t] = kk < FS_K;
    }
    // this wave's channel tiles (tile index in [0, C / 16), wave-uniform); a = gamma invstd (times the Dropout's scale) of
    // every channel in LDS, behind the two x tiles
    auto ctile = [&](int i) __attribute__((always_inline)) -> int {
        if (DQ) return i < NARROW ? hc * NARROW + i : NTW + hc * NARROW + (i - NARROW);          // NTW = C / 32 = primal tiles
        return hc * NTW + i;
    };
    float* const ab = lds + 2 * XT;
    for (int c = tid; c < p.C; c += 512) ab[c] = p.gamma[c] * p.invstd[c] * p.scale;
    auto acc_idx = [](int i, int u) constexpr { return i < NARROW ? i * 3 + u : NARROW * 3 + (i - NARROW) * 5 + u; };
    floatx4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const int wq = p.W >> 6, hq = p.H >> 3, PH = hq;
    const size_t PS = (size_t)PH * p.W;
    // The x tile of the NEXT block is fetched into registers while this block is multiplied and written to the other LDS
    // buffer at the end (one workgroup per CU at this register count: nobody else would cover a tile's memory latency).
    constexpr int QW = FS_WEXT / 4, NQ = 8 * FS_XR * QW, NST = (NQ + 511) / 512;
    float4 stage[NST];
    auto fetch_tile = [&](int b) __attribute__((always_inline)) {
        const int n = b / (hq * wq), rem = b - n * (hq * wq);
        const int h0 = (rem / wq) * 8, w0 = (rem % wq) * 64;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int f = tid + 512 * k;
            const int row = f / QW, quad = f - row * QW;
            const int ch = row / FS_XR, xr = row - ch * FS_XR;
            const int hh = h0 - 1 + xr, ww = w0 - FS_DPAD + 4 * quad;
            stage[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < NQ && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W)
                stage[k] = *reinterpret_cast<const float4*>(p.x + (((size_t)n * 8 + ch) * p.H + hh) * p.W + ww);
        }
    };
    auto store_tile = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int f = tid + 512 * k;
            if (f < NQ) *reinterpret_cast<float4*>(buf + f * 4) = stage[k];
        }
    };
    // pooled-size operands of 4-column group jg of this wave's 16 columns in block b: lane (fr, fk) = (channel, column).
    // Always one group AHEAD in registers -- across block boundaries too: a group's 3 * NTW loads come from HBM (the
    // pooled-size tensors are 0.45 GB) and need a whole group's MFMAs (2.6 us) to land.
    // (the loads of the NEXT group are issued -- all 3 * NTW unconditionally, no select that would make one wait for another
    // -- before this group's MFMAs and turned into operands after them: r3o's version waited for memory tile by tile)
    float vz[NTW];
    unsigned rpk = 0;                                             // the NTW arg-max rows of the current group, 3 bits each
    float ld_o[NTW], ld_d[NTW];
    unsigned ld_i[NTW];
    auto issue_group = [&](int b, int jg) __attribute__((always_inline)) {
        const int n = b / (hq * wq), rem = b - n * (hq * wq);
        const int q = rem / wq, w0 = (rem % wq) * 64;
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int chn = ctile(i) * 16 + fr;
            const size_t off = ((size_t)n * p.C + chn) * PS + (size_t)q * p.W + (size_t)(w0 + 16 * cq + 4 * jg + fk);
            ld_o[i] = p.out[off];
            ld_d[i] = p.dout[off];
            ld_i[i] = p.idx[off];
        }
    };
    auto finish_group = [&]() __attribute__((always_inline)) {
        unsigned pk = 0;
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const float a_ = ab[ctile(i) * 16 + fr];
            vz[i] = ld_o[i] != 0.f ? ld_d[i] * a_ : 0.f;
            pk |= (ld_i[i] & 7u) << (3 * i);
        }
        rpk = pk;
    };
    int bufsel = 0;
    if ((int)blockIdx.x < p.nblocks) { fetch_tile(blockIdx.x); store_tile(lds); }
    __syncthreads();                                              // (also: the a-table `ab` is complete)
    if ((int)blockIdx.x < p.nblocks) issue_group(blockIdx.x, 0);
    for (int b = blockIdx.x; b < p.nblocks; b += gridDim.x) {
        const float* xt = lds + bufsel * XT;
        const bool more = b + (int)gridDim.x < p.nblocks;
        if (more) fetch_tile(b + gridDim.x);
        float bv[5], bn_[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) bv[u] = gld[u] ? xt[goff[u]] : 0.f;
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
            finish_group();
            if (jg + 1 < 4) issue_group(b, jg + 1);
            else if (more) issue_group(b + gridDim.x, 0);
#pragma unroll 2
            for (int r = 0; r < 8; ++r) {
                // the xcol values of the NEXT k-step (next row; behind row 7: row 0 of the next column group) under this one's MFMAs
                const int rn = r + 1 < 8 ? r + 1 : 0, jn = r + 1 < 8 ? jg : (jg + 1 < 4 ? jg + 1 : 0);
#pragma unroll
                for (int u = 0; u < 5; ++u) bn_[u] = gld[u] ? xt[goff[u] + rn * FS_WEXT + 4 * jn] : 0.f;
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const float av = ((rpk >> (3 * i)) & 7u) == (unsigned)r ? vz[i] : 0.f;
#pragma unroll
                    for (int u = 0; u < (i < NARROW ? 3 : 5); ++u)
                        acc[acc_idx(i, u)] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[u], acc[acc_idx(i, u)], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) bv[u] = bn_[u];
            }
        }
        if (more) store_tile(lds + (bufsel ^ 1) * XT);
        bufsel ^= 1;
        __syncthreads();
    }
    // ---- the four column-quarter waves of each channel half -> one partial per workgroup (quarters 1..3 through LDS, added
    // by quarter 0 in order) ---------------------------------------------------------------------------------------------
#pragma unroll 1
    for (int k = 1; k < 4; ++k) {
        __syncthreads();
        if (cq == k) {
            float* dst = lds + hc * NACC * 256 + lane * 4;
#pragma unroll
            for (int i = 0; i < NACC; ++i) *reinterpret_cast<floatx4*>(dst + i * 256) = acc[i];
        }
        __syncthreads();
        if (cq == 0) {
            const float* src = lds + hc * NACC * 256 + lane * 4;
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] += *reinterpret_cast<const floatx4*>(src + i * 256);
        }
    }
    if (cq == 0) {
        float* out = p.part + (size_t)blockIdx.x * (2 * NTW) * 5 * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                floatx4 v = (floatx4){0.f, 0.f, 0.f, 0.f};
                if (u < (i < NARROW ? 3 : 5)) v = acc[acc_idx(i, u)];
                *reinterpret_cast<floatx4*>(out + (ctile(i) * 5 + u) * 256) = v;
            }
    }
}

// dWf[c][kk] = (workgroup partials, summed in order by fs_sum_parts_kernel) + c1_c (W G + bias s)[c][kk] + c0_c s[kk]
struct FsWsumP {
    const double* sum;          // [ctiles][5][256]
    const float* wg;            // (C, 72): W G
    const double* gram;
    const float* coef;          // [c1 | a | c0]
    const float* bias;          // nullable: the convolution's bias (y = W xcol + bias, so sum y xcol = W G + bias s)
    float* dwf;                 // (C, 72)
    int C;
};
__global__ __launch_bounds__(256) void fs_wsum_kernel(const FsWsumP p) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p.C * FS_K) return;
    const int c = e / FS_K, kk = e - c * FS_K;
    const int ct = c >> 4, row = c & 15, u = kk >> 4, col = kk & 15;
    double a = p.sum[((size_t)ct * 5 + u) * 256 + ((row >> 2) * 16 + col) * 4 + (row & 3)];
    const double sk = p.gram[FS_K * FS_KP + kk];
    a += (double)p.coef[c] * ((double)p.wg[e] + (p.bias ? (double)p.bias[c] * sk : 0.0)) + (double)p.coef[2 * p.C + c] * sk;
    p.dwf[e] = (float)a;
}
// component gradients: element (comp, ob, ib, tap) += sum over the blocks (po, qi) of the real matrix that hold it
struct FsWfoldP {
    const float* dwf;
    float* dw[8];
    int A, OB, IB, C;
};
__global__ __launch_bounds__(256) void fs_wfold_kernel(const FsWfoldP p) {
    const int per = p.OB * p.IB * 9;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p.A * per) return;
    const int comp = e / per;
    int rem = e - comp * per;
    const int ob = rem / (p.IB * 9);
    rem -= ob * p.IB * 9;
    const int ib = rem / 9, tap = rem - ib * 9;
    float a = 0.f;
    for (int po = 0; po < p.A; ++po)
        for (int qi = 0; qi < p.A; ++qi) {
            float sign;
            if (block_comp(p.A, po, qi, &sign) != comp) continue;
            a += sign * p.dwf[(size_t)(po * p.OB + ob) * FS_K + (qi * p.IB + ib) * 9 + tap];
        }
    p.dw[comp][((size_t)ob * p.IB + ib) * 9 + tap] += a;
}

static int fs_shape_ok(const seld_conv_desc* d) {
    if (hc_validate(d) != SELD_OK) return 0;
    if (d->ndim != 2 || d->Cin != 8 || d->k[0] != 3 || d->k[1] != 3) return 0;
    if (d->stride[0] != 1 || d->stride[1] != 1 || d->dil[0] != 1 || d->dil[1] != 1 || d->pad[0] != 1 || d->pad[1] != 1) return 0;
    if (d->in[0] % 8 || d->in[1] % 64) return 0;
    if ((long long)d->N * d->Cout * d->in[0] * d->in[1] * 4 >= (1LL << 32)) return 0;
    return 1;
}
static int fs_gram_wgs(const seld_conv_desc* d) {
    const long long nblocks = (long long)d->N * (d->in[0] / 8) * (d->in[1] / 64);
    return (int)(nblocks < 512 ? nblocks : 512);
}

}  // namespace seld
using namespace seld;

/* Scratch bytes of seld_first_stage_gram (0: not a first layer this path takes: 8 real input channels, 3x3 'same',
 * H % 8 == 0, W % 64 == 0).  Layout: [80 x 80 doubles: G, s in row / column 72, the position count at [72][72]] then
 * internal scratch (summed tiles, the workgroups' partial tiles). */
extern "C" size_t seld_first_stage_gram_workspace(const seld_conv_desc* d) {
    if (!fs_shape_ok(d)) return 0;
    return (size_t)FS_KP * FS_KP * sizeof(double) + (size_t)FS_NTILE * 256 * sizeof(double) +
           (size_t)fs_gram_wgs(d) * FS_NTILE * 256 * sizeof(float);
}

/* Second moments of the 3x3 neighbourhoods of x (N, 8, H, W): gram = first 80 x 80 doubles of `workspace`.  They depend on
 * the input only -- replaces the pass over the convolution output that torch.nn.BatchNorm2d's batch statistics make
 * (model.py:278-279). */
extern "C" int seld_first_stage_gram(const seld_conv_desc* d, const float* x, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !x || !workspace) return SELD_EINVAL;
    const size_t need = seld_first_stage_gram_workspace(d);
    if (!need) return SELD_EUNSUPPORTED;
    if (workspace_bytes < need || ((uintptr_t)workspace & 15)) return SELD_EWORKSPACE;
    FsGramP p{};
    p.x = x; p.N = d->N; p.H = d->in[0]; p.W = d->in[1];
    p.nblocks = d->N * (d->in[0] / 8) * (d->in[1] / 64);
    double* gram = (double*)workspace;
    double* tiles = gram + FS_KP * FS_KP;
    p.part = (float*)(tiles + FS_NTILE * 256);
    const int nwg = fs_gram_wgs(d);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fs_gram_kernel, dim3(nwg), dim3(256), 0, st, p);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(fs_sum_parts_kernel, dim3(FS_NTILE * 256 / 64), dim3(256), 0, st, p.part, nwg, FS_NTILE * 256, tiles);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(fs_gram_fold_kernel, dim3((FS_KP * FS_KP + 255) / 256), dim3(256), 0, st, tiles, gram);
    return check_launch();
}

/* BatchNorm2d batch statistics of the first layer's convolution output from the input's second moments (gram, as left by
 * seld_first_stage_gram) and the weights: mean / invstd (Cout), the running statistics and num_batches_tracked updated as
 * seld_bn_finalize_ex does; wg (nullable, Cout x 72): W G for the backward pass. */
extern "C" int seld_first_stage_bn(const seld_conv_desc* d, const float* const w[8], const float* bias, const double* gram,
                                   float eps, float momentum, float* mean, float* invstd, float* running_mean,
                                   float* running_var, int64_t* num_batches_tracked, float* wg, void* stream) {
    if (!d || !w || !gram || !mean || !invstd) return SELD_EINVAL;
    if (!fs_shape_ok(d)) return SELD_EUNSUPPORTED;
    FsBnP p{};
    p.w.A = d->algebra; p.w.OB = d->Cout / d->algebra; p.w.IB = 8 / d->algebra;
    for (int i = 0; i < d->algebra; ++i) {
        if (!w[i]) return SELD_EINVAL;
        p.w.w[i] = w[i];
    }
    p.gram = gram; p.bias = bias; p.mean = mean; p.invstd = invstd; p.rmean = running_mean; p.rvar = running_var;
    p.nbt = (long long*)num_batches_tracked; p.wg = wg; p.C = d->Cout; p.eps = eps; p.momentum = momentum;
    hipLaunchKernelGGL(fs_bn_from_gram_kernel, dim3(d->Cout), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch();
}

static int fs_wg_count(const seld_conv_desc* d) {
    // one 512-thread workgroup fits a CU at this kernel's register count: one per CU, each walking nblocks / 256 blocks
    const long long nblocks = (long long)d->N * (d->in[0] / 8) * (d->in[1] / 64);
    return (int)(nblocks < 256 ? nblocks : 256);
}
static int fs_red_chunks(const seld_conv_desc* d) {
    const long long total = (long long)d->N * (d->in[0] / 8) * d->in[1];
    long long n = total / 8192;
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return (int)n;
}

/* Scratch bytes of seld_first_stage_bwd (0: shape not taken -- additionally Cout % 64 == 0 and Cout <= 192). */
extern "C" size_t seld_first_stage_bwd_workspace(const seld_conv_desc* d) {
    if (!fs_shape_ok(d) || d->Cout % 64 || d->Cout > 192) return 0;
    if (d->algebra != 8 && d->Cout != 64) return 0;               // wider real / quaternion first layers: not instantiated
    const int C = d->Cout;
    return (size_t)C * fs_red_chunks(d) * 2 * sizeof(float) + (size_t)3 * C * sizeof(float) + (size_t)C * FS_K * sizeof(float) +
           (size_t)(C / 16) * 5 * 256 * sizeof(double) + (size_t)fs_wg_count(d) * (C / 16) * 5 * 256 * sizeof(float) + 64;
}

/* Backward pass of the first stage (training mode, batch statistics) WITHOUT the convolution output:
 *   dout (N, Cout, H/8, W) gradient w.r.t. the stage's output `out` (= relu(a raw + b) * Dropout mask / (1 - drop_p); its
 *   zeros replay ReLU and the Dropout, drop_p = 0: none); raw / idx as written by seld_hcq_first_pool[_bn] (raw is READ only
 *   for channels with gamma == 0: everywhere else xhat comes back from out); mean / invstd from
 *   seld_first_stage_bn, wg = W G from there, gram from seld_first_stage_gram; bias: the convolution's (nullable).
 * Adds the BatchNorm weight / bias gradients to dgamma / dbeta and the convolution's component weight gradients to dw[c]
 * (torch autograd through model.py:273-283 in the reference).  No atomics: reproducible. */
extern "C" int seld_first_stage_bwd(const seld_conv_desc* d, const float* x, const float* dout, const float* out,
                                    const float* raw, const uint8_t* idx, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, const float* bias, const double* gram, const float* wg, float* dgamma, float* dbeta,
                                    float* const dw[8], float drop_p, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !x || !dout || !out || !raw || !idx || !mean || !invstd || !gamma || !beta || !gram || !wg || !dgamma || !dbeta || !dw ||
        !workspace)
        return SELD_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return SELD_EINVAL;
    const size_t need = seld_first_stage_bwd_workspace(d);
    if (!need) return SELD_EUNSUPPORTED;
    if (workspace_bytes < need || ((uintptr_t)workspace & 15)) return SELD_EWORKSPACE;
    const int C = d->Cout, H = d->in[0], W = d->in[1], PH = H / 8;
    const int nchunk = fs_red_chunks(d), nwg = fs_wg_count(d);
    float* red_part = (float*)workspace;
    float* coef = red_part + (size_t)C * nchunk * 2;
    float* dwf = coef + 3 * C;
    double* wsumd = (double*)(((uintptr_t)(dwf + (size_t)C * FS_K) + 15) & ~(uintptr_t)15);
    float* part = (float*)(wsumd + (size_t)(C / 16) * 5 * 256);
    const float scale = 1.0f / (1.0f - drop_p);
    hipStream_t st = (hipStream_t)stream;

    FsRedP rp{};
    rp.dout = dout; rp.out = out; rp.raw = raw; rp.mean = mean; rp.invstd = invstd; rp.gamma = gamma; rp.beta = beta; rp.part = red_part;
    rp.N = d->N; rp.C = C; rp.S = PH * W; rp.nchunk = nchunk; rp.scale = scale;
    hipLaunchKernelGGL(fs_reduce_kernel, dim3(nchunk, C), dim3(256), 0, st, rp);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(fs_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, red_part, nchunk, mean, invstd, gamma, C,
                       (double)d->N * H * W, dgamma, dbeta, coef);
    rc = check_launch();
    if (rc) return rc;

    FsWgP wp{};
    wp.x = x; wp.dout = dout; wp.out = out; wp.idx = idx; wp.invstd = invstd; wp.gamma = gamma;
    wp.part = part; wp.N = d->N; wp.C = C; wp.H = H; wp.W = W; wp.nblocks = d->N * PH * (W / 64);
    wp.scale = scale;
    const bool dq = d->algebra == 8;
    if (C == 192) hipLaunchKernelGGL((fs_wgrad_kernel<6, true>), dim3(nwg), dim3(512), 0, st, wp);
    else if (C == 128) hipLaunchKernelGGL((fs_wgrad_kernel<4, true>), dim3(nwg), dim3(512), 0, st, wp);
    else { if (dq) hipLaunchKernelGGL((fs_wgrad_kernel<2, true>), dim3(nwg), dim3(512), 0, st, wp); else hipLaunchKernelGGL((fs_wgrad_kernel<2, false>), dim3(nwg), dim3(512), 0, st, wp); }
    rc = check_launch();
    if (rc) return rc;
    const int nsum = (C / 16) * 5 * 256;
    hipLaunchKernelGGL(fs_sum_parts_kernel, dim3(nsum / 64), dim3(256), 0, st, part, nwg, nsum, wsumd);
    rc = check_launch();
    if (rc) return rc;

    FsWsumP sp{};
    sp.sum = wsumd; sp.wg = wg; sp.gram = gram; sp.coef = coef; sp.bias = bias; sp.dwf = dwf; sp.C = C;
    hipLaunchKernelGGL(fs_wsum_kernel, dim3((C * FS_K + 255) / 256), dim3(256), 0, st, sp);
    rc = check_launch();
    if (rc) return rc;
    FsWfoldP fp{};
    fp.dwf = dwf; fp.A = d->algebra; fp.OB = C / d->algebra; fp.IB = 8 / d->algebra; fp.C = C;
    for (int i = 0; i < d->algebra; ++i) {
        if (!dw[i]) return SELD_EINVAL;
        fp.dw[i] = dw[i];
    }
    hipLaunchKernelGGL(fs_wfold_kernel, dim3((d->algebra * fp.OB * fp.IB * 9 + 255) / 256), dim3(256), 0, st, fp);
    return check_launch();
}
