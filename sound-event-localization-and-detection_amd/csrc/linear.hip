// Real / quaternion / dual-quaternion linear layers on (rows, features) tensors for gfx950.
//
//   y[r][o] = sum_i x[r][i] * M(i, o) + b[o]
//
// M is the expanded real matrix of the layer; it is never built: element M(i, o) is read from the
// component tensors with the Hamilton sign (quaternion_ops.py:310-314) or, for the dual
// quaternion layer, the transposed block arrangement of dual_quaternion_ops.py:170-188.
// These layers are the two classifier heads (rows = B*T/8, <= 0.2 % of the model's flops), so a
// plain LDS-tiled fp32 VALU GEMM is used (64 x 64 tile, 4 x 4 per thread).
#include "common.h"
#include "env.h"

namespace seld {

struct LinP {
    int kind;        // 1 real, 4 quat, 8 dualq
    int in_f, out_f;
    int IA, OA;      // in/A, out/A
    WPtrs w;
};

__device__ __forceinline__ float lin_elem(const LinP& p, int i, int o) {
    if (p.kind == SELD_LIN_REAL) return p.w.p[0][(size_t)o * p.in_f + i];
    const int a = i / p.IA, c = i - a * p.IA;
    const int b = o / p.OA, oo = o - b * p.OA;
    float sign;
    int comp;
    if (p.kind == SELD_LIN_QUAT) comp = block_comp(4, b, a, &sign);   // block (in a, out b) = table[b][a]
    else comp = block_comp(8, a, b, &sign);                           // block (in a, out b) = table8[a][b]
    if (comp < 0) return 0.f;
    return sign * p.w.p[comp][(size_t)c * p.OA + oo];
}

// out[r][n] = sum_k A[r][k] * Bm(k, n)  (+ bias[n]);  TRANS=0: Bm(k,n) = M(k,n), K = in, Nn = out
//                                                      TRANS=1: Bm(k,n) = M(n,k), K = out, Nn = in
template <int TRANS>
__global__ __launch_bounds__(256) void linear_gemm_kernel(const LinP p, int rows, const float* __restrict__ A,
                                                          const float* __restrict__ bias, float* __restrict__ out) {
    const int K = TRANS ? p.out_f : p.in_f;
    const int Nn = TRANS ? p.in_f : p.out_f;
    __shared__ float As[16][64 + 4];
    __shared__ float Bs[16][64 + 4];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tr = (tid >> 4) * 4, tn = (tid & 15) * 4;
    float acc[4][4] = {};
    // B-tile fill without per-element index arithmetic: thread -> column nn = tid & 63, k rows kq, kq+4, kq+8, kq+12
    const int nn = tid & 63, kq = tid >> 6;
    const int kblk = (p.kind == SELD_LIN_REAL) ? 16 : (TRANS ? p.OA : p.IA);     // K extent of one component block
    const int nblk = (p.kind == SELD_LIN_REAL) ? Nn : (TRANS ? p.IA : p.OA);
    const bool blk_fast = (kblk % 16 == 0);
    const bool nvalid = n0 + nn < Nn;
    const int nb = nvalid ? (n0 + nn) / nblk : 0;
    const int nc = nvalid ? (n0 + nn) - nb * nblk : 0;
    // register-prefetched K loop: the tiles of step k0+16 are loaded while step k0 is multiplied (the loop was two
    // exposed memory round trips per 16-deep step: 2.4 us each, 24 of them for K = 384)
    float ra[4], rb[4];
    auto fetch = [&](int k0) __attribute__((always_inline)) {
        // A tile: 64 rows x 16 k (row-major source, k contiguous)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + 256 * t;
            const int r = e >> 4, k = e & 15;
            ra[t] = (r0 + r < rows && k0 + k < K) ? A[(size_t)(r0 + r) * K + k0 + k] : 0.f;
        }
        if (blk_fast) {
            // the 16-deep K tile lies inside one component block (kblk | 16): component and sign depend only on this
            // thread's column, the four elements are one strided walk -- no division per element
            const int ka = k0 / kblk, kc = k0 - ka * kblk;
            float sign = 1.f;
            int comp = 0;
            if (p.kind == SELD_LIN_QUAT) comp = TRANS ? block_comp(4, ka, nb, &sign) : block_comp(4, nb, ka, &sign);
            else if (p.kind == SELD_LIN_DUALQ) comp = TRANS ? block_comp(8, nb, ka, &sign) : block_comp(8, ka, nb, &sign);
            const bool live = nvalid && comp >= 0;
            // M(i, o) = w[comp][c_in * OA + c_out]: TRANS=0: i = K index (c_in = kc + kk), o = this column (c_out = nc)
            //                                       TRANS=1: o = K index (c_out = kc + kk), i = this column (c_in = nc)
            const float* wp = live ? (p.kind == SELD_LIN_REAL
                                          ? p.w.p[0] + (TRANS ? (size_t)(k0 + kq) * p.in_f + (n0 + nn) : (size_t)(n0 + nn) * p.in_f + k0 + kq)
                                          : p.w.p[comp] + (TRANS ? (size_t)nc * p.OA + kc + kq : (size_t)(kc + kq) * p.OA + nc))
                                   : nullptr;
            const size_t wstep = (p.kind == SELD_LIN_REAL) ? (TRANS ? (size_t)4 * p.in_f : 4) : (TRANS ? 4 : (size_t)4 * p.OA);
#pragma unroll
            for (int t = 0; t < 4; ++t) rb[t] = (live && k0 + kq + 4 * t < K) ? sign * wp[t * wstep] : 0.f;
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = kq + 4 * t;
                rb[t] = (nvalid && k0 + k < K) ? (TRANS ? lin_elem(p, n0 + nn, k0 + k) : lin_elem(p, k0 + k, n0 + nn)) : 0.f;
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + 256 * t;
            As[e & 15][e >> 4] = ra[t];
            Bs[kq + 4 * t][nn] = rb[t];
        }
        __syncthreads();
        if (k0 + 16 < K) fetch(k0 + 16);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[k][tr + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[k][tn + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + tr + i, n = n0 + tn + j;
            if (r < rows && n < Nn) out[(size_t)r * Nn + n] = acc[i][j] + (bias ? bias[n] : 0.f);
        }
}

// The same GEMM with a 48-deep K tile for the classifier heads of the wide models (384 features = 8 blocks of 48):
// a tile still lies inside one component block, the rows of A are loaded as float4 along K, and both LDS images are read
// 16 bytes at a time -- 8 barriers instead of 24 for K = 384 and a quarter of the LDS read instructions (the 16-deep
// kernel above is latency-bound: 45 us for 0.6 GF).  Host: K % 48 == 0 and (real or component block % 48 == 0).
template <int TRANS>
__global__ __launch_bounds__(256) void linear_gemm48_kernel(const LinP p, int rows, const float* __restrict__ A,
                                                            const float* __restrict__ bias, float* __restrict__ out) {
    constexpr int KS = 48;
    const int K = TRANS ? p.out_f : p.in_f;
    const int Nn = TRANS ? p.in_f : p.out_f;
    __shared__ __attribute__((aligned(16))) float As[64][KS + 4];       // [row][k]
    __shared__ __attribute__((aligned(16))) float Bs[KS][64 + 16];      // [k][column]; 80 floats: the four k rows of a
                                                                        // B-operand read fall into four disjoint bank groups
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    // round 3: the products on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32): wave w owns rows 16w .. 16w + 15 of the
    // tile and its four 16-column tiles.  The VALU version (4 x 4 outputs per thread) spent 1.3 us of FMAs per 48-deep
    // chunk with one wave per SIMD: 34 us per head layer, the two heads' chains on the critical path of every step.
    const int lane = tid & 63, wave = tid >> 6;
    const int fm = lane & 15, fkk = lane >> 4;
    floatx4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (floatx4){0.f, 0.f, 0.f, 0.f};
    const int nn = tid & 63, kq = tid >> 6;
    const int kblk = (p.kind == SELD_LIN_REAL) ? KS : (TRANS ? p.OA : p.IA);
    const int nblk = (p.kind == SELD_LIN_REAL) ? Nn : (TRANS ? p.IA : p.OA);
    const bool nvalid = n0 + nn < Nn;
    const int nb = nvalid ? (n0 + nn) / nblk : 0;
    const int nc = nvalid ? (n0 + nn) - nb * nblk : 0;
    float4 ra[3];
    float rb[12];
    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int e = tid + 256 * t;
            const int r = e / 12, q = e - r * 12;
            ra[t] = (r0 + r < rows) ? *reinterpret_cast<const float4*>(A + (size_t)(r0 + r) * K + k0 + 4 * q)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int ka = k0 / kblk, kc = k0 - ka * kblk;
        float sign = 1.f;
        int comp = 0;
        if (p.kind == SELD_LIN_QUAT) comp = TRANS ? block_comp(4, ka, nb, &sign) : block_comp(4, nb, ka, &sign);
        else if (p.kind == SELD_LIN_DUALQ) comp = TRANS ? block_comp(8, nb, ka, &sign) : block_comp(8, ka, nb, &sign);
        const bool live = nvalid && comp >= 0;
        const float* wp = live ? (p.kind == SELD_LIN_REAL
                                      ? p.w.p[0] + (TRANS ? (size_t)(k0 + kq) * p.in_f + (n0 + nn) : (size_t)(n0 + nn) * p.in_f + k0 + kq)
                                      : p.w.p[comp] + (TRANS ? (size_t)nc * p.OA + kc + kq : (size_t)(kc + kq) * p.OA + nc))
                               : nullptr;
        const size_t wstep = (p.kind == SELD_LIN_REAL) ? (TRANS ? (size_t)4 * p.in_f : 4) : (TRANS ? 4 : (size_t)4 * p.OA);
#pragma unroll
        for (int t = 0; t < 12; ++t) rb[t] = live ? sign * wp[t * wstep] : 0.f;
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += KS) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int e = tid + 256 * t;
            const int r = e / 12, q = e - r * 12;
            *reinterpret_cast<float4*>(&As[r][4 * q]) = ra[t];
        }
#pragma unroll
        for (int t = 0; t < 12; ++t) Bs[kq + 4 * t][nn] = rb[t];
        __syncthreads();
        if (k0 + KS < K) fetch(k0 + KS);
        __builtin_amdgcn_sched_barrier(0);       // the next chunk's requests stay in front of this chunk's products
#pragma unroll
        for (int k4 = 0; k4 < KS; k4 += 4) {
            const float a = As[16 * wave + fm][k4 + fkk];              // A[row m][k]
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[k4 + fkk][16 * j + fm], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    // accumulator register r of tile j: row 16 wave + 4 fkk + r, column 16 j + fm
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 16 * j + fm;
        const float bv = (bias && n < Nn) ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + 16 * wave + 4 * fkk + r;
            if (row < rows && n < Nn) out[(size_t)row * Nn + n] = acc[j][r] + bv;
        }
    }
}

// dM[z][i][o] = sum over the rows of split z of x[r][i] * dy[r][o]; dB[z][o] = the column sums of dy over the same rows
// (written by the workgroups of the first input tile).  Plain stores, one writer per element: linear_fold_kernel adds the
// splits in a fixed order -- no zeroed buffer, no atomics, the same bits every run.
__global__ __launch_bounds__(256) void linear_wgrad_kernel(int rows, int rows_per_split, int in_f, int out_f, const float* __restrict__ x,
                                                           const float* __restrict__ dy, float* __restrict__ dM, float* __restrict__ dB) {
    __shared__ float Xs[16][64 + 4];
    __shared__ float Ds[16][64 + 4];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.y * 64, o0 = blockIdx.x * 64;
    const int ti = (tid >> 4) * 4, to = (tid & 15) * 4;
    float acc[4][4] = {};
    const int rbeg = blockIdx.z * rows_per_split;
    int rend = rbeg + rows_per_split;
    if (rend > rows) rend = rows;
    // register-prefetched: the tiles of the next 16 rows are loaded while these are multiplied
    const int cc = tid & 63, rq = tid >> 6;
    float rx[4], rd[4];
    auto fetch = [&](int r0) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = rq + 4 * t;
            rx[t] = (r0 + r < rend && i0 + cc < in_f) ? x[(size_t)(r0 + r) * in_f + i0 + cc] : 0.f;
            rd[t] = (r0 + r < rend && o0 + cc < out_f) ? dy[(size_t)(r0 + r) * out_f + o0 + cc] : 0.f;
        }
    };
    float bs = 0.f;
    if (rbeg < rend) fetch(rbeg);
    for (int r0 = rbeg; r0 < rend; r0 += 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            Xs[rq + 4 * t][cc] = rx[t];
            Ds[rq + 4 * t][cc] = rd[t];
            bs += rd[t];
        }
        __syncthreads();
        if (r0 + 16 < rend) fetch(r0 + 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = Xs[r][ti + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Ds[r][to + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ii = i0 + ti + i, oo = o0 + to + j;
            if (ii < in_f && oo < out_f) dM[((size_t)blockIdx.z * in_f + ii) * out_f + oo] = acc[i][j];
        }
    if (dB && blockIdx.y == 0) {                 // the loop's last barrier has passed: Xs is free
        Xs[rq][cc] = bs;
        __syncthreads();
        if (rq == 0 && o0 + cc < out_f) dB[(size_t)blockIdx.z * out_f + o0 + cc] = (Xs[0][cc] + Xs[1][cc]) + (Xs[2][cc] + Xs[3][cc]);
    }
}

// dw = the splits of dM added in order and folded onto the components; threads past the weights write dbias the same way
template <int KIND>
__global__ void linear_fold_kernel(const LinP p, const float* __restrict__ dM, const float* __restrict__ dB, int nz, WPtrsMut dw,
                                   float* __restrict__ dbias) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    // tab[a][comp] = b | (sign < 0 ? 16 : 0), or -1: inverse of the block pattern, built once per workgroup (scanning the
    // pattern per element cost 64 evaluations of block_comp + predicated sums: 15-20 us per layer)
    __shared__ int tab[64];
    if (KIND != SELD_LIN_REAL && threadIdx.x < KIND * KIND) {
        const int a = threadIdx.x / KIND, comp = threadIdx.x - a * KIND;
        int e = -1;
        for (int b = 0; b < KIND; ++b) {
            float sign = 0.f;
            const int cc = (KIND == SELD_LIN_QUAT) ? block_comp(4, b, a, &sign) : block_comp(8, a, b, &sign);
            if (cc == comp) e = b | (sign < 0.f ? 16 : 0);
        }
        tab[threadIdx.x] = e;
    }
    __syncthreads();
    const size_t zs = (size_t)p.in_f * p.out_f;
    auto at = [&](size_t e) {                     // the splits in order; all (at most 8) requests in flight together
        float v[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) v[z] = z < nz ? dM[z * zs + e] : 0.f;
        float t = 0.f;
#pragma unroll
        for (int z = 0; z < 8; ++z) t += v[z];
        return t;
    };
    const int nw = KIND == SELD_LIN_REAL ? p.in_f * p.out_f : p.IA * p.OA * KIND;
    if (idx >= nw) {
        const int o = idx - nw;
        if (dbias && o < p.out_f) {
            float t = 0.f;
            for (int z = 0; z < nz; ++z) t += dB[(size_t)z * p.out_f + o];
            dbias[o] = t;
        }
        return;
    }
    if (!dw.p[0]) return;
    if constexpr (KIND == SELD_LIN_REAL) {
        const int o = idx / p.in_f, i = idx - o * p.in_f;
        dw.p[0][idx] = at((size_t)i * p.out_f + o);
        return;
    } else {
        constexpr int A = KIND;
        const int per = p.IA * p.OA;
        const int comp = idx / per;
        const int rem = idx - comp * per;
        const int c = rem / p.OA, oo = rem - c * p.OA;
        float total = 0.f;
#pragma unroll 1
        for (int a = 0; a < A; ++a) {
            const int e = tab[a * A + comp];             // which output block b of input block a holds +-component `comp`
            if (e < 0) continue;
            const int b = e & 15;
            const float sign = (e & 16) ? -1.f : 1.f;
            total += sign * at((size_t)(a * p.IA + c) * p.out_f + (size_t)b * p.OA + oo);
        }
        dw.p[comp][rem] = total;
    }
}

static int mk_lin(LinP& p, int kind, int in_f, int out_f, const float* const w[8]) {
    if (kind != SELD_LIN_REAL && kind != SELD_LIN_QUAT && kind != SELD_LIN_DUALQ) return SELD_EINVAL;
    if (in_f <= 0 || out_f <= 0 || in_f % kind || out_f % kind) return SELD_EINVAL;
    p.kind = kind; p.in_f = in_f; p.out_f = out_f; p.IA = in_f / kind; p.OA = out_f / kind;
    for (int i = 0; i < 8; ++i) p.w.p[i] = (w && i < kind) ? w[i] : nullptr;
    return SELD_OK;
}

// the 48-deep K tile: K a multiple of 48 and every tile inside one component block
static bool lin_deep_tile(const LinP& p, int trans) {
    const int K = trans ? p.out_f : p.in_f;
    if (K % 48) return false;
    if (p.kind == SELD_LIN_REAL) return true;
    return (trans ? p.OA : p.IA) % 48 == 0;
}

}  // namespace seld
using namespace seld;

extern "C" int seld_hc_linear_fwd(int32_t kind, int32_t rows, int32_t in_features, int32_t out_features, const float* x,
                                  const float* const w[8], const float* bias, float* y, void* stream) {
    LinP p{};
    int rc = mk_lin(p, kind, in_features, out_features, w);
    if (rc) return rc;
    if (!x || !w || !y || rows <= 0) return SELD_EINVAL;
    dim3 grid((out_features + 63) / 64, (rows + 63) / 64);
    if (lin_deep_tile(p, 0)) hipLaunchKernelGGL((linear_gemm48_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, p, rows, x, bias, y);
    else hipLaunchKernelGGL((linear_gemm_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, p, rows, x, bias, y);
    return check_launch();
}

// row splits of the weight / bias gradient: at most LIN_SPLITS partial results, added in order by the fold
static constexpr int LIN_SPLITS = 8;        // linear_fold_kernel's `at` reads exactly this many (or fewer) partials
static int lin_rows_per_split(int rows) {
    int rps = (rows + LIN_SPLITS - 1) / LIN_SPLITS;
    rps = (rps + 15) / 16 * 16;
    return rps < 128 ? 128 : rps;
}

extern "C" size_t seld_hc_linear_bwd_workspace(int32_t kind, int32_t in_features, int32_t out_features) {
    (void)kind;
    if (in_features <= 0 || out_features <= 0) return 0;
    return (size_t)LIN_SPLITS * ((size_t)in_features * out_features + out_features) * sizeof(float);
}

extern "C" int seld_hc_linear_bwd(int32_t kind, int32_t rows, int32_t in_features, int32_t out_features, const float* x,
                                  const float* dy, const float* const w[8], float* dx, float* const dw[8], float* dbias,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    LinP p{};
    int rc = mk_lin(p, kind, in_features, out_features, w);
    if (rc) return rc;
    if (!dy || rows <= 0) return SELD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        if (!w) return SELD_EINVAL;
        dim3 grid((in_features + 63) / 64, (rows + 63) / 64);
        if (lin_deep_tile(p, 1)) hipLaunchKernelGGL((linear_gemm48_kernel<1>), grid, dim3(256), 0, st, p, rows, dy, (const float*)nullptr, dx);
        else hipLaunchKernelGGL((linear_gemm_kernel<1>), grid, dim3(256), 0, st, p, rows, dy, (const float*)nullptr, dx);
        rc = check_launch();
        if (rc) return rc;
    }
    if (dw || dbias) {
        if (dw && !x) return SELD_EINVAL;
        if (!workspace || workspace_bytes < seld_hc_linear_bwd_workspace(kind, in_features, out_features)) return SELD_EWORKSPACE;
        const int rps = lin_rows_per_split(rows);
        const int nz = (rows + rps - 1) / rps;
        float* dM = (float*)workspace;
        float* dB = dM + (size_t)LIN_SPLITS * in_features * out_features;
        if (dw) {
            dim3 grid((out_features + 63) / 64, (in_features + 63) / 64, nz);
            hipLaunchKernelGGL(linear_wgrad_kernel, grid, dim3(256), 0, st, rows, rps, in_features, out_features, x, dy, dM,
                               dbias ? dB : (float*)nullptr);
        } else {
            // bias only: the column sums alone (x is not read: the first input tile of a one-column input)
            dim3 grid((out_features + 63) / 64, 1, nz);
            hipLaunchKernelGGL(linear_wgrad_kernel, grid, dim3(256), 0, st, rows, rps, 0, out_features, dy, dy, dM, dB);
        }
        rc = check_launch();
        if (rc) return rc;
        WPtrsMut out{};
        for (int i = 0; i < 8; ++i) out.p[i] = (dw && i < kind) ? dw[i] : nullptr;
        const int total = in_features * out_features / (kind == SELD_LIN_REAL ? 1 : kind) + out_features;   // weights, then bias
        const dim3 fgrid((total + 255) / 256);
        if (kind == SELD_LIN_REAL)
            hipLaunchKernelGGL(linear_fold_kernel<SELD_LIN_REAL>, fgrid, dim3(256), 0, st, p, (const float*)dM, (const float*)dB, nz, out, dbias);
        else if (kind == SELD_LIN_QUAT)
            hipLaunchKernelGGL(linear_fold_kernel<SELD_LIN_QUAT>, fgrid, dim3(256), 0, st, p, (const float*)dM, (const float*)dB, nz, out, dbias);
        else
            hipLaunchKernelGGL(linear_fold_kernel<SELD_LIN_DUALQ>, fgrid, dim3(256), 0, st, p, (const float*)dM, (const float*)dB, nz, out, dbias);
        rc = check_launch();
    }
    return rc;
}
