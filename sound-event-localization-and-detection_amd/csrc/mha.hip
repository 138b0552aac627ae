// Multi-head self-attention core for gfx950, flash style (online softmax; the T x T energy tensor
// of model.py:40-46 is never materialised, which is what makes full-clip T = 2400 inference fit).
//
// Tensors are (N, E, T) -- the layout the surrounding 1x1 convolutions produce -- with head h on
// channels [h*hd, (h+1)*hd) (model.py:35-37).  out has the same layout; lse is (N, H, T).
//
// Attention is ~1 % of the model's flops and hd = E/8 is arbitrary (2 ... 64), so this is an
// fp32 VALU kernel: a query (forward, dQ) or a key (dK/dV) is owned by 8 lanes that split the
// 64-wide tile of the other index; row max / sums are 8-lane shuffles; K/V (or Q/dO) tiles are
// staged in LDS with coalesced loads along T.
#include "common.h"

namespace seld {

constexpr int TILE = 64;     // keys (queries) per LDS tile
constexpr int OWN = 32;      // queries (keys) owned by a workgroup: 4 waves x 8

__device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64));
    v = fmaxf(v, __shfl_xor(v, 2, 64));
    v = fmaxf(v, __shfl_xor(v, 4, 64));
    return v;
}
__device__ __forceinline__ float group8_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

template <int HDM>
__global__ __launch_bounds__(256) void mha_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, int T, int H, int hd, float scale,
                                                      float* __restrict__ out, float* __restrict__ lse) {
    __shared__ float Ks[HDM][TILE];
    __shared__ float Vs[HDM][TILE];
    __shared__ float Os[HDM][OWN + 1];
    __shared__ float Ss[OWN][TILE + 1];
    const int tid = threadIdx.x;
    const int nh = blockIdx.y;
    const int n = nh / H, h = nh - n * H;
    const int E = H * hd;
    const size_t base = ((size_t)n * E + (size_t)h * hd) * T;
    const int own = tid >> 3;            // 0..31: query owned by this lane group
    const int kl = tid & 7;
    const int tq = blockIdx.x * OWN + own;
    const bool qok = tq < T;

    float qv[HDM], o[HDM];
#pragma unroll
    for (int d = 0; d < HDM; ++d) {
        qv[d] = (qok && d < hd) ? q[base + (size_t)d * T + tq] * scale : 0.f;
        o[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;

    for (int k0 = 0; k0 < T; k0 += TILE) {
        __syncthreads();
        for (int e = tid; e < HDM * TILE; e += 256) {
            const int d = e / TILE, t = e - d * TILE;
            const bool ok = d < hd && k0 + t < T;
            Ks[d][t] = ok ? k[base + (size_t)d * T + k0 + t] : 0.f;
            Vs[d][t] = ok ? v[base + (size_t)d * T + k0 + t] : 0.f;
        }
        __syncthreads();
        float tmax = -INFINITY;
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            const int key = kl + 8 * j;
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < HDM; ++d) acc += qv[d] * Ks[d][key];
            acc = (k0 + key < T) ? acc : -INFINITY;
            Ss[own][key] = acc;                 // only this lane reads it back
            tmax = fmaxf(tmax, acc);
        }
        tmax = group8_max(tmax);
        const float mnew = fmaxf(m, tmax);
        const float alpha = (m == -INFINITY) ? 0.f : expf(m - mnew);
        l *= alpha;
#pragma unroll
        for (int d = 0; d < HDM; ++d) o[d] *= alpha;
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            const int key = kl + 8 * j;
            const float sj = Ss[own][key];
            const float p = (sj == -INFINITY) ? 0.f : expf(sj - mnew);
            l += p;
#pragma unroll
            for (int d = 0; d < HDM; ++d) o[d] += p * Vs[d][key];
        }
        m = mnew;
    }
    l = group8_sum(l);
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < HDM; ++d) {
        const float od = group8_sum(o[d]) * inv;
        if (kl == (d & 7)) Os[d][own] = od;
    }
    if (kl == 0 && qok) lse[(size_t)nh * T + tq] = m + logf(l);
    __syncthreads();
    for (int e = tid; e < HDM * OWN; e += 256) {
        const int d = e / OWN, t = e - d * OWN;
        const int tt = blockIdx.x * OWN + t;
        if (d < hd && tt < T) out[base + (size_t)d * T + tt] = Os[d][t];
    }
}

// delta[n][h][t] = sum_d dO[d][t] * O[d][t]
__global__ void mha_delta_kernel(const float* __restrict__ o, const float* __restrict__ dout, int T, int H, int hd,
                                 float* __restrict__ delta) {
    const int nh = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int n = nh / H, h = nh - n * H;
    const size_t base = ((size_t)n * H * hd + (size_t)h * hd) * T;
    float s = 0.f;
    for (int d = 0; d < hd; ++d) s += o[base + (size_t)d * T + t] * dout[base + (size_t)d * T + t];
    delta[(size_t)nh * T + t] = s;
}

// dQ: each query owned by 8 lanes, loop over key tiles
template <int HDM>
__global__ __launch_bounds__(256) void mha_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ dout,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         int T, int H, int hd, float scale, float* __restrict__ dq) {
    __shared__ float Ks[HDM][TILE];
    __shared__ float Vs[HDM][TILE];
    __shared__ float Os[HDM][OWN + 1];
    const int tid = threadIdx.x;
    const int nh = blockIdx.y;
    const int n = nh / H, h = nh - n * H;
    const size_t base = ((size_t)n * H * hd + (size_t)h * hd) * T;
    const int own = tid >> 3, kl = tid & 7;
    const int tq = blockIdx.x * OWN + own;
    const bool qok = tq < T;
    float qv[HDM], dov[HDM], acc[HDM];
#pragma unroll
    for (int d = 0; d < HDM; ++d) {
        const bool ok = qok && d < hd;
        qv[d] = ok ? q[base + (size_t)d * T + tq] * scale : 0.f;
        dov[d] = ok ? dout[base + (size_t)d * T + tq] : 0.f;
        acc[d] = 0.f;
    }
    const float my_lse = qok ? lse[(size_t)nh * T + tq] : 0.f;
    const float my_delta = qok ? delta[(size_t)nh * T + tq] : 0.f;
    for (int k0 = 0; k0 < T; k0 += TILE) {
        __syncthreads();
        for (int e = tid; e < HDM * TILE; e += 256) {
            const int d = e / TILE, t = e - d * TILE;
            const bool ok = d < hd && k0 + t < T;
            Ks[d][t] = ok ? k[base + (size_t)d * T + k0 + t] : 0.f;
            Vs[d][t] = ok ? v[base + (size_t)d * T + k0 + t] : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            const int key = kl + 8 * j;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HDM; ++d) {
                s += qv[d] * Ks[d][key];
                dp += dov[d] * Vs[d][key];
            }
            const float p = (qok && k0 + key < T) ? expf(s - my_lse) : 0.f;
            const float ds = p * (dp - my_delta) * scale;
#pragma unroll
            for (int d = 0; d < HDM; ++d) acc[d] += ds * Ks[d][key];
        }
    }
#pragma unroll
    for (int d = 0; d < HDM; ++d) {
        const float r = group8_sum(acc[d]);
        if (kl == (d & 7)) Os[d][own] = r;
    }
    __syncthreads();
    for (int e = tid; e < HDM * OWN; e += 256) {
        const int d = e / OWN, t = e - d * OWN;
        const int tt = blockIdx.x * OWN + t;
        if (d < hd && tt < T) dq[base + (size_t)d * T + tt] = Os[d][t];
    }
}

// dK, dV: each key owned by 8 lanes, loop over query tiles
template <int HDM>
__global__ __launch_bounds__(256) void mha_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ dout,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          int T, int H, int hd, float scale, float* __restrict__ dk,
                                                          float* __restrict__ dv) {
    __shared__ float Qs[HDM][TILE];
    __shared__ float Ds[HDM][TILE];
    __shared__ float Ls[TILE], Dl[TILE];
    __shared__ float Os[HDM][OWN + 1];
    const int tid = threadIdx.x;
    const int nh = blockIdx.y;
    const int n = nh / H, h = nh - n * H;
    const size_t base = ((size_t)n * H * hd + (size_t)h * hd) * T;
    const int own = tid >> 3, ql = tid & 7;
    const int tk = blockIdx.x * OWN + own;
    const bool kok = tk < T;
    float kv[HDM], vv[HDM], adk[HDM], adv[HDM];
#pragma unroll
    for (int d = 0; d < HDM; ++d) {
        const bool ok = kok && d < hd;
        kv[d] = ok ? k[base + (size_t)d * T + tk] : 0.f;
        vv[d] = ok ? v[base + (size_t)d * T + tk] : 0.f;
        adk[d] = 0.f;
        adv[d] = 0.f;
    }
    for (int q0 = 0; q0 < T; q0 += TILE) {
        __syncthreads();
        for (int e = tid; e < HDM * TILE; e += 256) {
            const int d = e / TILE, t = e - d * TILE;
            const bool ok = d < hd && q0 + t < T;
            Qs[d][t] = ok ? q[base + (size_t)d * T + q0 + t] * scale : 0.f;
            Ds[d][t] = ok ? dout[base + (size_t)d * T + q0 + t] : 0.f;
        }
        if (tid < TILE) {
            const bool ok = q0 + tid < T;
            Ls[tid] = ok ? lse[(size_t)nh * T + q0 + tid] : 0.f;
            Dl[tid] = ok ? delta[(size_t)nh * T + q0 + tid] : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
            const int qq = ql + 8 * j;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HDM; ++d) {
                s += Qs[d][qq] * kv[d];
                dp += Ds[d][qq] * vv[d];
            }
            const float p = (kok && q0 + qq < T) ? expf(s - Ls[qq]) : 0.f;
            const float ds = p * (dp - Dl[qq]);      // Qs already carries `scale`
#pragma unroll
            for (int d = 0; d < HDM; ++d) {
                adv[d] += p * Ds[d][qq];
                adk[d] += ds * Qs[d][qq];
            }
        }
    }
    // dK
#pragma unroll
    for (int d = 0; d < HDM; ++d) {
        const float r = group8_sum(adk[d]);
        if (ql == (d & 7)) Os[d][own] = r;
    }
    __syncthreads();
    for (int e = tid; e < HDM * OWN; e += 256) {
        const int d = e / OWN, t = e - d * OWN;
        const int tt = blockIdx.x * OWN + t;
        if (d < hd && tt < T) dk[base + (size_t)d * T + tt] = Os[d][t];
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < HDM; ++d) {
        const float r = group8_sum(adv[d]);
        if (ql == (d & 7)) Os[d][own] = r;
    }
    __syncthreads();
    for (int e = tid; e < HDM * OWN; e += 256) {
        const int d = e / OWN, t = e - d * OWN;
        const int tt = blockIdx.x * OWN + t;
        if (d < hd && tt < T) dv[base + (size_t)d * T + tt] = Os[d][t];
    }
}

template <int HDM>
static int launch_fwd(const float* q, const float* k, const float* v, int N, int T, int H, int hd, float* out, float* lse,
                      hipStream_t st) {
    dim3 grid((T + OWN - 1) / OWN, N * H);
    hipLaunchKernelGGL((mha_fwd_kernel<HDM>), grid, dim3(256), 0, st, q, k, v, T, H, hd, 1.0f / sqrtf((float)hd), out, lse);
    return check_launch();
}
template <int HDM>
static int launch_bwd(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                      const float* delta, int N, int T, int H, int hd, float* dq, float* dk, float* dv, hipStream_t st) {
    dim3 grid((T + OWN - 1) / OWN, N * H);
    const float scale = 1.0f / sqrtf((float)hd);
    hipLaunchKernelGGL((mha_bwd_dq_kernel<HDM>), grid, dim3(256), 0, st, q, k, v, dout, lse, delta, T, H, hd, scale, dq);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL((mha_bwd_dkv_kernel<HDM>), grid, dim3(256), 0, st, q, k, v, dout, lse, delta, T, H, hd, scale, dk, dv);
    return check_launch();
}

// fp32-MFMA kernels for head dims 16/32/48/64 and T % 16 == 0 (mha_mfma.hip)
bool mha_mfma_ok(int T, int hd);
int mha_mfma_fwd(const float* q, const float* k, const float* v, int N, int T, int H, int hd, long long in_bs, float* out, float* lse,
                 hipStream_t st);
int mha_mfma_bwd(const float* q, const float* k, const float* v, const float* dout, const float* lse, const float* delta,
                 int N, int T, int H, int hd, long long in_bs, float* dq, float* dk, float* dv, hipStream_t st);

}  // namespace seld
using namespace seld;

extern "C" int seld_mha_fwd(const float* q, const float* k, const float* v, int32_t N, int32_t T, int32_t H, int32_t hd,
                            float* out, float* lse, void* stream) {
    if (!q || !k || !v || !out || !lse || N <= 0 || T <= 0 || H <= 0 || hd <= 0) return SELD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (mha_mfma_ok(T, hd)) return mha_mfma_fwd(q, k, v, N, T, H, hd, (long long)H * hd * T, out, lse, st);
    if (hd <= 8) return launch_fwd<8>(q, k, v, N, T, H, hd, out, lse, st);
    if (hd <= 16) return launch_fwd<16>(q, k, v, N, T, H, hd, out, lse, st);
    if (hd <= 32) return launch_fwd<32>(q, k, v, N, T, H, hd, out, lse, st);
    if (hd <= 48) return launch_fwd<48>(q, k, v, N, T, H, hd, out, lse, st);
    if (hd <= 64) return launch_fwd<64>(q, k, v, N, T, H, hd, out, lse, st);
    return SELD_EUNSUPPORTED;
}

extern "C" size_t seld_mha_bwd_workspace(int32_t N, int32_t T, int32_t H) {
    if (N <= 0 || T <= 0 || H <= 0) return 0;
    return (size_t)N * H * T * sizeof(float);
}

extern "C" int seld_mha_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                            const float* lse, int32_t N, int32_t T, int32_t H, int32_t hd, float* dq, float* dk,
                            float* dv, void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || N <= 0 || T <= 0 || H <= 0 || hd <= 0)
        return SELD_EINVAL;
    if (!workspace || workspace_bytes < seld_mha_bwd_workspace(N, T, H)) return SELD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* delta = (float*)workspace;
    hipLaunchKernelGGL(mha_delta_kernel, dim3((T + 255) / 256, N * H), dim3(256), 0, st, out, dout, T, H, hd, delta);
    int rc = check_launch();
    if (rc) return rc;
    if (mha_mfma_ok(T, hd)) return mha_mfma_bwd(q, k, v, dout, lse, delta, N, T, H, hd, (long long)H * hd * T, dq, dk, dv, st);
    if (hd <= 8) return launch_bwd<8>(q, k, v, dout, lse, delta, N, T, H, hd, dq, dk, dv, st);
    if (hd <= 16) return launch_bwd<16>(q, k, v, dout, lse, delta, N, T, H, hd, dq, dk, dv, st);
    if (hd <= 32) return launch_bwd<32>(q, k, v, dout, lse, delta, N, T, H, hd, dq, dk, dv, st);
    if (hd <= 48) return launch_bwd<48>(q, k, v, dout, lse, delta, N, T, H, hd, dq, dk, dv, st);
    if (hd <= 64) return launch_bwd<64>(q, k, v, dout, lse, delta, N, T, H, hd, dq, dk, dv, st);
    return SELD_EUNSUPPORTED;
}

/* Self-attention on ONE projected tensor qkv (N, 3E, T) = [values | keys | queries] along the channels (the three 1x1
 * convolutions of model.py:31-33 run as one, hip_ops.QkvFn): same arithmetic as seld_mha_fwd / seld_mha_bwd on the three
 * channel slices, dqkv in the same layout.  Matrix-core kernels only: SELD_EUNSUPPORTED unless seld_mha_packed_ok(T, hd). */
extern "C" int seld_mha_packed_ok(int32_t T, int32_t hd) { return T > 0 && hd > 0 && mha_mfma_ok(T, hd); }

extern "C" int seld_mha_fwd_packed(const float* qkv, int32_t N, int32_t T, int32_t H, int32_t hd, float* out, float* lse,
                                   void* stream) {
    if (!qkv || !out || !lse || N <= 0 || T <= 0 || H <= 0 || hd <= 0) return SELD_EINVAL;
    if (!mha_mfma_ok(T, hd)) return SELD_EUNSUPPORTED;
    const size_t E = (size_t)H * hd * T;
    return mha_mfma_fwd(qkv + 2 * E, qkv + E, qkv, N, T, H, hd, (long long)(3 * E), out, lse, (hipStream_t)stream);
}

extern "C" int seld_mha_bwd_packed(const float* qkv, const float* out, const float* dout, const float* lse, int32_t N,
                                   int32_t T, int32_t H, int32_t hd, float* dqkv, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (!qkv || !out || !dout || !lse || !dqkv || N <= 0 || T <= 0 || H <= 0 || hd <= 0) return SELD_EINVAL;
    if (!mha_mfma_ok(T, hd)) return SELD_EUNSUPPORTED;
    if (!workspace || workspace_bytes < seld_mha_bwd_workspace(N, T, H)) return SELD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* delta = (float*)workspace;
    hipLaunchKernelGGL(mha_delta_kernel, dim3((T + 255) / 256, N * H), dim3(256), 0, st, out, dout, T, H, hd, delta);
    int rc = check_launch();
    if (rc) return rc;
    const size_t E = (size_t)H * hd * T;
    return mha_mfma_bwd(qkv + 2 * E, qkv + E, qkv, dout, lse, delta, N, T, H, hd, (long long)(3 * E), dqkv + 2 * E, dqkv + E, dqkv, st);
}
