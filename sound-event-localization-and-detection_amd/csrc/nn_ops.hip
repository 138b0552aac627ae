// BatchNorm (+ fused activations), gated activation, pooling, dropout, loss and Adam kernels
// for gfx950.  All HBM-bound: 16-byte loads/stores along the contiguous S axis, per-channel
// reductions by wave shuffles + one atomic per wave (guide: Appendix B "Reduction").
#include "common.h"
#include "env.h"

namespace seld {

__device__ __forceinline__ float act_apply(float z, int act) {
    switch (act) {
        case SELD_ACT_RELU: return z > 0.f ? z : 0.f;
        case SELD_ACT_TANH: return tanhf(z);
        case SELD_ACT_SIGMOID: return 1.0f / (1.0f + expf(-z));
        default: return z;
    }
}
// derivative expressed through the OUTPUT y = act(z)
__device__ __forceinline__ float act_grad_from_y(float y, int act) {
    switch (act) {
        case SELD_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case SELD_ACT_TANH: return 1.f - y * y;
        case SELD_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// ------------------------------------------------------------------------------------------
// per-channel sum / sum of squares over (N, S)
// grid: (chunks, C); each block walks `chunk` elements of channel c's N*S index space.
// ------------------------------------------------------------------------------------------
constexpr int RED_CHUNK = 8192;

template <typename F>
__device__ __forceinline__ void channel_walk(int N, int C, int S, int c, F&& f) {
    // calls f(offset_of_4_or_1_elements, count) for this block's slice of channel c
    const long long M = (long long)N * S;
    const long long chunk = gridDim.x == 1 ? M : RED_CHUNK;      // one workgroup per channel (SELD_DETERMINISTIC): the whole range
    const long long beg = (long long)blockIdx.x * chunk;
    long long end = beg + chunk;
    if (end > M) end = M;
    if ((S & 3) == 0) {
        for (long long i = beg + (long long)threadIdx.x * 4; i < end; i += (long long)blockDim.x * 4) {
            long long n = i / S;
            int s = (int)(i - n * S);
            f(((size_t)n * C + c) * S + s, 4);
        }
    } else {
        for (long long i = beg + threadIdx.x; i < end; i += blockDim.x) {
            long long n = i / S;
            int s = (int)(i - n * S);
            f(((size_t)n * C + c) * S + s, 1);
        }
    }
}

template <int NV>
__device__ __forceinline__ void block_atomic(float (&v)[NV], float* const (&dst)[NV]) {
    __shared__ float red[NV][4];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float s = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int k = threadIdx.x;
        atomicAdd(dst[k], red[k][0] + red[k][1] + red[k][2] + red[k][3]);
    }
}

__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, int N, int C, int S,
                                                            float* __restrict__ stats) {
    const int c = blockIdx.y;
    float v[2] = {0.f, 0.f};
    channel_walk(N, C, S, c, [&](size_t off, int cnt) {
        if (cnt == 4) {
            float4 a = *reinterpret_cast<const float4*>(x + off);
            v[0] += (a.x + a.y) + (a.z + a.w);
            v[1] += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        } else {
            float a = x[off];
            v[0] += a;
            v[1] += a * a;
        }
    });
    float* rep = stats + (size_t)(blockIdx.x % SELD_STATS_REPLICAS) * 2 * C;
    float* const dst[2] = {rep + c, rep + C + c};
    block_atomic<2>(v, dst);
}

// One WAVE per channel: lane r owns replica row r (SELD_STATS_REPLICAS == 64), the 64 partial sums are added in fp64
// with a shuffle tree.  (A thread per channel walking the 64 rows took 8 us -- per BatchNorm, 33 times a step.)
__global__ __launch_bounds__(256) void bn_finalize_kernel(float* __restrict__ stats, int C, double count, float eps,
                                                          float momentum, float* __restrict__ mean,
                                                          float* __restrict__ invstd, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, long long* __restrict__ nbt, int clear) {
    static_assert(SELD_STATS_REPLICAS == 64, "one lane per replica row");
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;      // num_batches_tracked (torch.nn.BatchNorm bookkeeping)
    if (c >= C) return;
    float* p1 = stats + (size_t)lane * 2 * C + c;
    float* p2 = p1 + C;
    double s1 = (double)*p1, s2 = (double)*p2;
    if (clear) { *p1 = 0.f; *p2 = 0.f; }                              // hand the buffer back zeroed (pooled by the host mirror)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (lane != 0) return;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    if (rvar) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
}

// Two BatchNorms of the same channel count in one launch (the gate's batch_filter2 / batch_gate2, whose statistics come
// from one pair convolution): blockIdx.y selects the layer.
struct BnFin {
    float* stats; float* mean; float* invstd; float* rmean; float* rvar; long long* nbt;
};
__global__ __launch_bounds__(256) void bn_finalize2_kernel(BnFin a, BnFin b, int C, double count, float eps, float momentum, int clear) {
    const BnFin& f = blockIdx.y ? b : a;
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0 && f.nbt) *f.nbt += 1;
    if (c >= C) return;
    float* p1 = f.stats + (size_t)lane * 2 * C + c;
    float* p2 = p1 + C;
    double s1 = (double)*p1, s2 = (double)*p2;
    if (clear) { *p1 = 0.f; *p2 = 0.f; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (lane != 0) return;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    f.mean[c] = (float)m;
    f.invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (f.rmean) f.rmean[c] = (1.f - momentum) * f.rmean[c] + momentum * (float)m;
    if (f.rvar) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        f.rvar[c] = (1.f - momentum) * f.rvar[c] + momentum * (float)unbiased;
    }
}

// eval mode: mean = running_mean, invstd = rsqrt(running_var + eps)
__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    invstd[c] = 1.0f / sqrtf(rvar[c] + eps);
}

// ------------------------------------------------------------------------------------------
// elementwise over (N, C, S) with per-channel constants
// ------------------------------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ void ncs_walk(long long total, int C, int S, F&& f) {
    // grid-stride over groups of 4 (S % 4 == 0) or single elements
    if ((S & 3) == 0) {
        const long long groups = total >> 2;
        for (long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x; gidx < groups;
             gidx += (long long)gridDim.x * blockDim.x) {
            const long long i = gidx << 2;
            const int c = (int)((i / S) % C);
            f((size_t)i, c, 4);
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
             i += (long long)gridDim.x * blockDim.x) {
            const int c = (int)((i / S) % C);
            f((size_t)i, c, 1);
        }
    }
}

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, long long total, int C, int S,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int act, float* __restrict__ y) {
    ncs_walk(total, C, S, [&](size_t i, int c, int cnt) {
        const float a = gamma[c] * invstd[c];
        const float b = beta[c] - mean[c] * a;
        if (cnt == 4) {
            float4 v = *reinterpret_cast<const float4*>(x + i);
            v.x = act_apply(v.x * a + b, act);
            v.y = act_apply(v.y * a + b, act);
            v.z = act_apply(v.z * a + b, act);
            v.w = act_apply(v.w * a + b, act);
            *reinterpret_cast<float4*>(y + i) = v;
        } else {
            y[i] = act_apply(x[i] * a + b, act);
        }
    });
}

__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ y, int N, int C, int S,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int act,
                                                                float* __restrict__ red) {
    const int c = blockIdx.y;
    const float mu = mean[c], is = invstd[c];
    float v[2] = {0.f, 0.f};   // dgamma, dbeta
    channel_walk(N, C, S, c, [&](size_t off, int cnt) {
        if (cnt == 4) {
            float4 g = *reinterpret_cast<const float4*>(dy + off);
            float4 xx = *reinterpret_cast<const float4*>(x + off);
            float4 yy = *reinterpret_cast<const float4*>(y + off);
            float d0 = g.x * act_grad_from_y(yy.x, act), d1 = g.y * act_grad_from_y(yy.y, act);
            float d2 = g.z * act_grad_from_y(yy.z, act), d3 = g.w * act_grad_from_y(yy.w, act);
            v[0] += d0 * (xx.x - mu) * is + d1 * (xx.y - mu) * is + d2 * (xx.z - mu) * is + d3 * (xx.w - mu) * is;
            v[1] += (d0 + d1) + (d2 + d3);
        } else {
            float d = dy[off] * act_grad_from_y(y[off], act);
            v[0] += d * (x[off] - mu) * is;
            v[1] += d;
        }
    });
    float* const dst[2] = {red + c, red + C + c};
    block_atomic<2>(v, dst);
}

__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ y, long long total, int C, int S,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, int act,
                                                               const float* __restrict__ red, float inv_count, int train,
                                                               float* __restrict__ dx) {
    ncs_walk(total, C, S, [&](size_t i, int c, int cnt) {
        const float mu = mean[c], is = invstd[c];
        const float a = gamma[c] * is;
        const float k1 = train ? red[C + c] * inv_count : 0.f;       // mean(dz)
        const float k2 = train ? red[c] * inv_count : 0.f;           // mean(dz * xhat)
        auto one = [&](float g, float xx, float yy) {
            float dz = g * act_grad_from_y(yy, act);
            return a * (dz - k1 - (xx - mu) * is * k2);
        };
        if (cnt == 4) {
            float4 g = *reinterpret_cast<const float4*>(dy + i);
            float4 xx = *reinterpret_cast<const float4*>(x + i);
            float4 yy = *reinterpret_cast<const float4*>(y + i);
            float4 o = make_float4(one(g.x, xx.x, yy.x), one(g.y, xx.y, yy.y), one(g.z, xx.z, yy.z), one(g.w, xx.w, yy.w));
            *reinterpret_cast<float4*>(dx + i) = o;
        } else {
            dx[i] = one(dy[i], x[i], y[i]);
        }
    });
}

// tanh and the logistic function of the gate on the hardware exponential and reciprocal (v_exp_f32 / v_rcp_f32, 1 ulp
// each): absolute error < 3e-7 over the whole range, a handful of instructions instead of the ~45 of libm's tanhf -- the
// one-pass channel backward was VALU-bound on it (3370 VALU instructions per wave, 40 us per launch ten times a step).
// Forward and backward use the same two functions.
__device__ __forceinline__ float gate_sig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float gate_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// ------------------------------------------------------------------------------------------
// gated activation  y = tanh(bn_f(yf)) * sigmoid(bn_g(yg)) * mask[n,c]
// ------------------------------------------------------------------------------------------
struct GateBN {
    const float *mean_f, *invstd_f, *gamma_f, *beta_f;
    const float *mean_g, *invstd_g, *gamma_g, *beta_g;
};

__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ yf, const float* __restrict__ yg,
                                                       long long total, int C, int S, GateBN bn,
                                                       const float* __restrict__ mask, float* __restrict__ y) {
    ncs_walk(total, C, S, [&](size_t i, int c, int cnt) {
        const float af = bn.gamma_f[c] * bn.invstd_f[c], bf = bn.beta_f[c] - bn.mean_f[c] * af;
        const float ag = bn.gamma_g[c] * bn.invstd_g[c], bg = bn.beta_g[c] - bn.mean_g[c] * ag;
        const float mk = mask ? mask[i / S] : 1.0f;
        auto one = [&](float f, float g) { return gate_tanh(f * af + bf) * gate_sig(g * ag + bg) * mk; };
        if (cnt == 4) {
            float4 f = *reinterpret_cast<const float4*>(yf + i);
            float4 g = *reinterpret_cast<const float4*>(yg + i);
            *reinterpret_cast<float4*>(y + i) = make_float4(one(f.x, g.x), one(f.y, g.y), one(f.z, g.z), one(f.w, g.w));
        } else {
            y[i] = one(yf[i], yg[i]);
        }
    });
}

__global__ __launch_bounds__(256) void gate_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ yf,
                                                              const float* __restrict__ yg, int N, int C, int S,
                                                              GateBN bn, const float* __restrict__ mask,
                                                              float* __restrict__ red) {
    const int c = blockIdx.y;
    const float muf = bn.mean_f[c], isf = bn.invstd_f[c], af = bn.gamma_f[c] * isf, bf = bn.beta_f[c] - muf * af;
    const float mug = bn.mean_g[c], isg = bn.invstd_g[c], ag = bn.gamma_g[c] * isg, bg = bn.beta_g[c] - mug * ag;
    float v[4] = {0.f, 0.f, 0.f, 0.f};   // dgamma_f, dbeta_f, dgamma_g, dbeta_g
    auto one = [&](size_t off) {
        const float mk = mask ? mask[off / S] : 1.0f;
        const float f = yf[off], g = yg[off];
        const float t = gate_tanh(f * af + bf);
        const float s = gate_sig(g * ag + bg);
        const float d = dy[off] * mk;
        const float dzf = d * s * (1.f - t * t);
        const float dzg = d * t * s * (1.f - s);
        v[0] += dzf * (f - muf) * isf;
        v[1] += dzf;
        v[2] += dzg * (g - mug) * isg;
        v[3] += dzg;
    };
    channel_walk(N, C, S, c, [&](size_t off, int cnt) {
        for (int k = 0; k < cnt; ++k) one(off + k);
    });
    float* const dst[4] = {red + c, red + C + c, red + 2 * C + c, red + 3 * C + c};
    block_atomic<4>(v, dst);
}

__global__ __launch_bounds__(256) void gate_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ yf,
                                                             const float* __restrict__ yg, long long total, int C, int S,
                                                             GateBN bn, const float* __restrict__ mask,
                                                             const float* __restrict__ red, float inv_count, int train,
                                                             float* __restrict__ dyf, float* __restrict__ dyg) {
    ncs_walk(total, C, S, [&](size_t i, int c, int cnt) {
        const float muf = bn.mean_f[c], isf = bn.invstd_f[c], af = bn.gamma_f[c] * isf, bf = bn.beta_f[c] - muf * af;
        const float mug = bn.mean_g[c], isg = bn.invstd_g[c], ag = bn.gamma_g[c] * isg, bg = bn.beta_g[c] - mug * ag;
        const float kf2 = train ? red[c] * inv_count : 0.f, kf1 = train ? red[C + c] * inv_count : 0.f;
        const float kg2 = train ? red[2 * C + c] * inv_count : 0.f, kg1 = train ? red[3 * C + c] * inv_count : 0.f;
        const float mk = mask ? mask[i / S] : 1.0f;
        for (int k = 0; k < cnt; ++k) {
            const size_t off = i + k;
            const float f = yf[off], g = yg[off];
            const float t = gate_tanh(f * af + bf);
            const float s = gate_sig(g * ag + bg);
            const float d = dy[off] * mk;
            const float dzf = d * s * (1.f - t * t);
            const float dzg = d * t * s * (1.f - s);
            dyf[off] = af * (dzf - kf1 - (f - muf) * isf * kf2);
            dyg[off] = ag * (dzg - kg1 - (g - mug) * isg * kg2);
        }
    });
}

// ---- row forms (S % 4 == 0): one WAVE per (n, c) row -- the channel's constants are scalar loads, no index
// division per element, every access 16 bytes.  (The element-walk forms above spend a 64-bit division and eight
// constant loads per 4 elements: 33-37 us for a 12.6 MB TCN tensor, these run in about half.)

__global__ __launch_bounds__(256) void gate_fwd_row_kernel(const float* __restrict__ yf, const float* __restrict__ yg,
                                                           int rows, int C, int S, GateBN bn,
                                                           const float* __restrict__ mask, float* __restrict__ y) {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int c = row % C;
        const float af = bn.gamma_f[c] * bn.invstd_f[c], bf = bn.beta_f[c] - bn.mean_f[c] * af;
        const float ag = bn.gamma_g[c] * bn.invstd_g[c], bg = bn.beta_g[c] - bn.mean_g[c] * ag;
        const float mk = mask ? mask[row] : 1.0f;
        const size_t base = (size_t)row * S;
        for (int s = lane * 4; s < S; s += 256) {
            const float4 f = *reinterpret_cast<const float4*>(yf + base + s);
            const float4 g = *reinterpret_cast<const float4*>(yg + base + s);
            float4 o;
            o.x = gate_tanh(f.x * af + bf) * gate_sig(g.x * ag + bg) * mk;
            o.y = gate_tanh(f.y * af + bf) * gate_sig(g.y * ag + bg) * mk;
            o.z = gate_tanh(f.z * af + bf) * gate_sig(g.z * ag + bg) * mk;
            o.w = gate_tanh(f.w * af + bf) * gate_sig(g.w * ag + bg) * mk;
            *reinterpret_cast<float4*>(y + base + s) = o;
        }
    }
}

__global__ __launch_bounds__(256) void gate_bwd_reduce_row_kernel(const float* __restrict__ dy, const float* __restrict__ yf,
                                                                  const float* __restrict__ yg, int rows, int C, int S,
                                                                  GateBN bn, const float* __restrict__ mask,
                                                                  float* __restrict__ red) {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int c = row % C;
        const float muf = bn.mean_f[c], isf = bn.invstd_f[c], af = bn.gamma_f[c] * isf, bf = bn.beta_f[c] - muf * af;
        const float mug = bn.mean_g[c], isg = bn.invstd_g[c], ag = bn.gamma_g[c] * isg, bg = bn.beta_g[c] - mug * ag;
        const float mk = mask ? mask[row] : 1.0f;
        const size_t base = (size_t)row * S;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;      // dgamma_f, dbeta_f, dgamma_g, dbeta_g
        for (int s = lane * 4; s < S; s += 256) {
            const float4 f4 = *reinterpret_cast<const float4*>(yf + base + s);
            const float4 g4 = *reinterpret_cast<const float4*>(yg + base + s);
            const float4 d4 = *reinterpret_cast<const float4*>(dy + base + s);
            const float ff[4] = {f4.x, f4.y, f4.z, f4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = gate_tanh(ff[e] * af + bf);
                const float sg = gate_sig(gg[e] * ag + bg);
                const float d = dd[e] * mk;
                const float dzf = d * sg * (1.f - t * t);
                const float dzg = d * t * sg * (1.f - sg);
                v0 += dzf * (ff[e] - muf) * isf;
                v1 += dzf;
                v2 += dzg * (gg[e] - mug) * isg;
                v3 += dzg;
            }
        }
        v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2); v3 = wave_sum(v3);
        if (lane == 0) {
            atomicAdd(red + c, v0);
            atomicAdd(red + C + c, v1);
            atomicAdd(red + 2 * C + c, v2);
            atomicAdd(red + 3 * C + c, v3);
        }
    }
}

__global__ __launch_bounds__(256) void gate_bwd_apply_row_kernel(const float* __restrict__ dy, const float* __restrict__ yf,
                                                                 const float* __restrict__ yg, int rows, int C, int S,
                                                                 GateBN bn, const float* __restrict__ mask,
                                                                 const float* __restrict__ red, float inv_count, int train,
                                                                 float* __restrict__ dyf, float* __restrict__ dyg) {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const int c = row % C;
        const float muf = bn.mean_f[c], isf = bn.invstd_f[c], af = bn.gamma_f[c] * isf, bf = bn.beta_f[c] - muf * af;
        const float mug = bn.mean_g[c], isg = bn.invstd_g[c], ag = bn.gamma_g[c] * isg, bg = bn.beta_g[c] - mug * ag;
        const float kf2 = train ? red[c] * inv_count : 0.f, kf1 = train ? red[C + c] * inv_count : 0.f;
        const float kg2 = train ? red[2 * C + c] * inv_count : 0.f, kg1 = train ? red[3 * C + c] * inv_count : 0.f;
        const float mk = mask ? mask[row] : 1.0f;
        const size_t base = (size_t)row * S;
        for (int s = lane * 4; s < S; s += 256) {
            const float4 f4 = *reinterpret_cast<const float4*>(yf + base + s);
            const float4 g4 = *reinterpret_cast<const float4*>(yg + base + s);
            const float4 d4 = *reinterpret_cast<const float4*>(dy + base + s);
            const float ff[4] = {f4.x, f4.y, f4.z, f4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
            float of[4], og[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = gate_tanh(ff[e] * af + bf);
                const float sg = gate_sig(gg[e] * ag + bg);
                const float d = dd[e] * mk;
                const float dzf = d * sg * (1.f - t * t);
                const float dzg = d * t * sg * (1.f - sg);
                of[e] = af * (dzf - kf1 - (ff[e] - muf) * isf * kf2);
                og[e] = ag * (dzg - kg1 - (gg[e] - mug) * isg * kg2);
            }
            *reinterpret_cast<float4*>(dyf + base + s) = make_float4(of[0], of[1], of[2], of[3]);
            *reinterpret_cast<float4*>(dyg + base + s) = make_float4(og[0], og[1], og[2], og[3]);
        }
    }
}

// ---- one-pass backward forms (training mode, S % 4 == 0, a channel's N*S values fit the workgroup's registers):
// ONE workgroup of 1024 threads owns a whole channel.  It reads dy and the saved tensors once, keeps dz and the
// normalised input in registers across the block-wide reduction, then writes the input gradient: the two-pass forms
// above read every operand twice (and evaluate tanh / exp twice) -- 150 + 25 MB against 75 + 25 MB for a TCN
// BatchNorm at B = 32, 150 + 50 against 75 + 50 for the gate.  dgamma / dbeta are ADDED to red (one writer per
// channel, so no atomics), which lets the caller hand in the flat-gradient slots whether or not they are still zero.
constexpr int CH_THREADS = 512;

template <int NV>
__device__ __forceinline__ void channel_block_sum(float (&v)[NV]) {
    __shared__ float part[NV][CH_THREADS / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float s = wave_sum(v[k]);
        if (lane == 0) part[k][wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < CH_THREADS / 64; ++w) t += part[k][w];
        v[k] = t;
    }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int PER>
__global__ __launch_bounds__(CH_THREADS) void bn_act_bwd_channel_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y, int N, int C, int S,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma, int act,
    float* __restrict__ red, const float* __restrict__ dy2, float* __restrict__ dx, float inv_count) {
    const int c = blockIdx.x;
    const int S4 = S >> 2, G = N * S4;
    const float mu = mean[c], is = invstd[c], a = gamma[c] * is;
    float4 dz[PER], xh[PER];
    float v[2] = {0.f, 0.f};      // dgamma, dbeta
    // all loads first, from clamped addresses (see gate_bwd_channel_kernel)
    float4 ds_[PER], xs_[PER], ys_[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int g = min((int)threadIdx.x + k * CH_THREADS, G - 1);
        const int n = g / S4, s4 = g - n * S4;
        const size_t o = ((size_t)n * C + c) * S + (size_t)s4 * 4;
        ds_[k] = ld4(dy + o); xs_[k] = ld4(x + o); ys_[k] = ld4(y + o);
        if (dy2) {                           // y had two consumers: their gradients are summed here, not by a separate kernel
            const float4 e = ld4(dy2 + o);
            ds_[k].x += e.x; ds_[k].y += e.y; ds_[k].z += e.z; ds_[k].w += e.w;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int g = (int)threadIdx.x + k * CH_THREADS;
        dz[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        xh[k] = dz[k];
        if (g < G) {
            const float4 d = ds_[k];
            const float4 xx = xs_[k], yy = ys_[k];
            dz[k] = make_float4(d.x * act_grad_from_y(yy.x, act), d.y * act_grad_from_y(yy.y, act),
                                d.z * act_grad_from_y(yy.z, act), d.w * act_grad_from_y(yy.w, act));
            xh[k] = make_float4((xx.x - mu) * is, (xx.y - mu) * is, (xx.z - mu) * is, (xx.w - mu) * is);
            v[0] += dz[k].x * xh[k].x + dz[k].y * xh[k].y + dz[k].z * xh[k].z + dz[k].w * xh[k].w;
            v[1] += (dz[k].x + dz[k].y) + (dz[k].z + dz[k].w);
        }
    }
    channel_block_sum<2>(v);
    if (threadIdx.x == 0) {
        red[c] += v[0];
        red[C + c] += v[1];
    }
    if (!dx) return;
    const float k1 = v[1] * inv_count, k2 = v[0] * inv_count;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int g = (int)threadIdx.x + k * CH_THREADS;
        if (g < G) {
            const int n = g / S4, s4 = g - n * S4;
            const size_t o = ((size_t)n * C + c) * S + (size_t)s4 * 4;
            float4 r = make_float4(a * (dz[k].x - k1 - xh[k].x * k2), a * (dz[k].y - k1 - xh[k].y * k2),
                                   a * (dz[k].z - k1 - xh[k].z * k2), a * (dz[k].w - k1 - xh[k].w * k2));
            *reinterpret_cast<float4*>(dx + o) = r;
        }
    }
}

template <int PER>
__global__ __launch_bounds__(CH_THREADS) void gate_bwd_channel_kernel(
    const float* __restrict__ dy, const float* __restrict__ yf, const float* __restrict__ yg, int N, int C, int S,
    GateBN bn, const float* __restrict__ mask, float* __restrict__ red, float* __restrict__ dyf,
    float* __restrict__ dyg, float inv_count) {
    const int c = blockIdx.x;
    const int S4 = S >> 2, G = N * S4;
    const float muf = bn.mean_f[c], isf = bn.invstd_f[c], af = bn.gamma_f[c] * isf, bf = bn.beta_f[c] - muf * af;
    const float mug = bn.mean_g[c], isg = bn.invstd_g[c], ag = bn.gamma_g[c] * isg, bg = bn.beta_g[c] - mug * ag;
    float dzf[PER][4], dzg[PER][4], fh[PER][4], gh[PER][4];
    float v[4] = {0.f, 0.f, 0.f, 0.f};      // dgamma_f, dbeta_f, dgamma_g, dbeta_g
    // every operand of the channel is requested before the first one is used (unconditional loads from clamped addresses:
    // with the loads inside `if (g < G)` the compiler kept each group's load -> wait -> tanh / exp sequence in order, eight
    // exposed memory latencies per workgroup)
    float4 d4s[PER], f4s[PER], g4s[PER];
    float mks[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int g = min((int)threadIdx.x + k * CH_THREADS, G - 1);
        const int n = g / S4, s4 = g - n * S4;
        const size_t o = ((size_t)n * C + c) * S + (size_t)s4 * 4;
        d4s[k] = ld4(dy + o); f4s[k] = ld4(yf + o); g4s[k] = ld4(yg + o);
        mks[k] = mask ? mask[n * C + c] : 1.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int g = (int)threadIdx.x + k * CH_THREADS;
#pragma unroll
        for (int e = 0; e < 4; ++e) dzf[k][e] = dzg[k][e] = fh[k][e] = gh[k][e] = 0.f;
        if (g < G) {
            const float4 d4 = d4s[k], f4 = f4s[k], g4 = g4s[k];
            const float mk = mks[k];
            const float dd[4] = {d4.x, d4.y, d4.z, d4.w}, ff[4] = {f4.x, f4.y, f4.z, f4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = gate_tanh(ff[e] * af + bf);
                const float sg = gate_sig(gg[e] * ag + bg);
                const float d = dd[e] * mk;
                dzf[k][e] = d * sg * (1.f - t * t);
                dzg[k][e] = d * t * sg * (1.f - sg);
                fh[k][e] = (ff[e] - muf) * isf;
                gh[k][e] = (gg[e] - mug) * isg;
                v[0] += dzf[k][e] * fh[k][e];
                v[1] += dzf[k][e];
                v[2] += dzg[k][e] * gh[k][e];
                v[3] += dzg[k][e];
            }
        }
    }
    channel_block_sum<4>(v);
    if (threadIdx.x == 0) {
        red[c] += v[0];
        red[C + c] += v[1];
        red[2 * C + c] += v[2];
        red[3 * C + c] += v[3];
    }
    const float kf2 = v[0] * inv_count, kf1 = v[1] * inv_count, kg2 = v[2] * inv_count, kg1 = v[3] * inv_count;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int g = (int)threadIdx.x + k * CH_THREADS;
        if (g < G) {
            const int n = g / S4, s4 = g - n * S4;
            const size_t o = ((size_t)n * C + c) * S + (size_t)s4 * 4;
            float of[4], og[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                of[e] = af * (dzf[k][e] - kf1 - fh[k][e] * kf2);
                og[e] = ag * (dzg[k][e] - kg1 - gh[k][e] * kg2);
            }
            *reinterpret_cast<float4*>(dyf + o) = make_float4(of[0], of[1], of[2], of[3]);
            *reinterpret_cast<float4*>(dyg + o) = make_float4(og[0], og[1], og[2], og[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// plain activations, add
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, long long n, int act,
                                                      float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = act_apply(x[i], act);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      long long n, int act, float* __restrict__ dx) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dx[i] = dy[i] * act_grad_from_y(y[i], act);
}
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  long long n, float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = a[i] + b[i];
}

// ------------------------------------------------------------------------------------------
// max pooling, window == stride, floor mode.  x (NC, H, W) -> y (NC, H/ph, W/pw)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, long long NC, int H, int W,
                                                          int ph, int pw, int OH, int OW, float* __restrict__ y,
                                                          uint8_t* __restrict__ idx) {
    const long long total = NC * OH * OW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ow = (int)(i % OW);
        const long long t = i / OW;
        const int oh = (int)(t % OH);
        const long long nc = t / OH;
        const float* base = x + ((size_t)nc * H + (size_t)oh * ph) * W + (size_t)ow * pw;
        float best = base[0];
        int bi = 0;
        for (int a = 0; a < ph; ++a)
            for (int b = 0; b < pw; ++b) {
                const float v = base[(size_t)a * W + b];
                if (v > best || v != v) {   // first maximum wins; NaN propagates (torch semantics)
                    best = v;
                    bi = a * pw + b;
                }
            }
        y[i] = best;
        if (idx) idx[i] = (uint8_t)bi;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          long long NC, int H, int W, int ph, int pw, int OH, int OW,
                                                          float* __restrict__ dx) {
    const long long total = NC * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const long long t = i / W;
        const int h = (int)(t % H);
        const long long nc = t / H;
        const int oh = h / ph, ow = w / pw;
        float v = 0.f;
        if (oh < OH && ow < OW) {
            const size_t o = ((size_t)nc * OH + oh) * OW + ow;
            const int local = (h - oh * ph) * pw + (w - ow * pw);
            if (idx[o] == local) v = dy[o];
        }
        dx[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// Philox-4x32-10 dropout
// ------------------------------------------------------------------------------------------
// philox4x32_10 / u01: common.h

// `state` (nullable): the device-resident step state of seld_step_begin; state[0] is added to `offset`, so that a
// launch recorded in a HIP graph draws fresh numbers at every replay.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, long long n, float p, float scale,
                                                      uint64_t seed, uint64_t offset, const uint64_t* __restrict__ state,
                                                      float* __restrict__ y) {
    if (state) offset += state[0];
    const long long groups = (n + 3) >> 2;
    for (long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (long long)gridDim.x * blockDim.x) {
        const uint4 r = philox4x32_10(offset + (uint64_t)gi, seed);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
        const long long i = gi << 2;
        if (i + 3 < n) {
            float4 v = *reinterpret_cast<const float4*>(x + i);
            v.x = u01(rr[0]) >= p ? v.x * scale : 0.f;
            v.y = u01(rr[1]) >= p ? v.y * scale : 0.f;
            v.z = u01(rr[2]) >= p ? v.z * scale : 0.f;
            v.w = u01(rr[3]) >= p ? v.w * scale : 0.f;
            *reinterpret_cast<float4*>(y + i) = v;
        } else {
            for (int k = 0; k < 4 && i + k < n; ++k) y[i + k] = u01(rr[k]) >= p ? x[i + k] * scale : 0.f;
        }
    }
}

// pooled = relu(a * raw + b) on the pooled-size tensor the pooling convolution kernel left (hcq_first_pool_kernel), and --
// the stage's Dropout (model.py:282) in the same pass -- out = dropout(pooled) with exactly the mask dropout_kernel would
// draw for the same (seed, offset): group gi = linear element index / 4.  One workgroup per (n, c) plane.
__global__ __launch_bounds__(256) void bn_pool_finish_kernel(const float* __restrict__ raw, int C, int S,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ pooled, float p, float scale, uint64_t seed,
                                                             uint64_t offset, const uint64_t* __restrict__ state,
                                                             float* __restrict__ out) {
    const int c = blockIdx.x % C;
    const float a = gamma[c] * invstd[c], b = beta[c] - mean[c] * a;
    const size_t base = (size_t)blockIdx.x * S;
    if (out && state) offset += state[0];
    for (int s = threadIdx.x * 4; s < S; s += 256 * 4) {
        const float4 v = *reinterpret_cast<const float4*>(raw + base + s);
        float4 o;
        o.x = fmaxf(v.x * a + b, 0.f); o.y = fmaxf(v.y * a + b, 0.f);
        o.z = fmaxf(v.z * a + b, 0.f); o.w = fmaxf(v.w * a + b, 0.f);
        if (v.x != v.x) o.x = v.x;                                   // NaN propagates (torch's relu / max_pool)
        if (v.y != v.y) o.y = v.y;
        if (v.z != v.z) o.z = v.z;
        if (v.w != v.w) o.w = v.w;
        if (pooled) *reinterpret_cast<float4*>(pooled + base + s) = o;
        if (out) {
            const uint4 r = philox4x32_10(offset + (uint64_t)((base + s) >> 2), seed);
            float4 d;
            d.x = u01(r.x) >= p ? o.x * scale : 0.f;
            d.y = u01(r.y) >= p ? o.y * scale : 0.f;
            d.z = u01(r.z) >= p ? o.z * scale : 0.f;
            d.w = u01(r.w) >= p ? o.w * scale : 0.f;
            *reinterpret_cast<float4*>(out + base + s) = d;
        }
    }
}

__global__ void dropout_mask_rows_kernel(long long rows, float p, float scale, uint64_t seed, uint64_t offset,
                                         const uint64_t* __restrict__ state, float* __restrict__ mask) {
    if (state) offset += state[0];
    const long long groups = (rows + 3) >> 2;
    for (long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (long long)gridDim.x * blockDim.x) {
        const uint4 r = philox4x32_10(offset + (uint64_t)gi, seed);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
        for (int k = 0; k < 4 && gi * 4 + k < rows; ++k) mask[gi * 4 + k] = u01(rr[k]) >= p ? scale : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// (N, C, T) <-> (N, T, C)
// ------------------------------------------------------------------------------------------
__global__ void transpose_kernel(const float* __restrict__ x, int R, int Cc, float* __restrict__ y) {
    // per batch item: x (R, Cc) -> y (Cc, R)
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * R * Cc;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        if (r < R && c < Cc) tile[j][threadIdx.x] = x[base + (size_t)r * Cc + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < R && c < Cc) y[base + (size_t)c * R + r] = tile[threadIdx.x][j];
    }
}

// ------------------------------------------------------------------------------------------
// loss = w_sed * mean BCE(sed, t_sed) + w_doa * mean MSE(doa, t_doa)     (train.py:186-204)
// ------------------------------------------------------------------------------------------
constexpr int LOSS_BLOCKS = 256;
__device__ float g_loss_part[LOSS_BLOCKS];
__device__ unsigned g_loss_ticket;          // zero at module load, handed back zero by every launch
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ sed, const float* __restrict__ doa,
                                                   const float* __restrict__ target, long long rows, int n_sed, int n_doa,
                                                   float w_sed, float w_doa, float* __restrict__ loss,
                                                   float* __restrict__ dsed, float* __restrict__ ddoa) {
    const int ncol = n_sed + n_doa;
    const long long total = rows * ncol;
    const float inv_sed = 1.0f / (float)(rows * n_sed);
    const float inv_doa = 1.0f / (float)(rows * n_doa);
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ncol;
        const int c = (int)(i - r * ncol);
        const float t = target[i];
        if (c < n_sed) {
            const size_t o = (size_t)r * n_sed + c;
            const float s = sed[o];
            // torch.nn.BCELoss clamps the logs at -100
            const float l1 = fmaxf(logf(s), -100.f), l0 = fmaxf(logf(1.f - s), -100.f);
            acc += -(t * l1 + (1.f - t) * l0) * inv_sed * w_sed;
            if (dsed) dsed[o] = w_sed * inv_sed * (s - t) / fmaxf(s * (1.f - s), 1e-12f);
        } else {
            const size_t o = (size_t)r * n_doa + (c - n_sed);
            const float d = doa[o] - t;
            acc += d * d * inv_doa * w_doa;
            if (ddoa) ddoa[o] = w_doa * inv_doa * 2.f * d;
        }
    }
    // workgroup sums -> g_loss_part; the workgroup that draws the last ticket adds them in a fixed order and writes the
    // loss: no zeroed accumulator (the host mirror used to launch a fill per step), the same bits for any grid
    __shared__ float red[4];
    __shared__ int last;
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        g_loss_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        last = atomicAdd(&g_loss_ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    __threadfence();
    float t = 0.f;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 64) t += *(volatile float*)&g_loss_part[b];
    t = wave_sum(t);
    if (threadIdx.x == 0) {
        *loss = t;
        g_loss_ticket = 0;          // ready for the next evaluation (launches of this kernel on one device must not overlap)
    }
}

// ------------------------------------------------------------------------------------------
// Adam over a flat buffer (torch.optim.Adam, no amsgrad, L2 weight decay added to the grad)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                   float b1, float b2, float eps, float wd, float step,
                                                   float gscale) {
    // bias corrections ON THE DEVICE, with the expressions of adam_state_kernel: an eager step and a replayed one then
    // produce the same bits (the host's powf rounds differently now and then)
    const float bc1 = 1.0f - powf(b1, step);
    const float bc2_sqrt = sqrtf(1.0f - powf(b2, step));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float grad = g[i] * gscale;
        const float pv = p[i];
        if (wd != 0.f) grad += wd * pv;
        const float mi = b1 * m[i] + (1.f - b1) * grad;
        const float vi = b2 * v[i] + (1.f - b2) * grad * grad;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pv - (lr / bc1) * (mi / denom);
    }
}

// The same update with the step number and the learning rate read from the device-resident step state
// (state[1] = 1-based step, low 32 bits of state[2] = lr as float bits): a launch recorded in a HIP graph then follows
// the optimiser's bias correction and the scheduler's learning rate from replay to replay.
__global__ __launch_bounds__(256) void adam_state_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, long long n,
                                                         float b1, float b2, float eps, float wd, float gscale,
                                                         uint64_t* __restrict__ state) {
    // end of the step: the Philox base moves past this step's draws (nothing after Adam draws), so whatever is issued
    // next -- a replay or an eager launch with host offsets from 0 -- sees fresh counters
    if (blockIdx.x == 0 && threadIdx.x == 0) state[0] += state[3];
    const float step = (float)state[1];
    const float lr = __uint_as_float((unsigned)state[2]);
    const float bc1 = 1.0f - powf(b1, step);
    const float bc2_sqrt = sqrtf(1.0f - powf(b2, step));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float grad = g[i] * gscale;
        const float pv = p[i];
        if (wd != 0.f) grad += wd * pv;
        const float mi = b1 * m[i] + (1.f - b1) * grad;
        const float vi = b2 * v[i] + (1.f - b2) * grad * grad;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pv - (lr / bc1) * (mi / denom);
    }
}

// Start of a training step: zero the flat gradient buffer (16-byte stores) and advance the optimiser step
//   state[1] += 1
// (the Philox base state[0] advances by state[3] = draws per step at the END of the step, in adam_state_kernel)
__global__ __launch_bounds__(256) void step_begin_kernel(float* __restrict__ g, long long n, uint64_t* __restrict__ state) {
    if (state && blockIdx.x == 0 && threadIdx.x == 0) state[1] += 1;
    const long long n4 = n >> 2;
    float4* g4 = reinterpret_cast<float4*>(g);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) g[(n4 << 2) + threadIdx.x] = 0.f;
}

// one wave per row, 4 rows per block
static inline unsigned row_grid(long long rows) {
    long long b = (rows + 3) / 4;
    return (unsigned)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

static inline unsigned grid_for(long long work_items, int per_block = 256, int cap = 8192) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace seld

using namespace seld;
#define ST(s) ((hipStream_t)(s))
// chunks of a per-channel reduction: one (the block's single atomic add per output is then the only contribution) when
// SELD_DETERMINISTIC is set
static inline unsigned red_chunks(long long M) { return env().deterministic ? 1u : (unsigned)((M + RED_CHUNK - 1) / RED_CHUNK); }

extern "C" int seld_channel_stats(const float* x, int32_t N, int32_t C, int32_t S, float* stats, void* stream) {
    if (!x || !stats || N <= 0 || C <= 0 || S <= 0) return SELD_EINVAL;
    const long long M = (long long)N * S;
    dim3 grid(red_chunks(M), C);
    hipLaunchKernelGGL(channel_stats_kernel, grid, dim3(256), 0, ST(stream), x, N, C, S, stats);
    return check_launch();
}

extern "C" int seld_bn_finalize_ex(float* stats, int32_t C, int64_t count, float eps, float momentum, float* mean,
                                   float* invstd, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, int32_t clear_stats, void* stream) {
    if (!stats || !mean || !invstd || C <= 0 || count <= 0) return SELD_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, ST(stream), stats, C, (double)count, eps,
                       momentum, mean, invstd, running_mean, running_var, (long long*)num_batches_tracked, clear_stats);
    return check_launch();
}

extern "C" int seld_bn_finalize2_ex(float* statsA, float* statsB, int32_t C, int64_t count, float eps, float momentum,
                                    float* meanA, float* invstdA, float* running_meanA, float* running_varA,
                                    int64_t* nbtA, float* meanB, float* invstdB, float* running_meanB,
                                    float* running_varB, int64_t* nbtB, int32_t clear_stats, void* stream) {
    if (!statsA || !statsB || !meanA || !invstdA || !meanB || !invstdB || C <= 0 || count <= 0) return SELD_EINVAL;
    const BnFin a{statsA, meanA, invstdA, running_meanA, running_varA, (long long*)nbtA};
    const BnFin b{statsB, meanB, invstdB, running_meanB, running_varB, (long long*)nbtB};
    hipLaunchKernelGGL(bn_finalize2_kernel, dim3((C + 3) / 4, 2), dim3(256), 0, ST(stream), a, b, C, (double)count, eps,
                       momentum, clear_stats);
    return check_launch();
}

extern "C" int seld_bn_finalize(const float* stats, int32_t C, int64_t count, float eps, float momentum, float* mean,
                                float* invstd, float* running_mean, float* running_var, void* stream) {
    return seld_bn_finalize_ex(const_cast<float*>(stats), C, count, eps, momentum, mean, invstd, running_mean,
                               running_var, nullptr, 0, stream);
}

extern "C" int seld_bn_eval_stats(const float* running_mean, const float* running_var, int32_t C, float eps,
                                  float* mean, float* invstd, void* stream) {
    if (!running_mean || !running_var || !mean || !invstd || C <= 0) return SELD_EINVAL;
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, ST(stream), running_mean, running_var,
                       C, eps, mean, invstd);
    return check_launch();
}

extern "C" int seld_bn_act_fwd(const float* x, int32_t N, int32_t C, int32_t S, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, int32_t act, float* y, void* stream) {
    if (!x || !y || !mean || !invstd || !gamma || !beta) return SELD_EINVAL;
    const long long total = (long long)N * C * S;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid_for(total / 4 + 1)), dim3(256), 0, ST(stream), x, total, C, S, mean,
                       invstd, gamma, beta, act, y);
    return check_launch();
}

extern "C" int seld_bn_act_bwd_reduce(const float* dy, const float* x, const float* y, int32_t N, int32_t C, int32_t S,
                                      const float* mean, const float* invstd, const float* gamma, const float* beta,
                                      int32_t act, float* red, void* stream) {
    (void)gamma; (void)beta;
    if (!dy || !x || !y || !red) return SELD_EINVAL;
    const long long M = (long long)N * S;
    dim3 grid(red_chunks(M), C);
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, grid, dim3(256), 0, ST(stream), dy, x, y, N, C, S, mean, invstd, act, red);
    return check_launch();
}

extern "C" int seld_bn_act_bwd_apply(const float* dy, const float* x, const float* y, int32_t N, int32_t C, int32_t S,
                                     const float* mean, const float* invstd, const float* gamma, const float* beta,
                                     int32_t act, const float* red, int32_t train, float* dx, void* stream) {
    (void)beta;
    if (!dy || !x || !y || !dx || (train && !red)) return SELD_EINVAL;
    const long long total = (long long)N * C * S;
    const float inv_count = 1.0f / (float)((long long)N * S);
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(grid_for(total / 4 + 1)), dim3(256), 0, ST(stream), dy, x, y, total,
                       C, S, mean, invstd, gamma, act, red, inv_count, train, dx);
    return check_launch();
}

// One-pass training-mode backward (see bn_act_bwd_channel_kernel).  SELD_EUNSUPPORTED when the channel does not fit
// the workgroup's registers or S is not a multiple of 4: the caller then takes the reduce + apply pair.
extern "C" int seld_bn_act_bwd_fused(const float* dy, const float* x, const float* y, int32_t N, int32_t C, int32_t S,
                                     const float* mean, const float* invstd, const float* gamma, int32_t act,
                                     float* red, const float* dy2, float* dx, void* stream) {
    if (!dy || !x || !y || !red || !mean || !invstd || !gamma || N <= 0 || C <= 0 || S <= 0) return SELD_EINVAL;
    const long long G = (long long)N * S / 4;
    if ((S & 3) || G > 16LL * CH_THREADS) return SELD_EUNSUPPORTED;
    const float inv_count = 1.0f / (float)((long long)N * S);
    const int per = (int)((G + CH_THREADS - 1) / CH_THREADS);
#define SELD_BN_CH(P)                                                                                                   \
    hipLaunchKernelGGL(bn_act_bwd_channel_kernel<P>, dim3(C), dim3(CH_THREADS), 0, ST(stream), dy, x, y, N, C, S, mean,  \
                       invstd, gamma, act, red, dy2, dx, inv_count)
    if (per <= 1) SELD_BN_CH(1);
    else if (per <= 2) SELD_BN_CH(2);
    else if (per <= 4) SELD_BN_CH(4);
    else if (per <= 8) SELD_BN_CH(8);
    else SELD_BN_CH(16);
#undef SELD_BN_CH
    return check_launch();
}

static GateBN mk_gate(const float* mean_f, const float* invstd_f, const float* gamma_f, const float* beta_f,
                      const float* mean_g, const float* invstd_g, const float* gamma_g, const float* beta_g) {
    GateBN b{mean_f, invstd_f, gamma_f, beta_f, mean_g, invstd_g, gamma_g, beta_g};
    return b;
}

extern "C" int seld_gate_fwd(const float* yf, const float* yg, int32_t N, int32_t C, int32_t S, const float* mean_f,
                             const float* invstd_f, const float* gamma_f, const float* beta_f, const float* mean_g,
                             const float* invstd_g, const float* gamma_g, const float* beta_g, const float* mask,
                             float* y, void* stream) {
    if (!yf || !yg || !y) return SELD_EINVAL;
    const long long total = (long long)N * C * S;
    if (S % 4 == 0 && (long long)N * C < (1LL << 31))
        hipLaunchKernelGGL(gate_fwd_row_kernel, dim3(row_grid((long long)N * C)), dim3(256), 0, ST(stream), yf, yg, N * C, C, S,
                           mk_gate(mean_f, invstd_f, gamma_f, beta_f, mean_g, invstd_g, gamma_g, beta_g), mask, y);
    else
        hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(total / 4 + 1)), dim3(256), 0, ST(stream), yf, yg, total, C, S,
                           mk_gate(mean_f, invstd_f, gamma_f, beta_f, mean_g, invstd_g, gamma_g, beta_g), mask, y);
    return check_launch();
}

extern "C" int seld_gate_bwd_reduce(const float* dy, const float* yf, const float* yg, int32_t N, int32_t C, int32_t S,
                                    const float* mean_f, const float* invstd_f, const float* gamma_f, const float* beta_f,
                                    const float* mean_g, const float* invstd_g, const float* gamma_g, const float* beta_g,
                                    const float* mask, float* red, void* stream) {
    if (!dy || !yf || !yg || !red) return SELD_EINVAL;
    const long long M = (long long)N * S;
    dim3 grid(red_chunks(M), C);
    if (S % 4 == 0 && (long long)N * C < (1LL << 31) && !env().deterministic)
        hipLaunchKernelGGL(gate_bwd_reduce_row_kernel, dim3(row_grid((long long)N * C)), dim3(256), 0, ST(stream), dy, yf, yg,
                           N * C, C, S, mk_gate(mean_f, invstd_f, gamma_f, beta_f, mean_g, invstd_g, gamma_g, beta_g), mask, red);
    else
        hipLaunchKernelGGL(gate_bwd_reduce_kernel, grid, dim3(256), 0, ST(stream), dy, yf, yg, N, C, S,
                           mk_gate(mean_f, invstd_f, gamma_f, beta_f, mean_g, invstd_g, gamma_g, beta_g), mask, red);
    return check_launch();
}

extern "C" int seld_gate_bwd_apply(const float* dy, const float* yf, const float* yg, int32_t N, int32_t C, int32_t S,
                                   const float* mean_f, const float* invstd_f, const float* gamma_f, const float* beta_f,
                                   const float* mean_g, const float* invstd_g, const float* gamma_g, const float* beta_g,
                                   const float* mask, const float* red, int32_t train, float* dyf, float* dyg,
                                   void* stream) {
    if (!dy || !yf || !yg || !dyf || !dyg || (train && !red)) return SELD_EINVAL;
    const long long total = (long long)N * C * S;
    const float inv_count = 1.0f / (float)((long long)N * S);
    if (S % 4 == 0 && (long long)N * C < (1LL << 31))
        hipLaunchKernelGGL(gate_bwd_apply_row_kernel, dim3(row_grid((long long)N * C)), dim3(256), 0, ST(stream), dy, yf, yg,
                           N * C, C, S, mk_gate(mean_f, invstd_f, gamma_f, beta_f, mean_g, invstd_g, gamma_g, beta_g), mask, red,
                           inv_count, train, dyf, dyg);
    else
        hipLaunchKernelGGL(gate_bwd_apply_kernel, dim3(grid_for(total / 4 + 1)), dim3(256), 0, ST(stream), dy, yf, yg, total,
                           C, S, mk_gate(mean_f, invstd_f, gamma_f, beta_f, mean_g, invstd_g, gamma_g, beta_g), mask, red,
                           inv_count, train, dyf, dyg);
    return check_launch();
}

extern "C" int seld_gate_bwd_fused(const float* dy, const float* yf, const float* yg, int32_t N, int32_t C, int32_t S,
                                   const float* mean_f, const float* invstd_f, const float* gamma_f, const float* beta_f,
                                   const float* mean_g, const float* invstd_g, const float* gamma_g, const float* beta_g,
                                   const float* mask, float* red, float* dyf, float* dyg, void* stream) {
    if (!dy || !yf || !yg || !red || !dyf || !dyg || N <= 0 || C <= 0 || S <= 0) return SELD_EINVAL;
    const long long G = (long long)N * S / 4;
    if ((S & 3) || G > 8LL * CH_THREADS) return SELD_EUNSUPPORTED;
    const float inv_count = 1.0f / (float)((long long)N * S);
    const int per = (int)((G + CH_THREADS - 1) / CH_THREADS);
    const GateBN bn = mk_gate(mean_f, invstd_f, gamma_f, beta_f, mean_g, invstd_g, gamma_g, beta_g);
#define SELD_GATE_CH(P)                                                                                                 \
    hipLaunchKernelGGL(gate_bwd_channel_kernel<P>, dim3(C), dim3(CH_THREADS), 0, ST(stream), dy, yf, yg, N, C, S, bn,    \
                       mask, red, dyf, dyg, inv_count)
    if (per <= 1) SELD_GATE_CH(1);
    else if (per <= 2) SELD_GATE_CH(2);
    else if (per <= 4) SELD_GATE_CH(4);
    else SELD_GATE_CH(8);
#undef SELD_GATE_CH
    return check_launch();
}

extern "C" int seld_act_fwd(const float* x, int64_t n, int32_t act, float* y, void* stream) {
    if (!x || !y || n < 0) return SELD_EINVAL;
    if (n == 0) return SELD_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, (long long)n, act, y);
    return check_launch();
}
extern "C" int seld_act_bwd(const float* dy, const float* y, int64_t n, int32_t act, float* dx, void* stream) {
    if (!dy || !y || !dx || n < 0) return SELD_EINVAL;
    if (n == 0) return SELD_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), dy, y, (long long)n, act, dx);
    return check_launch();
}
extern "C" int seld_accumulate(float* dst, const float* src, int64_t n, void* stream) {
    if (!dst || !src || n < 0) return SELD_EINVAL;
    if (n == 0) return SELD_OK;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), (const float*)dst, src, (long long)n, dst);
    return check_launch();
}

extern "C" int seld_add(const float* a, const float* b, int64_t n, float* y, void* stream) {
    if (!a || !b || !y || n < 0) return SELD_EINVAL;
    if (n == 0) return SELD_OK;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), a, b, (long long)n, y);
    return check_launch();
}

extern "C" int seld_maxpool_fwd(const float* x, int64_t NC, int32_t H, int32_t W, int32_t ph, int32_t pw, float* y,
                                uint8_t* idx, void* stream) {
    if (!x || !y || ph <= 0 || pw <= 0 || ph * pw > 255 || H < ph || W < pw) return SELD_EINVAL;
    const int OH = H / ph, OW = W / pw;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(NC * OH * OW)), dim3(256), 0, ST(stream), x, (long long)NC, H, W,
                       ph, pw, OH, OW, y, idx);
    return check_launch();
}
extern "C" int seld_maxpool_bwd(const float* dy, const uint8_t* idx, int64_t NC, int32_t H, int32_t W, int32_t ph,
                                int32_t pw, float* dx, void* stream) {
    if (!dy || !idx || !dx || ph <= 0 || pw <= 0 || H < ph || W < pw) return SELD_EINVAL;
    const int OH = H / ph, OW = W / pw;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(NC * H * W)), dim3(256), 0, ST(stream), dy, idx, (long long)NC, H,
                       W, ph, pw, OH, OW, dx);
    return check_launch();
}

extern "C" int seld_dropout_fwd(const float* x, int64_t n, float p, uint64_t seed, uint64_t offset,
                                const uint64_t* state, float* y, void* stream) {
    if (!x || !y || n < 0 || p < 0.f || p >= 1.f) return SELD_EINVAL;
    if (n == 0) return SELD_OK;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, ST(stream), x, (long long)n, p,
                       1.0f / (1.0f - p), seed, offset, state, y);
    return check_launch();
}
extern "C" int seld_bn_pool_finish(const float* raw, int32_t N, int32_t C, int32_t S, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, float* pooled, float p, uint64_t seed,
                                   uint64_t offset, const uint64_t* state, float* out, void* stream) {
    if (!raw || !mean || !invstd || !gamma || !beta || (!pooled && !out) || N <= 0 || C <= 0 || S <= 0 || (S & 3)) return SELD_EINVAL;
    if (out && (p < 0.f || p >= 1.f)) return SELD_EINVAL;
    hipLaunchKernelGGL(bn_pool_finish_kernel, dim3((unsigned)(N * C)), dim3(256), 0, ST(stream), raw, C, S, mean, invstd,
                       gamma, beta, pooled, p, out ? 1.0f / (1.0f - p) : 1.0f, seed, offset, state, out);
    return check_launch();
}

extern "C" int seld_dropout_mask_rows(int64_t rows, float p, uint64_t seed, uint64_t offset, const uint64_t* state,
                                      float* mask, void* stream) {
    if (!mask || rows <= 0 || p < 0.f || p >= 1.f) return SELD_EINVAL;
    hipLaunchKernelGGL(dropout_mask_rows_kernel, dim3(grid_for((rows + 3) / 4)), dim3(256), 0, ST(stream), (long long)rows,
                       p, 1.0f / (1.0f - p), seed, offset, state, mask);
    return check_launch();
}

extern "C" int seld_transpose_nct_ntc(const float* x, int32_t N, int32_t C, int32_t T, float* y, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || T <= 0) return SELD_EINVAL;
    dim3 grid((T + 31) / 32, (C + 31) / 32, N);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, ST(stream), x, C, T, y);
    return check_launch();
}
extern "C" int seld_transpose_ntc_nct(const float* x, int32_t N, int32_t T, int32_t C, float* y, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || T <= 0) return SELD_EINVAL;
    dim3 grid((C + 31) / 32, (T + 31) / 32, N);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, ST(stream), x, T, C, y);
    return check_launch();
}

extern "C" int seld_loss_fwd_bwd(const float* sed, const float* doa, const float* target, int64_t rows, int32_t n_sed,
                                 int32_t n_doa, float w_sed, float w_doa, float* loss, float* dsed, float* ddoa,
                                 void* stream) {
    if (!sed || !doa || !target || !loss || rows <= 0 || n_sed <= 0 || n_doa <= 0) return SELD_EINVAL;
    hipLaunchKernelGGL(loss_kernel, dim3(grid_for(rows * (n_sed + n_doa), 256, LOSS_BLOCKS)), dim3(256), 0, ST(stream), sed, doa,
                       target, (long long)rows, n_sed, n_doa, w_sed, w_doa, loss, dsed, ddoa);
    return check_launch();
}

extern "C" int seld_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                              void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return SELD_EINVAL;
    if (n == 0) return SELD_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), param, grad, exp_avg, exp_avg_sq,
                       (long long)n, lr, beta1, beta2, eps, weight_decay, (float)step, grad_scale);
    return check_launch();
}
extern "C" int seld_adam_flat_state(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                    uint64_t* state, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !state || n < 0) return SELD_EINVAL;
    if (n == 0) return SELD_OK;
    hipLaunchKernelGGL(adam_state_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), param, grad, exp_avg, exp_avg_sq,
                       (long long)n, beta1, beta2, eps, weight_decay, grad_scale, state);
    return check_launch();
}
extern "C" int seld_step_begin(float* flat_grad, int64_t n, uint64_t* state, void* stream) {
    if ((n > 0 && !flat_grad) || n < 0) return SELD_EINVAL;
    if (n > 0 && ((uintptr_t)flat_grad & 15)) return SELD_EINVAL;
    hipLaunchKernelGGL(step_begin_kernel, dim3(grid_for((n + 3) / 4, 256, 2048)), dim3(256), 0, ST(stream), flat_grad,
                       (long long)n, state);
    return check_launch();
}
