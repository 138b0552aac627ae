// Fused CNN-stage tail for gfx950:  BatchNorm2d -> ReLU -> MaxPool2d(ph, pw)   (model.py:278-281)
//
// The reference runs these as three full-resolution passes forward and three backward; on the first
// stage the map is 1.6 GB at batch 32, so they dominate the HBM traffic of a training step.  Here:
//   forward : ONE pass reads the conv output y and writes only the pooled map (1/ph/pw of the size)
//             plus a uint8 argmax per pooled element;
//   backward: the per-channel reductions need only pooled-size tensors, because at an arg-max with
//             z = relu(gamma * xhat + beta) > 0 the normalised input is xhat = (z - beta) / gamma and
//             every other position of the window has dz = 0; ONE full-resolution pass then writes the
//             gradient w.r.t. the conv output.
// The BatchNorm batch statistics themselves come from the convolution's epilogue (SELD_EPI_STATS).
#include "common.h"
#include "env.h"

namespace seld {

struct PoolGeom {
    long long NC;
    int C, H, W, ph, pw, OH, OW;
};

__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float* __restrict__ y, PoolGeom g,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               float* __restrict__ pooled, uint8_t* __restrict__ idx) {
    // grid: (ceil(OW / 256), OH, N*C): no index division on the hot path, channel constants are wave-uniform
    {
        const int ow = blockIdx.x * blockDim.x + threadIdx.x;
        if (ow >= g.OW) return;
        const int oh = blockIdx.y;
        const long long nc = blockIdx.z;
        const int c = (int)(nc % g.C);
        const long long i = (nc * g.OH + oh) * g.OW + ow;
        const float a = gamma[c] * invstd[c];
        const float b = beta[c] - mean[c] * a;
        const float* base = y + ((size_t)nc * g.H + (size_t)oh * g.ph) * g.W + (size_t)ow * g.pw;
        float best = 0.f;
        int bi = 0;
        for (int r = 0; r < g.ph; ++r)
            for (int s = 0; s < g.pw; ++s) {
                float z = base[(size_t)r * g.W + s] * a + b;
                z = z > 0.f ? z : 0.f;
                if ((r | s) == 0 || z > best || z != z) { best = z; bi = r * g.pw + s; }
            }
        pooled[i] = best;
        idx[i] = (uint8_t)bi;
    }
}

// Same for pooling along H only (pw == 1, the SELD CNN stages) and W a multiple of 4: a thread owns 4 adjacent
// columns, every access is 16 bytes.
// PH: window height known at compile time (0 = run-time g.ph): all PH row loads are issued before the first compare.
template <int PH>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_v4_kernel(const float* __restrict__ y, PoolGeom g,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  float* __restrict__ pooled, uint8_t* __restrict__ idx,
                                                                  DropP dr, float* __restrict__ dropped) {
    const int ow = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (ow >= g.OW) return;
    const int oh = blockIdx.y;
    const long long nc = blockIdx.z;
    const int c = (int)(nc % g.C);
    const long long i = (nc * g.OH + oh) * g.OW + ow;
    if (dr.p > 0.f && dr.state) dr.offset += dr.state[0];
    const float a = gamma[c] * invstd[c];
    const float b = beta[c] - mean[c] * a;
    const float* base = y + ((size_t)nc * g.H + (size_t)oh * g.ph) * g.W + ow;
    float best[4] = {0.f, 0.f, 0.f, 0.f};
    int bi[4] = {0, 0, 0, 0};
    auto take = [&](const float4 v, int r) __attribute__((always_inline)) {
        const float zz[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float z = zz[e] * a + b;
            z = z > 0.f ? z : 0.f;
            if (r == 0 || z > best[e] || z != z) { best[e] = z; bi[e] = r; }
        }
    };
    if constexpr (PH > 0) {
        float4 v[PH];
#pragma unroll
        for (int r = 0; r < PH; ++r) v[r] = *reinterpret_cast<const float4*>(base + (size_t)r * g.W);
#pragma unroll
        for (int r = 0; r < PH; ++r) take(v[r], r);
    } else {
        for (int r = 0; r < g.ph; ++r) take(*reinterpret_cast<const float4*>(base + (size_t)r * g.W), r);
    }
    *reinterpret_cast<float4*>(pooled + i) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<uchar4*>(idx + i) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
    if (dr.p > 0.f) {            // the stage's Dropout in the same pass: the mask seld_dropout_fwd draws for element group i / 4
        const float4 mk = dropout_mask4(dr.offset + (uint64_t)(i >> 2), dr.seed, dr.p, dr.scale);
        *reinterpret_cast<float4*>(dropped + i) = make_float4(best[0] * mk.x, best[1] * mk.y, best[2] * mk.z, best[3] * mk.w);
    }
}

// red[c] += sum dz * xhat, red[C + c] += sum dz   over the pooled elements of channel c
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_reduce_kernel(const float* __restrict__ dpooled,
                                                                      const float* __restrict__ pooled, int N, int C,
                                                                      int S /* OH*OW */, const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta,
                                                                      const float* __restrict__ y,
                                                                      const uint8_t* __restrict__ idx, PoolGeom g,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd,
                                                                      float* __restrict__ red, DropP dr) {
    const int c = blockIdx.y;
    if (dr.p > 0.f && dr.state) dr.offset += dr.state[0];
    const bool degenerate = gamma[c] == 0.f;      // xhat cannot be recovered from z: gather it from y instead
    const float inv_g = degenerate ? 0.f : 1.0f / gamma[c];
    const float be = beta[c];
    const long long M = (long long)N * S;
    const long long chunk = gridDim.x == 1 ? M : 8192;          // SELD_DETERMINISTIC: one workgroup per channel
    const long long beg = (long long)blockIdx.x * chunk;
    long long end = beg + chunk;
    if (end > M) end = M;
    float v0 = 0.f, v1 = 0.f;
    // One 64-bit division per workgroup, 32-bit arithmetic per element; a thread owns 4 consecutive elements (S % 4 == 0:
    // they lie in one sample) and both 16-byte loads are issued before either is used.
    const long long n0 = beg / S;
    const unsigned r_beg = (unsigned)(beg - n0 * S);
    const unsigned count = (unsigned)(end - beg);
    auto one = [&](float z, float d, long long n, unsigned s_, size_t off) __attribute__((always_inline)) {
        if (z > 0.f) {
            float xh = (z - be) * inv_g;
            if (degenerate) {
                const int oh = s_ / g.OW, ow = s_ - oh * g.OW;
                const int am = idx[off];
                const int r = am / g.pw, q = am - r * g.pw;
                xh = (y[(((size_t)n * C + c) * g.H + (size_t)oh * g.ph + r) * g.W + (size_t)ow * g.pw + q] - mean[c]) * invstd[c];
            }
            v0 += d * xh;
            v1 += d;
        }
    };
    if ((S & 3) == 0) {
        for (unsigned k = threadIdx.x * 4; k < count; k += blockDim.x * 4) {
            const unsigned local = r_beg + k;
            const unsigned dn = local / (unsigned)S;
            const unsigned s_ = local - dn * (unsigned)S;
            const long long n = n0 + dn;
            const size_t off = ((size_t)n * C + c) * S + s_;
            const float4 z4 = *reinterpret_cast<const float4*>(pooled + off);
            float4 d4 = *reinterpret_cast<const float4*>(dpooled + off);
            if (dr.p > 0.f) {                     // dpooled is the gradient behind the stage's Dropout: the same mask
                const float4 mk = dropout_mask4(dr.offset + (uint64_t)(off >> 2), dr.seed, dr.p, dr.scale);
                d4.x *= mk.x; d4.y *= mk.y; d4.z *= mk.z; d4.w *= mk.w;
            }
            one(z4.x, d4.x, n, s_, off);
            one(z4.y, d4.y, n, s_ + 1, off + 1);
            one(z4.z, d4.z, n, s_ + 2, off + 2);
            one(z4.w, d4.w, n, s_ + 3, off + 3);
        }
    } else {
        for (unsigned k = threadIdx.x; k < count; k += blockDim.x) {
            const unsigned local = r_beg + k;
            const unsigned dn = local / (unsigned)S;
            const unsigned s_ = local - dn * (unsigned)S;
            const long long n = n0 + dn;
            const size_t off = ((size_t)n * C + c) * S + s_;
            float d = dpooled[off];
            if (dr.p > 0.f) {
                const float4 mk = dropout_mask4(dr.offset + (uint64_t)(off >> 2), dr.seed, dr.p, dr.scale);
                const float m4[4] = {mk.x, mk.y, mk.z, mk.w};
                d *= m4[off & 3];
            }
            one(pooled[off], d, n, s_, off);
        }
    }
    __shared__ float r0[4], r1[4];
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = v0; r1[threadIdx.x >> 6] = v1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(red + c, r0[0] + r0[1] + r0[2] + r0[3]);
        atomicAdd(red + C + c, r1[0] + r1[1] + r1[2] + r1[3]);
    }
}

// dy[n,c,h,w] = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat)),  dz = dpooled at the arg-max with z > 0
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_apply_kernel(const float* __restrict__ dpooled,
                                                                     const float* __restrict__ pooled,
                                                                     const uint8_t* __restrict__ idx,
                                                                     const float* __restrict__ y, PoolGeom g,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ red, float inv_count,
                                                                     int train, float* __restrict__ dy) {
    // one thread per (nc, oh', ow') where oh' also covers the rows that floor-mode pooling drops
    const int OHx = (g.H + g.ph - 1) / g.ph, OWx = (g.W + g.pw - 1) / g.pw;
    {
        const int ow = blockIdx.x * blockDim.x + threadIdx.x;
        if (ow >= OWx) return;
        const int oh = blockIdx.y;
        const long long nc = blockIdx.z;
        const int c = (int)(nc % g.C);
        const float mu = mean[c], is = invstd[c];
        const float a = gamma[c] * is;
        const float k1 = train ? red[g.C + c] * inv_count : 0.f;
        const float k2 = train ? red[c] * inv_count : 0.f;
        float dz = 0.f;
        int am = -1;
        if (oh < g.OH && ow < g.OW) {
            const size_t o = ((size_t)nc * g.OH + oh) * g.OW + ow;
            if (pooled[o] > 0.f) { dz = dpooled[o]; am = idx[o]; }
        }
        for (int r = 0; r < g.ph; ++r) {
            const int h = oh * g.ph + r;
            if (h >= g.H) break;
            for (int s = 0; s < g.pw; ++s) {
                const int w = ow * g.pw + s;
                if (w >= g.W) break;
                const size_t off = ((size_t)nc * g.H + h) * g.W + w;
                const float xh = (y[off] - mu) * is;
                const float d = (r * g.pw + s == am) ? dz : 0.f;
                dy[off] = a * (d - k1 - xh * k2);
            }
        }
    }
}

// pw == 1, W % 4 == 0, H % ph == 0: 4 adjacent columns per thread, 16-byte accesses
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_apply_v4_kernel(const float* __restrict__ dpooled,
                                                                        const float* __restrict__ pooled,
                                                                        const uint8_t* __restrict__ idx,
                                                                        const float* __restrict__ y, PoolGeom g,
                                                                        const float* __restrict__ mean,
                                                                        const float* __restrict__ invstd,
                                                                        const float* __restrict__ gamma,
                                                                        const float* __restrict__ red, float inv_count,
                                                                        int train, float* __restrict__ dy, DropP dr) {
    const int ow = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (ow >= g.OW) return;
    if (dr.p > 0.f && dr.state) dr.offset += dr.state[0];
    const int oh = blockIdx.y;
    const long long nc = blockIdx.z;
    const int c = (int)(nc % g.C);
    const float mu = mean[c], is = invstd[c];
    const float a = gamma[c] * is;
    const float k1 = train ? red[g.C + c] * inv_count : 0.f;
    const float k2 = train ? red[c] * inv_count : 0.f;
    // dy = a * (d - k1 - (y - mu) * is * k2) = y * c1 + d * a + c0
    const float c1 = -a * is * k2, c0 = a * (mu * is * k2 - k1);
    const size_t o = ((size_t)nc * g.OH + oh) * g.OW + ow;
    const float4 pz = *reinterpret_cast<const float4*>(pooled + o);
    float4 dp = *reinterpret_cast<const float4*>(dpooled + o);
    if (dr.p > 0.f) {                     // dpooled is the gradient behind the stage's Dropout: the same mask
        const float4 mk = dropout_mask4(dr.offset + (uint64_t)(o >> 2), dr.seed, dr.p, dr.scale);
        dp.x *= mk.x; dp.y *= mk.y; dp.z *= mk.z; dp.w *= mk.w;
    }
    const uchar4 am = *reinterpret_cast<const uchar4*>(idx + o);
    const float dz[4] = {pz.x > 0.f ? dp.x : 0.f, pz.y > 0.f ? dp.y : 0.f, pz.z > 0.f ? dp.z : 0.f, pz.w > 0.f ? dp.w : 0.f};
    const int ai[4] = {am.x, am.y, am.z, am.w};
    const size_t base = ((size_t)nc * g.H + (size_t)oh * g.ph) * g.W + ow;
    for (int r = 0; r < g.ph; ++r) {
        const float4 v = *reinterpret_cast<const float4*>(y + base + (size_t)r * g.W);
        float4 out;
        out.x = v.x * c1 + (ai[0] == r ? dz[0] * a : 0.f) + c0;
        out.y = v.y * c1 + (ai[1] == r ? dz[1] * a : 0.f) + c0;
        out.z = v.z * c1 + (ai[2] == r ? dz[2] * a : 0.f) + c0;
        out.w = v.w * c1 + (ai[3] == r ? dz[3] * a : 0.f) + c0;
        *reinterpret_cast<float4*>(dy + base + (size_t)r * g.W) = out;
    }
}

// coef = [c1 | a | c0] with  dy = y*c1 + dz*a + c0  (the apply kernel's formula, per channel); dbias += sum(dy)
__global__ void bn_pool_coef_kernel(const float* __restrict__ red, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma, int C,
                                    float inv_count, int train, float* __restrict__ coef, float* __restrict__ dbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mu = mean[c], is = invstd[c], a = gamma[c] * is;
    const float k1 = train ? red[C + c] * inv_count : 0.f;
    const float k2 = train ? red[c] * inv_count : 0.f;
    coef[c] = -a * is * k2;
    coef[C + c] = a;
    coef[2 * C + c] = a * (mu * is * k2 - k1);
    // sum over positions of dy: a * (sum dz - P*k1 - k2 * sum xhat); with batch statistics both corrections cancel it
    if (dbias) dbias[c] += train ? 0.f : a * red[C + c];
}

static inline unsigned grid_cap(long long items) {
    long long b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > 16384) b = 16384;
    return (unsigned)b;
}

}  // namespace seld
using namespace seld;

static int mk_geom(PoolGeom& g, int N, int C, int H, int W, int ph, int pw) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || ph * pw > 255 || H < ph || W < pw) return SELD_EINVAL;
    if ((long long)N * C > 65535 || (H + ph - 1) / ph > 65535) return SELD_EUNSUPPORTED;   // grid.z / grid.y limits
    g.NC = (long long)N * C; g.C = C; g.H = H; g.W = W; g.ph = ph; g.pw = pw; g.OH = H / ph; g.OW = W / pw;
    return SELD_OK;
}

static int bn_relu_pool_fwd_impl(const float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw,
                                const float* mean, const float* invstd, const float* gamma, const float* beta,
                                float* pooled, uint8_t* idx, const DropP& dr, float* dropped, void* stream) {
    PoolGeom g;
    int rc = mk_geom(g, N, C, H, W, ph, pw);
    if (rc) return rc;
    if (!y || !mean || !invstd || !gamma || !beta || !pooled || !idx) return SELD_EINVAL;
    if (pw == 1 && W % 4 == 0) {
        const dim3 grid((g.OW / 4 + 127) / 128, g.OH, (unsigned)g.NC);
        if (ph == 8)
            hipLaunchKernelGGL(bn_relu_pool_fwd_v4_kernel<8>, grid, dim3(128), 0, (hipStream_t)stream, y, g, mean, invstd, gamma,
                               beta, pooled, idx, dr, dropped);
        else if (ph == 2)
            hipLaunchKernelGGL(bn_relu_pool_fwd_v4_kernel<2>, grid, dim3(128), 0, (hipStream_t)stream, y, g, mean, invstd, gamma,
                               beta, pooled, idx, dr, dropped);
        else
            hipLaunchKernelGGL(bn_relu_pool_fwd_v4_kernel<0>, grid, dim3(128), 0, (hipStream_t)stream, y, g, mean, invstd, gamma,
                               beta, pooled, idx, dr, dropped);
    } else {
        if (dr.p > 0.f) return SELD_EUNSUPPORTED;
        hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3((g.OW + 255) / 256, g.OH, (unsigned)g.NC), dim3(256), 0, (hipStream_t)stream, y, g,
                           mean, invstd, gamma, beta, pooled, idx);
    }
    return check_launch();
}

extern "C" int seld_bn_relu_pool_fwd(const float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw,
                                     const float* mean, const float* invstd, const float* gamma, const float* beta,
                                     float* pooled, uint8_t* idx, void* stream) {
    return bn_relu_pool_fwd_impl(y, N, C, H, W, ph, pw, mean, invstd, gamma, beta, pooled, idx, DropP{0.f, 1.f, 0, 0, nullptr},
                                 nullptr, stream);
}

extern "C" int seld_bn_relu_pool_drop_ok(int32_t H, int32_t W, int32_t ph, int32_t pw) {
    return pw == 1 && W % 4 == 0 && ph > 0 && H % ph == 0;
}

extern "C" int seld_bn_relu_pool_fwd_drop(const float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw,
                                          const float* mean, const float* invstd, const float* gamma, const float* beta,
                                          float* pooled, uint8_t* idx, float drop_p, uint64_t seed, uint64_t offset,
                                          const uint64_t* state, float* dropped, void* stream) {
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !dropped)) return SELD_EINVAL;
    if (drop_p > 0.f && !seld_bn_relu_pool_drop_ok(H, W, ph, pw)) return SELD_EUNSUPPORTED;
    return bn_relu_pool_fwd_impl(y, N, C, H, W, ph, pw, mean, invstd, gamma, beta, pooled, idx,
                                 DropP{drop_p, 1.0f / (1.0f - drop_p), seed, offset, state}, dropped, stream);
}

// First half of seld_bn_relu_pool_bwd for a convolution whose INPUT needs no gradient (the first layer): the
// reductions (red = dgamma | dbeta sums, from pooled-size tensors) and the three per-channel coefficients with which
// seld_hc_conv_bwd_weight_bnpool_acc forms dy on the fly.  The 1.6 GB gradient w.r.t. the conv output is never written.
extern "C" int seld_bn_relu_pool_bwd_coef_drop(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y,
                                               int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw,
                                               const float* mean, const float* invstd, const float* gamma, const float* beta,
                                               int32_t train, float* red, float* coef, float* conv_dbias, float drop_p,
                                               uint64_t seed, uint64_t offset, const uint64_t* state, void* stream) {
    if (drop_p < 0.f || drop_p >= 1.f) return SELD_EINVAL;
    const DropP dr{drop_p, 1.0f / (1.0f - drop_p), seed, offset, state};
    PoolGeom g;
    int rc = mk_geom(g, N, C, H, W, ph, pw);
    if (rc) return rc;
    if (!dpooled || !pooled || !idx || !y || !mean || !invstd || !gamma || !beta || !red || !coef) return SELD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int S = g.OH * g.OW;
    const long long M = (long long)N * S;
    hipLaunchKernelGGL(bn_relu_pool_bwd_reduce_kernel, dim3(env().deterministic ? 1u : (unsigned)((M + 8191) / 8192), C), dim3(256), 0, st, dpooled, pooled,
                       N, C, S, gamma, beta, y, idx, g, mean, invstd, red, dr);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(bn_pool_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, red, mean, invstd, gamma, C,
                       1.0f / (float)((long long)N * H * W), train, coef, conv_dbias);
    return check_launch();
}

extern "C" int seld_bn_relu_pool_bwd_coef(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y,
                                          int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw,
                                          const float* mean, const float* invstd, const float* gamma, const float* beta,
                                          int32_t train, float* red, float* coef, float* conv_dbias, void* stream) {
    return seld_bn_relu_pool_bwd_coef_drop(dpooled, pooled, idx, y, N, C, H, W, ph, pw, mean, invstd, gamma, beta, train, red,
                                           coef, conv_dbias, 0.f, 0, 0, nullptr, stream);
}

static int bn_relu_pool_bwd_impl(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y,
                                int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, int32_t train, float* red, float* dy,
                                const DropP& dr, void* stream) {
    PoolGeom g;
    int rc = mk_geom(g, N, C, H, W, ph, pw);
    if (rc) return rc;
    if (!dpooled || !pooled || !idx || !y || !mean || !invstd || !gamma || !beta || !red || !dy) return SELD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int S = g.OH * g.OW;
    const long long M = (long long)N * S;
    const bool v4 = pw == 1 && W % 4 == 0 && H % ph == 0;
    if (dr.p > 0.f && !v4) return SELD_EUNSUPPORTED;
    hipLaunchKernelGGL(bn_relu_pool_bwd_reduce_kernel, dim3(env().deterministic ? 1u : (unsigned)((M + 8191) / 8192), C), dim3(256), 0, st, dpooled, pooled,
                       N, C, S, gamma, beta, y, idx, g, mean, invstd, red, dr);
    rc = check_launch();
    if (rc) return rc;
    const int OHx = (H + ph - 1) / ph, OWx = (W + pw - 1) / pw;
    if (v4)
        hipLaunchKernelGGL(bn_relu_pool_bwd_apply_v4_kernel, dim3((g.OW / 4 + 127) / 128, g.OH, (unsigned)g.NC), dim3(128), 0, st,
                           dpooled, pooled, idx, y, g, mean, invstd, gamma, red, 1.0f / (float)((long long)N * H * W), train, dy, dr);
    else
        hipLaunchKernelGGL(bn_relu_pool_bwd_apply_kernel, dim3((OWx + 255) / 256, OHx, (unsigned)g.NC), dim3(256), 0, st, dpooled, pooled, idx, y,
                           g, mean, invstd, gamma, red, 1.0f / (float)((long long)N * H * W), train, dy);
    return check_launch();
}

extern "C" int seld_bn_relu_pool_bwd(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y,
                                     int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, int32_t train,
                                     float* red /* (2C) pre-zeroed: dgamma | dbeta */, float* dy, void* stream) {
    return bn_relu_pool_bwd_impl(dpooled, pooled, idx, y, N, C, H, W, ph, pw, mean, invstd, gamma, beta, train, red, dy,
                                 DropP{0.f, 1.f, 0, 0, nullptr}, stream);
}

extern "C" int seld_bn_relu_pool_bwd_drop(const float* dpooled, const float* pooled, const uint8_t* idx, const float* y,
                                          int32_t N, int32_t C, int32_t H, int32_t W, int32_t ph, int32_t pw, const float* mean,
                                          const float* invstd, const float* gamma, const float* beta, int32_t train,
                                          float* red, float* dy, float drop_p, uint64_t seed, uint64_t offset,
                                          const uint64_t* state, void* stream) {
    if (drop_p < 0.f || drop_p >= 1.f) return SELD_EINVAL;
    return bn_relu_pool_bwd_impl(dpooled, pooled, idx, y, N, C, H, W, ph, pw, mean, invstd, gamma, beta, train, red, dy,
                                 DropP{drop_p, 1.0f / (1.0f - drop_p), seed, offset, state}, stream);
}
