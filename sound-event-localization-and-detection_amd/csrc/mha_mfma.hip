// Multi-head self-attention core on the fp32 MFMA (v_mfma_f32_16x16x4_f32), gfx950, for head dims that are a multiple
// of 16 and sequence lengths that are a multiple of 16 (config 3: hd = 48, T = 256).  Same contract as the VALU kernels
// of mha.hip (model.py:39-48): tensors (N, E, T) with head h on channels [h*hd, (h+1)*hd), lse = m + log(sum exp).
//
// The (E, T) layout with T contiguous is exactly what the 16x16x4 operands want, so NOTHING is staged through LDS and
// no wave ever waits on another: a wave owns a 16-wide tile of queries (forward, dQ) or keys (dK/dV) and streams the
// other index in 16-wide tiles straight from global memory (the four waves of a workgroup read the same tiles: L1/L2
// hits).  The trick that makes the second GEMM of each step free of data movement: the first product is computed
// TRANSPOSED so that the reduction index of the second one runs over the ROWS of its accumulator tile.  A lane of a
// 16x16 accumulator holds rows 4*(lane/16) .. +3 of column lane%16, which is precisely the B operand (k = lane/16,
// column = lane%16) of four k-steps if step r takes rows {r, 4+r, 8+r, 12+r}; the matching A operand is one 16-byte
// load of 4 consecutive t.
//
//   forward : S^T[key][q] = K^T Q        -> softmax over rows (4 registers + 2 shuffles)   -> O^T[d][q] += V  P^T
//   dQ      : S^T, dP^T[key][q] = V^T dO -> dS^T = P^T (dP^T - delta_q)                   -> dQ^T[d][q] += K  dS^T
//   dK/dV   : S[q][key] = Q^T K, dP[q][key] = dO^T V -> P, dS                              -> dV^T[d][key] += dO P,
//                                                                                             dK^T[d][key] += Q  dS
#include "common.h"
#include "env.h"

namespace seld {

template <int HD>
__global__ __launch_bounds__(256) void mha_fwd_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, int T, int H, long long in_bs,
                                                           float scale, float* __restrict__ out, float* __restrict__ lse) {
    constexpr int KS = HD / 4;       // k-steps of the d reduction
    constexpr int DT = HD / 16;      // 16-row tiles of O^T
    const int lane = threadIdx.x & 63, c = lane & 15, fk = lane >> 4;
    const int wave = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    if (q0 >= T) return;
    const int nh = blockIdx.y;
    const size_t base = (size_t)nh * HD * T;                 // (n*E + h*hd) * T with E = H*hd: out, dout
    // q / k / v (and their gradients) may be channel slices of one (N, 3E, T) tensor: batch stride in_bs floats
    const size_t ibase = (size_t)(nh / H) * (size_t)in_bs + (size_t)(nh % H) * HD * T;
    const float* qb = q + ibase;
    const float* kb = k + ibase;
    const float* vb = v + ibase;

    float qf[KS];                                            // B operand of S^T: Q[d = 4s + fk][query c], pre-scaled
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = qb[(size_t)(4 * s + fk) * T + q0 + c] * scale;
    floatx4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;                            // l: this lane's share of the row sum (its 4 keys per tile)

    // A key tile's operands are requested TOGETHER and one tile ahead of their use (two register stages): the first
    // version asked for each K value right before its MFMA and waited for it there -- 15 exposed cache latencies per tile,
    // 77 us for config 3's 256 (sample, head) pairs.
    float ka[2][KS];                                         // A of S^T: K[d = 4s + fk][key k0 + c]
    float4 va[2][DT];                                        // A of O^T: V[d = 16dt + c][key k0 + 4fk .. +3]
    auto load_tile = [&](int k0, float (&kr)[KS], float4 (&vr)[DT]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KS; ++s) kr[s] = kb[(size_t)(4 * s + fk) * T + k0 + c];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vr[dt] = *reinterpret_cast<const float4*>(vb + (size_t)(dt * 16 + c) * T + k0 + 4 * fk);
    };
    auto tile = [&](const float (&kr)[KS], const float4 (&vr)[DT]) __attribute__((always_inline)) {
        floatx4 st = {0.f, 0.f, 0.f, 0.f};                   // S^T[key k0 + 4fk + r][query c]
#pragma unroll
        for (int s = 0; s < KS; ++s) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[s], qf[s], st, 0, 0, 0);
        float mx = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __expf(m - mn);
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __expf(st[r] - mn);
        l = l * alpha + (p[0] + p[1]) + (p[2] + p[3]);
        m = mn;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            o[dt] *= alpha;
            // A[row = d][k]: step r takes key k0 + 4fk + r -> the 4 consecutive keys of one 16-byte load
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[dt].x, p[0], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[dt].y, p[1], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[dt].z, p[2], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[dt].w, p[3], o[dt], 0, 0, 0);
        }
    };
    load_tile(0, ka[0], va[0]);
    for (int k0 = 0; k0 < T; k0 += 32) {
        if (k0 + 16 < T) load_tile(k0 + 16, ka[1], va[1]);
        tile(ka[0], va[0]);
        if (k0 + 16 >= T) break;
        if (k0 + 32 < T) load_tile(k0 + 32, ka[0], va[0]);
        tile(ka[1], va[1]);
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    float* ob = out + base;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[(size_t)(dt * 16 + 4 * fk + r) * T + q0 + c] = o[dt][r] * inv;
    if (fk == 0) lse[(size_t)nh * T + q0 + c] = m + logf(l);
}

template <int HD>
__global__ __launch_bounds__(256) void mha_bwd_dq_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ dout,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              int T, int H, long long in_bs, float scale,
                                                              float* __restrict__ dq) {
    constexpr int KS = HD / 4, DT = HD / 16;
    const int lane = threadIdx.x & 63, c = lane & 15, fk = lane >> 4;
    const int wave = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    if (q0 >= T) return;
    const int nh = blockIdx.y;
    const size_t base = (size_t)nh * HD * T;
    const size_t ibase = (size_t)(nh / H) * (size_t)in_bs + (size_t)(nh % H) * HD * T;
    const float* qb = q + ibase;
    const float* kb = k + ibase;
    const float* vb = v + ibase;
    const float* gb = dout + base;

    float qf[KS], gf[KS];                                    // B operands: Q (scaled), dO at [d = 4s + fk][query c]
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = qb[(size_t)(4 * s + fk) * T + q0 + c] * scale;
        gf[s] = gb[(size_t)(4 * s + fk) * T + q0 + c];
    }
    const float my_lse = lse[(size_t)nh * T + q0 + c];
    const float my_delta = delta[(size_t)nh * T + q0 + c];
    floatx4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = (floatx4){0.f, 0.f, 0.f, 0.f};

    // operands of a key tile requested together, one tile ahead (see the forward kernel)
    float ka[2][KS], va[2][KS];                              // A of S^T / dP^T: K, V at [d = 4s + fk][key k0 + c]
    float4 kk[2][DT];                                        // A of dQ^T: K[d = 16dt + c][key k0 + 4fk .. +3]
    auto load_tile = [&](int k0, float (&kr)[KS], float (&vr)[KS], float4 (&k4)[DT]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kr[s] = kb[(size_t)(4 * s + fk) * T + k0 + c];
            vr[s] = vb[(size_t)(4 * s + fk) * T + k0 + c];
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) k4[dt] = *reinterpret_cast<const float4*>(kb + (size_t)(dt * 16 + c) * T + k0 + 4 * fk);
    };
    auto tile = [&](const float (&kr)[KS], const float (&vr)[KS], const float4 (&k4)[DT]) __attribute__((always_inline)) {
        floatx4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};     // S^T, dP^T [key k0 + 4fk + r][query c]
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[s], qf[s], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[s], gf[s], dp, 0, 0, 0);
        }
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = __expf(st[r] - my_lse) * (dp[r] - my_delta) * scale;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[dt].x, ds[0], acc[dt], 0, 0, 0);
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[dt].y, ds[1], acc[dt], 0, 0, 0);
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[dt].z, ds[2], acc[dt], 0, 0, 0);
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[dt].w, ds[3], acc[dt], 0, 0, 0);
        }
    };
    load_tile(0, ka[0], va[0], kk[0]);
    for (int k0 = 0; k0 < T; k0 += 32) {
        if (k0 + 16 < T) load_tile(k0 + 16, ka[1], va[1], kk[1]);
        tile(ka[0], va[0], kk[0]);
        if (k0 + 16 >= T) break;
        if (k0 + 32 < T) load_tile(k0 + 32, ka[0], va[0], kk[0]);
        tile(ka[1], va[1], kk[1]);
    }
    float* ob = dq + ibase;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[(size_t)(dt * 16 + 4 * fk + r) * T + q0 + c] = acc[dt][r];
}

template <int HD>
__global__ __launch_bounds__(256) void mha_bwd_dkv_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const float* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               int T, int H, long long in_bs, float scale,
                                                               float* __restrict__ dk, float* __restrict__ dv) {
    constexpr int KS = HD / 4, DT = HD / 16;
    const int lane = threadIdx.x & 63, c = lane & 15, fk = lane >> 4;
    const int wave = threadIdx.x >> 6;
    const int k0 = (blockIdx.x * 4 + wave) * 16;
    if (k0 >= T) return;
    const int nh = blockIdx.y;
    const size_t base = (size_t)nh * HD * T;
    const size_t ibase = (size_t)(nh / H) * (size_t)in_bs + (size_t)(nh % H) * HD * T;
    const float* qb = q + ibase;
    const float* kb = k + ibase;
    const float* vb = v + ibase;
    const float* gb = dout + base;

    float kf[KS], vf[KS];                                    // B operands: K, V at [d = 4s + fk][key c]
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kf[s] = kb[(size_t)(4 * s + fk) * T + k0 + c];
        vf[s] = vb[(size_t)(4 * s + fk) * T + k0 + c];
    }
    floatx4 ak[DT], av[DT];                                  // dK^T, dV^T [d = 16dt + 4fk + r][key c]
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        ak[dt] = (floatx4){0.f, 0.f, 0.f, 0.f};
        av[dt] = (floatx4){0.f, 0.f, 0.f, 0.f};
    }
    // operands of the first two products (S, dP) of a query tile are requested together one tile ahead; those of the last
    // two (dV, dK) at the top of their own tile, 24 MFMAs before their use -- two stages of everything is 260 registers
    struct QTile {
        float aq[KS], ag[KS];                                // A of S / dP: Q, dO at [d = 4s + fk][query q0 + c]
        float4 ls, dl;                                       // lse, delta of queries q0 + 4fk .. +3
    };
    QTile qt[2];
    auto load_tile = [&](int q0, QTile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            t.aq[s] = qb[(size_t)(4 * s + fk) * T + q0 + c];
            t.ag[s] = gb[(size_t)(4 * s + fk) * T + q0 + c];
        }
        t.ls = *reinterpret_cast<const float4*>(lse + (size_t)nh * T + q0 + 4 * fk);
        t.dl = *reinterpret_cast<const float4*>(delta + (size_t)nh * T + q0 + 4 * fk);
    };
    auto tile = [&](int q0, const QTile& t) __attribute__((always_inline)) {
        float4 gg[DT], qq[DT];                               // A of dV^T / dK^T: dO, Q at [d = 16dt + c][query q0 + 4fk .. +3]
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            gg[dt] = *reinterpret_cast<const float4*>(gb + (size_t)(dt * 16 + c) * T + q0 + 4 * fk);
            qq[dt] = *reinterpret_cast<const float4*>(qb + (size_t)(dt * 16 + c) * T + q0 + 4 * fk);
        }
        __builtin_amdgcn_sched_barrier(0);       // (left alone the compiler sinks each of these loads to just before its MFMAs)
        floatx4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};     // S, dP [query q0 + 4fk + r][key c]
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(t.aq[s] * scale, kf[s], st, 0, 0, 0);   // A[row = query c][k = d]
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(t.ag[s], vf[s], dp, 0, 0, 0);
        }
        float p[4], ds[4];
        p[0] = __expf(st[0] - t.ls.x); p[1] = __expf(st[1] - t.ls.y); p[2] = __expf(st[2] - t.ls.z); p[3] = __expf(st[3] - t.ls.w);
        ds[0] = p[0] * (dp[0] - t.dl.x) * scale; ds[1] = p[1] * (dp[1] - t.dl.y) * scale;
        ds[2] = p[2] * (dp[2] - t.dl.z) * scale; ds[3] = p[3] * (dp[3] - t.dl.w) * scale;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gg[dt].x, p[0], av[dt], 0, 0, 0);
            av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gg[dt].y, p[1], av[dt], 0, 0, 0);
            av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gg[dt].z, p[2], av[dt], 0, 0, 0);
            av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gg[dt].w, p[3], av[dt], 0, 0, 0);
            ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[dt].x, ds[0], ak[dt], 0, 0, 0);
            ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[dt].y, ds[1], ak[dt], 0, 0, 0);
            ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[dt].z, ds[2], ak[dt], 0, 0, 0);
            ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qq[dt].w, ds[3], ak[dt], 0, 0, 0);
        }
    };
    load_tile(0, qt[0]);
    for (int q0 = 0; q0 < T; q0 += 32) {
        if (q0 + 16 < T) load_tile(q0 + 16, qt[1]);
        tile(q0, qt[0]);
        if (q0 + 16 >= T) break;
        if (q0 + 32 < T) load_tile(q0 + 32, qt[0]);
        tile(q0 + 16, qt[1]);
    }
    float* okb = dk + ibase;
    float* ovb = dv + ibase;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            okb[(size_t)(dt * 16 + 4 * fk + r) * T + k0 + c] = ak[dt][r];
            ovb[(size_t)(dt * 16 + 4 * fk + r) * T + k0 + c] = av[dt][r];
        }
}

bool mha_mfma_ok(int T, int hd) {
    return (hd == 16 || hd == 32 || hd == 48 || hd == 64) && T % 16 == 0 && !env().mha_no_mfma;
}

template <int HD>
static void fwd_t(const float* q, const float* k, const float* v, int N, int T, int H, long long in_bs, float* out, float* lse,
                  hipStream_t st) {
    hipLaunchKernelGGL((mha_fwd_mfma_kernel<HD>), dim3((T / 16 + 3) / 4, N * H), dim3(256), 0, st, q, k, v, T, H, in_bs,
                       1.0f / sqrtf((float)HD), out, lse);
}
template <int HD>
static int bwd_t(const float* q, const float* k, const float* v, const float* dout, const float* lse, const float* delta,
                 int N, int T, int H, long long in_bs, float* dq, float* dk, float* dv, hipStream_t st) {
    const dim3 grid((T / 16 + 3) / 4, N * H);
    const float scale = 1.0f / sqrtf((float)HD);
    hipLaunchKernelGGL((mha_bwd_dq_mfma_kernel<HD>), grid, dim3(256), 0, st, q, k, v, dout, lse, delta, T, H, in_bs, scale, dq);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL((mha_bwd_dkv_mfma_kernel<HD>), grid, dim3(256), 0, st, q, k, v, dout, lse, delta, T, H, in_bs, scale, dk, dv);
    return check_launch();
}

// in_bs: floats between consecutive samples of q / k / v / dq / dk / dv (H * hd * T for separate tensors)
int mha_mfma_fwd(const float* q, const float* k, const float* v, int N, int T, int H, int hd, long long in_bs, float* out, float* lse,
                 hipStream_t st) {
    if (hd == 16) fwd_t<16>(q, k, v, N, T, H, in_bs, out, lse, st);
    else if (hd == 32) fwd_t<32>(q, k, v, N, T, H, in_bs, out, lse, st);
    else if (hd == 48) fwd_t<48>(q, k, v, N, T, H, in_bs, out, lse, st);
    else fwd_t<64>(q, k, v, N, T, H, in_bs, out, lse, st);
    return check_launch();
}

int mha_mfma_bwd(const float* q, const float* k, const float* v, const float* dout, const float* lse, const float* delta,
                 int N, int T, int H, int hd, long long in_bs, float* dq, float* dk, float* dv, hipStream_t st) {
    if (hd == 16) return bwd_t<16>(q, k, v, dout, lse, delta, N, T, H, in_bs, dq, dk, dv, st);
    if (hd == 32) return bwd_t<32>(q, k, v, dout, lse, delta, N, T, H, in_bs, dq, dk, dv, st);
    if (hd == 48) return bwd_t<48>(q, k, v, dout, lse, delta, N, T, H, in_bs, dq, dk, dv, st);
    return bwd_t<64>(q, k, v, dout, lse, delta, N, T, H, in_bs, dq, dk, dv, st);
}

}  // namespace seld
