// Hypercomplex convolution weight gradient: the row-chunk variant (gfx950).
//
// Same split-over-positions implicit GEMM as hc_wgrad32_kernel (hc_wgrad.hip):
//
//     dWfull[co][kk] = sum_pos dy[co][pos] * xcol[kk][pos],   kk = ci*KK + tap
//
// folded into the component gradients with float atomics (quaternion_ops.py:131-147 /
// dual_quaternion_ops.py:122-153 differentiated).  That kernel spent ~2.7 instructions per MFMA on staging
// (PMC: 39 % MFMA busy): sixteen 4-byte im2col gathers per thread and step, each with its own halo test.
// Here a step is 32 consecutive output positions INSIDE ONE OUTPUT ROW (the host requires outW % 32 == 0, true
// for every T = 512 layer), so (image, row, column) of the step are wave-uniform scalars:
//
//   * the step's position goes into the BASE of two buffer descriptors (rebuilt on the scalar unit); the
//     per-thread offsets -- row co of dy, column (ci, tap) of x -- are loop-invariant;
//   * a step whose 32 positions and all taps stay inside the input row ("interior", a scalar test) stages
//     with 16-byte loads and NO vector address arithmetic at all; only the steps at the row ends gather per
//     element with the halo test;
//   * rows / columns outside the tensor carry an out-of-range offset that the descriptor zero-fills.
//
// Tiling, LDS images ([k-group][row][4 positions], one ds_read_b128 feeds four MFMAs), the dual-quaternion
// zero-quadrant exit and the atomic fold are those of hc_wgrad32_kernel.
#include <type_traits>
#include "hc_common.h"

namespace seld {

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

template <int WRW, int RT, int CTL, int KH_T, int KW_T, int FUSED>
__global__ __launch_bounds__(256) void hc_wgrad_row_kernel(const WgradP p) {
    constexpr int WCW = 4 / WRW;
    constexpr int BM = WRW * RT * 16;
    constexpr int BN = WCW * CTL * 16;
    constexpr int AR = (BM + 31) / 32;       // dy rows staged per thread (32 rows per pass)
    constexpr int BR = (BN + 31) / 32;       // x columns staged per thread
    constexpr int KK = KH_T * KW_T;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    // Row pitch = 2 (mod 16) float4: the 16 lanes of one LDS cycle (8 position groups x 2 rows) then write 16
    // different 16-byte bank groups.  (With the natural pitch all 8 position groups hit the same banks: measured
    // one third of the kernel's time.)
    __shared__ __attribute__((aligned(16))) float As[2][8][AR * 32 + 2][4];   // dy   [k-group][co][4 positions]
    __shared__ __attribute__((aligned(16))) float Bs[2][8][BR * 32 + 2][4];   // xcol [k-group][col][4 positions]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr_ = wave / WCW, wc_ = wave % WCW;
    int tile_m, tile_n;
    wgrad_tile(p, &tile_m, &tile_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int CK = p.IA * KK;
    const float* const dyz = blockIdx.y ? p.dy2 : p.dy;            // pair launch: which of the two gradients
    const WPtrsMut& gwz = blockIdx.y ? p.gw2 : p.gw;

    if (p.algebra == 8 && m0 + BM <= (p.Cout >> 1) && n0 >= (p.Ktot >> 1)) return;   // zero quadrant

    const long long pbeg = (long long)blockIdx.x * p.split_len;          // multiple of 32
    long long pend = pbeg + p.split_len;
    if (pend > p.Ptot) pend = p.Ptot;                                    // Ptot is a multiple of 32 too
    const int nchunks = pbeg < pend ? (int)((pend - pbeg) >> 5) : 0;

    const int g = tid & 7;                   // 4-position group inside the 32-position step
    const int rsub = tid >> 3;               // 0..31

    // ---- loop-invariant per-thread offsets (bytes) ------------------------------------------------------------
    unsigned a_voff[AR];
    // FUSED: the row's offset in the pooled-size tensors and its three BatchNorm-backward coefficients
    const int pooledS = FUSED ? (p.outH / p.poolh) * p.outW : 0;
    const uint64_t drop_off = (FUSED && p.drop.p > 0.f) ? p.drop.offset + (p.drop.state ? p.drop.state[0] : 0) : 0;
    unsigned p_voff[AR];
    float c_1[AR], c_a[AR], c_0[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int r = rsub + 32 * j;
        const bool ok = r < BM && (m0 + r) < p.Cout;
        a_voff[j] = ok ? (unsigned)(((m0 + r) * p.outS + 4 * g) * 4) : OOB;
        if (FUSED) {
            p_voff[j] = ok ? (unsigned)((m0 + r) * pooledS + 4 * g) : OOB;        // in ELEMENTS (float and uint8 tensors)
            c_1[j] = ok ? p.coef[m0 + r] : 0.f;
            c_a[j] = ok ? p.coef[p.Cout + m0 + r] : 0.f;
            c_0[j] = ok ? p.coef[2 * p.Cout + m0 + r] : 0.f;
        }
    }
    // The x descriptor is based (ph rows + pw columns) BEFORE the step's first input element, so that every
    // in-range element has a non-negative offset: column (ci, kh, kw) of position group g sits at
    // ci*inS + kh*dh*inW + kw*dw + 4g.
    unsigned b_voff[BR];
    int b_row[BR], b_col[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int c = rsub + 32 * j;
        const int kk = n0 + c;
        const bool ok = c < BN && kk < p.Ktot;
        const int kkc = ok ? kk : 0;
        const int ci = kkc / KK;
        const int tap = kkc - ci * KK;
        const int kh = tap / KW_T, kw = tap - kh * KW_T;
        b_voff[j] = ok ? (unsigned)((ci * p.inS + kh * p.dh * p.inW + kw * p.dw + 4 * g) * 4) : OOB;
        b_row[j] = kh * p.dh - p.ph;
        b_col[j] = kw * p.dw - p.pw + 4 * g;
    }
    const bool rows_trivial = (KH_T == 1) && (p.ph == 0) && (p.sh == 1);   // 1-D layers: the only row is always valid

    // ---- wave-uniform tracker of the step: position pbeg + 32*chunk = (img, oh, ow) ----------------------------
    // FUSED walks an image window by window: the ph rows of a pooling window are consecutive steps, so that the
    // pooled-size tensors are loaded once per window (with the plain row order each pooled element was re-fetched
    // ph times, 16 steps apart: PMC showed 6.5 GB of traffic for 2 GB of operands).
    int t_img, t_oh, t_ow;
    bool pool_new = true;                  // FUSED: the step to load starts a new window (or the split)
    {
        const long long im = pbeg / p.outS;
        const int rem = (int)(pbeg - im * p.outS);
        t_img = (int)im;
        if (FUSED) {
            const int cu = rem >> 5, wc = p.outW >> 5;             // step inside the image, 32-wide columns per row
            const int qh = cu / (wc * p.poolh);
            const int r2 = cu - qh * (wc * p.poolh);
            const int owc = r2 / p.poolh;
            t_oh = qh * p.poolh + (r2 - owc * p.poolh);
            t_ow = owc * 32;
        } else {
            t_oh = rem / p.outW;
            t_ow = rem - t_oh * p.outW;
        }
    }
    const long long dy_img = (long long)p.Cout * p.outS;
    const long long x_img = (long long)p.Cin * p.inS;
    const unsigned nrec_a = (unsigned)(dy_img * 4 > (long long)OOB ? (long long)OOB : dy_img * 4);
    const long long xb = (x_img + (long long)(KH_T * p.dh + p.ph + 1) * p.inW + KW_T * p.dw + 64) * 4;
    const unsigned nrec_b = (unsigned)(xb > (long long)OOB ? (long long)OOB : xb);
    const int wspan = (KW_T - 1) * p.dw - p.pw;          // last tap's column shift

    // DEEP (not FUSED): loads run TWO steps ahead of the MFMAs, in two register sets -- a step's 32-128 MFMAs per wave are
    // shorter than the loaded memory system's latency, so with one step of lead every step ended waiting for its loads
    constexpr bool DEEP = !FUSED && (AR + BR) <= 7;
    constexpr int NSET = DEEP ? 2 : 1;
    floatx4 ars[NSET][AR], brs[NSET][BR];
    floatx4 pz[AR], pd[AR];          // FUSED: pooled activations and their gradient for the step's window row
    unsigned pi[AR];                 // FUSED: four arg-max bytes
    int prow = 0;                    // FUSED: row of the step inside its pooling window

    // `advance` is false for the prefetch issued during the last step: it re-reads that step (valid addresses) into
    // the LDS buffer nobody reads, so the loop body needs no branch around its loads and stores.
    auto load_chunk = [&](bool advance, auto setc) __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value;
        floatx4 (&ar)[AR] = ars[SET];
        floatx4 (&br)[BR] = brs[SET];
        const float* abase = dyz + (long long)t_img * dy_img + (long long)t_oh * p.outW + t_ow;
        const float* bbase = p.x + (long long)t_img * x_img + (long long)(t_oh * p.sh - p.ph) * p.inW + (t_ow - p.pw);
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, nrec_a, 0x00020000);
        const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)bbase, 0, nrec_b, 0x00020000);
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const uintx4 v = __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_voff[j], 0, 0);
            ar[j] = (floatx4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        }
        if (FUSED) {
            const int qh = t_oh / p.poolh;
            prow = t_oh - qh * p.poolh;
          if (pool_new) {
            const long long pq = (long long)t_img * p.Cout * pooledS + (long long)qh * p.outW + t_ow;
            const unsigned nrec_p = (unsigned)((long long)p.Cout * pooledS * 4);
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.pooled + pq), 0, nrec_p, 0x00020000);
            const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dpooled + pq), 0, nrec_p, 0x00020000);
            const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc((void*)(p.pidx + pq), 0, nrec_p / 4, 0x00020000);
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                const unsigned eo = p_voff[j];
                const uintx4 z = __builtin_amdgcn_raw_buffer_load_b128(zr, eo == OOB ? OOB : eo * 4u, 0, 0);
                const uintx4 d = __builtin_amdgcn_raw_buffer_load_b128(dr, eo == OOB ? OOB : eo * 4u, 0, 0);
                pz[j] = (floatx4){__uint_as_float(z[0]), __uint_as_float(z[1]), __uint_as_float(z[2]), __uint_as_float(z[3])};
                pd[j] = (floatx4){__uint_as_float(d[0]), __uint_as_float(d[1]), __uint_as_float(d[2]), __uint_as_float(d[3])};
                if (p.drop.p > 0.f && eo != OOB) {
                    const float4 mk = dropout_mask4(drop_off + (uint64_t)((pq + (long long)eo) >> 2), p.drop.seed, p.drop.p, p.drop.scale);
                    pd[j][0] *= mk.x; pd[j][1] *= mk.y; pd[j][2] *= mk.z; pd[j][3] *= mk.w;
                }
                pi[j] = __builtin_amdgcn_raw_buffer_load_b32(ir, eo, 0, 0);
            }
          }
        }
        const int ihb = t_oh * p.sh;
        const bool interior = (t_ow - p.pw >= 0) && (t_ow + 31 + wspan < p.inW);      // scalar
        if (interior) {
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                unsigned off = b_voff[j];
                if (!rows_trivial) off = ((unsigned)(ihb + b_row[j]) < (unsigned)p.inH) ? off : OOB;
                const uintx4 v = __builtin_amdgcn_raw_buffer_load_b128(brsrc, off, 0, 0);
                br[j] = (floatx4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
            }
        } else {
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                const bool rowok = (unsigned)(ihb + b_row[j]) < (unsigned)p.inH;
                const int iw = t_ow + b_col[j];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const unsigned off = (rowok && (unsigned)(iw + s) < (unsigned)p.inW) ? b_voff[j] + 4u * s : OOB;
                    br[j][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brsrc, off, 0, 0));
                }
            }
        }
        // advance by 32 positions: rows are a multiple of 32 wide, so a step never straddles two rows
        if (advance) {
            if (FUSED) {
                ++t_oh;
                pool_new = (t_oh % p.poolh) == 0;
                if (pool_new) {                                   // window done: next 32 columns, same window row block
                    t_oh -= p.poolh;
                    t_ow += 32;
                    if (t_ow >= p.outW) {
                        t_ow = 0;
                        t_oh += p.poolh;
                        if (t_oh >= p.outH) { t_oh = 0; ++t_img; }
                    }
                }
            } else {
                t_ow += 32;
                if (t_ow >= p.outW) {
                    t_ow = 0;
                    if (++t_oh >= p.outH) { t_oh = 0; ++t_img; }
                }
            }
        }
    };
    auto store_chunk = [&](int buf, auto setc) __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value;
        floatx4 (&ar)[AR] = ars[SET];
        floatx4 (&br)[BR] = brs[SET];
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            floatx4 a4 = ar[j];
            if (FUSED) {
                // gradient w.r.t. the conv output from the conv output itself and the pooled-size tensors (the
                // formula of bn_relu_pool_bwd_apply_kernel); rows outside the tensor have all-zero coefficients
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool hit = (pz[j][e] > 0.f) && ((int)((pi[j] >> (8 * e)) & 0xFFu) == prow);
                    a4[e] = ar[j][e] * c_1[j] + (hit ? pd[j][e] * c_a[j] : 0.f) + c_0[j];
                }
            }
            *reinterpret_cast<floatx4*>(&As[buf][g][rsub + 32 * j][0]) = a4;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) *reinterpret_cast<floatx4*>(&Bs[buf][g][rsub + 32 * j][0]) = br[j];
    };

    floatx4 acc[RT][CTL];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CTL; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fk = lane >> 4;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NSET - 1>;
    auto mfma_step = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            floatx4 av[RT], bv[CTL];
#pragma unroll
            for (int i = 0; i < RT; ++i) av[i] = *reinterpret_cast<const floatx4*>(&As[buf][half * 4 + fk][wr_ * (RT * 16) + i * 16 + fr][0]);
#pragma unroll
            for (int j = 0; j < CTL; ++j) bv[j] = *reinterpret_cast<const floatx4*>(&Bs[buf][half * 4 + fk][wc_ * (CTL * 16) + j * 16 + fr][0]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (half == 1 && s == 3) __builtin_amdgcn_sched_barrier(0);    // stores may mix with the last k-step only
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CTL; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
            }
        }
    };
    if constexpr (DEEP) {
        // step c multiplies LDS[c & 1]; register set (c & 1) holds step c + 1 (stored to LDS after the MFMAs); the loads
        // issued in step c are those of step c + 2, into the other set
        if (nchunks > 0) { load_chunk(nchunks > 1, S0{}); store_chunk(0, S0{}); }
        if (nchunks > 1) load_chunk(nchunks > 2, S0{});
        __syncthreads();
        auto body = [&](int chunk, auto cur, auto oth) __attribute__((always_inline)) {
            const int buf = chunk & 1;
            if (chunk + 2 < nchunks) load_chunk(chunk + 3 < nchunks, oth);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(buf);
            if (chunk + 1 < nchunks) store_chunk(buf ^ 1, cur);
            __syncthreads();
        };
        for (int chunk = 0; chunk < nchunks; chunk += 2) {
            body(chunk, S0{}, S1{});
            if (chunk + 1 < nchunks) body(chunk + 1, S1{}, S0{});
        }
    } else {
        if (nchunks > 0) { load_chunk(nchunks > 1, S0{}); store_chunk(0, S0{}); }
        __syncthreads();
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int buf = chunk & 1;
            if (!(p.dbg & 1)) load_chunk(chunk + 2 < nchunks, S0{});
            // The loads must stay at the top and the LDS stores at the bottom of the step: left alone, the scheduler
            // hoists the stores (and their vmcnt wait) to the middle, which leaves the loads 48 MFMAs to land.
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(buf);
            if (!(p.dbg & 2)) store_chunk(buf ^ 1, S0{});
            __syncthreads();
        }
    }

    // ---- fold the tile into the component gradients ------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < CTL; ++j) {
        const int kk = n0 + wc_ * (CTL * 16) + j * 16 + fr;
        if (kk >= p.Ktot) continue;
        const int qq = kk / CK;
        const int ckl = kk - qq * CK;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m0 + wr_ * (RT * 16) + i * 16 + fk * 4 + r;
                if (co >= p.Cout) continue;
                const int pp = co / p.OA;
                const int o = co - pp * p.OA;
                bool zero, neg;
                const int comp = hc_comp(p.algebra, pp, qq, &zero, &neg);
                if (zero) continue;
                const float v = acc[i][j][r];
                atomicAdd(gwz.p[comp] + (size_t)o * CK + ckl, neg ? -v : v);
            }
        }
    }
}

bool hc_wgrad_row_ok(const WgradP& p) {
    if (env().wgrad_norow) return false;
    const bool taps = (p.KH == 1 && p.KW == 1) || (p.KH == 1 && p.KW == 3) || (p.KH == 3 && p.KW == 3);
    return taps && p.sw == 1 && (p.outW % 32 == 0) && (p.split_len % 32 == 0) &&
           (long long)p.Cout * p.outS < (1LL << 29) && (long long)p.Cin * p.inS < (1LL << 29);
}

template <int WRW, int RT, int CTL>
static void launch_row(const WgradP& p, hipStream_t st) {
    constexpr int BM = WRW * RT * 16, BN = (4 / WRW) * CTL * 16;
    dim3 grid(p.nsplit, p.nslots > 1 ? 2 : 1, p.mz * p.nact + ((p.Cout + BM - 1) / BM - p.mz) * p.nt);
    if (p.coef) hipLaunchKernelGGL((hc_wgrad_row_kernel<WRW, RT, CTL, 3, 3, 1>), grid, dim3(256), 0, st, p);   // fused BN/pool backward
    else if (p.KH == 3) hipLaunchKernelGGL((hc_wgrad_row_kernel<WRW, RT, CTL, 3, 3, 0>), grid, dim3(256), 0, st, p);
    else if (p.KW == 3) hipLaunchKernelGGL((hc_wgrad_row_kernel<WRW, RT, CTL, 1, 3, 0>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((hc_wgrad_row_kernel<WRW, RT, CTL, 1, 1, 0>), grid, dim3(256), 0, st, p);
}

// cfg as in wgrad_cfg(): 0 = 128 x 128, 1 = 192 x 80, 2 = 64 x 64, 3 = 96 x 128, 4 = 64 x 80, 5 = 64 x 160
void hc_wgrad_row_launch(const WgradP& p_in, int cfg, hipStream_t st) {
    WgradP p = p_in;
    p.dbg = env().wgrad_dbg;            // non-zero only in -DSELD_TUNING builds (timing experiments, wrong results)
    if (cfg == 0) launch_row<2, 4, 4>(p, st);
    else if (cfg == 1) launch_row<4, 3, 5>(p, st);
    else if (cfg == 3) launch_row<2, 3, 4>(p, st);
    else if (cfg == 4) launch_row<4, 1, 5>(p, st);
    else if (cfg == 5) launch_row<4, 1, 10>(p, st);
    else launch_row<2, 2, 2>(p, st);
}

}  // namespace seld
