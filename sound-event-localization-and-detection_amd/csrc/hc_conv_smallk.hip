// Hypercomplex convolution forward for SHORT reductions (Cin * kh * kw <= 160), gfx950: the first CNN layer.
//
// For `cnn.0` (8 -> 192 channels, 3x3, 52 MB and 1.36 GFLOP per sample, model.py:273-274) the reduction is
// only 72 deep, so the general kernel's K loop is all prologue: it ran latency-bound at ~1.1 TB/s while the
// layer's roof is the 1.6 GB output write.  This variant is persistent and weight-stationary:
//   * one 8-wave workgroup per CU keeps the WHOLE expanded, signed weight tile [K/4][Cout][4] in LDS (55 KB)
//     for its lifetime -- the Hamilton assembly happens once per workgroup, not once per tile;
//   * it walks 128-position tiles; the im2col image of tile t+1 (18 KB) is gathered with buffer loads
//     (hardware zero-fill for the halo) into registers while tile t is multiplied, then dropped into the
//     other half of a double-buffered LDS image: one barrier per tile;
//   * MFMA granularity is one k-group (4 taps) per v_mfma_f32_16x16x4_f32 with ds_read_b32 operands, so the
//     72-deep reduction costs exactly 18 MFMAs and the primal output channels stop after the 9 groups that
//     are not structurally zero (162 instead of 240 MFMAs per 16 x 192 output tile);
//   * each lane stores 4 consecutive positions of a channel (16-byte stores); BatchNorm statistics are kept in
//     registers across tiles and leave the workgroup as ONE set of atomics.
#include "hc_common.h"
#include <type_traits>

namespace seld {

// Waves per workgroup.  4-wave workgroups, two per CU (each with its own copy of the weights in LDS), run out of
// phase with each other, so one workgroup's stores / gathers hide under the other's MFMAs; an 8-wave workgroup
// per CU keeps its two waves per SIMD in lock-step.  SELD_SMALLK_NW=8 selects the latter (tuning aid).
// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence over ALL address spaces: on
// gfx9 it waits vmcnt(0), i.e. for the write acknowledgement of every global store the wave has in flight -- in a loop
// that stores 12 KB per wave per tile that wait was a third of the kernel.  The image in LDS is the only data the waves
// exchange here; results go to memory nobody in the workgroup reads.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NW> struct SkGeom { static constexpr int TP = NW * 16, NT = NW * 64; };

// NG_T / NGP_T: k-group counts known at compile time (0 = take the run-time arguments).  With constants the two MFMA loops
// unroll completely and the compiler hoists the LDS operand reads of later groups above the MFMAs of earlier ones; as
// run-time loops every group started with an exposed LDS round trip (the MFMA pipe sat idle for a third of the loop).
template <int CT, int KH_T, int KW_T, int SK_NW, int NG_T, int NGP_T>
__global__ __launch_bounds__(SK_NW * 64, (SK_NW == 8 && NG_T) ? 4 : 1) void hc_conv_smallk_kernel(const ConvP p, int NG_arg, int NGP_arg, long long ntiles) {
    const int NG = NG_T ? NG_T : NG_arg;
    const int NGP = NG_T ? NGP_T : NGP_arg;
    constexpr int BC = CT * 16;
    constexpr int SK_TP = SkGeom<SK_NW>::TP, SK_NT = SkGeom<SK_NW>::NT;
    // SPLIT (dual quaternion first layer, 4 waves): a wave owns 32 positions x (CT/4 primal + CT/4 dual) channel tiles instead
    // of 16 positions x all CT tiles.  Same 12 accumulators and MFMA count, but (a) the two 16-position halves of a
    // channel are stored back to back by ONE wave, completing 128-byte lines (with 16 positions per wave every store
    // instruction left sixteen 64-byte half lines for another wave to finish: the write stream ran at 2.3 TB/s), and
    // (b) 8 instead of 13 LDS operand reads per k-group.  Primal and dual tiles are dealt evenly, so the waves stay
    // balanced although primal channels stop after half the groups.
    constexpr bool SPLIT = NG_T && (NGP_T * 2 == NG_T) && (SK_NW == 4 || SK_NW == 8) && (CT % 4 == 0);
    constexpr bool DB = (SK_NW == 8) && !SPLIT;          // double-buffered X image (else single buffer, two barriers)
    // SPLIT keeps the structurally-zero quadrant out of the weight image (groups >= NGP hold the dual channels only): 41
    // instead of 55 KB.  With 8 waves sharing it and a 128-position image that is 78 KB per workgroup: two per CU, four waves
    // per SIMD.
    constexpr int PQ = SK_NW / 2;                        // 32-position quarters (halves) of a tile, one per wave pair
    constexpr int QT = CT / 4;                           // primal (and dual) channel tiles per wave in SPLIT mode
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ws_groups = SPLIT ? NGP * BC + (NG - NGP) * (BC / 2) : NG * BC;     // 4-float weight groups in LDS
    float* Ws = smem;                                    // [NG][BC][4]  (SPLIT: [NGP][BC][4] then [NG-NGP][BC/2][4])
    float* Xs = smem + (size_t)ws_groups * 4;            // [2][NG][SK_TP][4]
    __shared__ const float* wptr_s[8];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int KH = KH_T ? KH_T : p.KH;
    const int KW = KW_T ? KW_T : p.KW;
    const int KK = KH * KW;
    const int A = p.algebra;
    const int CK = p.IA * KK;

    if (tid < 8) wptr_s[tid] = p.w.p[tid];
    __syncthreads();

    // ---- weights: expanded + signed, once per workgroup ------------------------------------------------
    for (int idx = tid; idx < NG * BC; idx += SK_NT) {
        const int g = idx / BC, ch = idx - g * BC;
        const int pp = ch / p.OA, o = ch - pp * p.OA;
        float v[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kk = g * 4 + s;
            float x = 0.f;
            if (kk < p.Ktot) {
                const int qq = kk / CK, l = kk - qq * CK;
                bool zero, neg;
                const int comp = hc_comp(A, pp, qq, &zero, &neg);
                if (!zero) {
                    x = wptr_s[comp][o * CK + l];
                    if (neg) x = -x;
                }
            }
            v[s] = x;
        }
        if constexpr (SPLIT) {
            if (g >= NGP && ch < BC / 2) continue;        // zero quadrant: not stored
            const int slot = g < NGP ? idx : NGP * BC + (g - NGP) * (BC / 2) + (ch - BC / 2);
            *reinterpret_cast<float4*>(&Ws[(size_t)slot * 4]) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            *reinterpret_cast<float4*>(&Ws[(size_t)idx * 4]) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.src, 0, (unsigned)(p.src_elems * 4 > 0xFFFFFFFFLL ? 0xFFFFFFFFu : p.src_elems * 4), 0x00020000);

    // im2col items of one tile: (g, pos), pos = tid & 127 for every item of this thread, g = (tid >> 7) + 4*i
    const int xpos = tid & (SK_TP - 1);
    const int xg0 = __builtin_amdgcn_readfirstlane(tid / SK_TP);
    constexpr int XI_MAX = NG_T ? (NG_T + 3) / 4 : 10;   // NG <= 40
    const int xi_n = (NG - xg0 + 3) / 4;                 // items of this thread
    float xr[XI_MAX][4];

    auto gather = [&](long long tile) __attribute__((always_inline)) {
        const unsigned pos = (unsigned)(tile * SK_TP) + xpos;
        const bool pvalid = pos < (unsigned)p.Ptot;
        const unsigned img = pos / (unsigned)p.dstS;
        const unsigned rem = pos - img * (unsigned)p.dstS;
        const int oh = rem / (unsigned)p.dstW;
        const int ow = rem - oh * p.dstW;
        const int base_h = oh + p.OFFh, base_w = ow + p.OFFw;      // stride 1
        const unsigned img_b = img * (unsigned)(p.Csrc * p.srcS) * 4u;
#pragma unroll
        for (int i = 0; i < XI_MAX; ++i) {
            if (NG_T || i < xi_n) {                      // compile-time group count: no branch, surplus groups load zeros
                const int g = xg0 + 4 * i;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kk = g * 4 + s;
                    const int kkc = kk < p.Ktot ? kk : 0;
                    const int chan = kkc / KK;
                    const int kidx = kkc - chan * KK;
                    const int kh = kidx / KW, kw = kidx - kh * KW;
                    const int ih = base_h + kh * p.KDh, iw = base_w + kw * p.KDw;
                    const bool ok = pvalid && kk < p.Ktot && g < NG && (unsigned)ih < (unsigned)p.srcH && (unsigned)iw < (unsigned)p.srcW;
                    const unsigned off = ok ? img_b + (unsigned)((chan * p.srcS + ih * p.srcW + iw) * 4) : 0xFFFFFFFFu;
                    xr[i][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
                }
            }
        }
    };
    // SPLIT gather: the k-groups of a thread depend only on its wave (g = wave + 4 i), so with the wave index as a
    // compile-time constant every (channel, kh, kw) of its 20 loads is a constant; a 64-position tile lies in one output
    // row (dstW % 64 == 0), so image / row arithmetic and the row validity are scalar, the column validity is three lane
    // masks per tile.  ~100 instead of ~360 instructions per tile (the loop issued 6.6 non-MFMA instructions per MFMA).
    auto gather_fast = [&](long long tile, auto wc) __attribute__((always_inline)) {
        constexpr int W = decltype(wc)::value;
        const unsigned pos0 = __builtin_amdgcn_readfirstlane((unsigned)(tile * SK_TP));
        const unsigned img = pos0 / (unsigned)p.dstS;
        const unsigned rem = pos0 - img * (unsigned)p.dstS;
        const unsigned oh = rem / (unsigned)p.dstW;
        const unsigned ow0 = rem - oh * (unsigned)p.dstW;
        const int base_h = (int)oh + p.OFFh;
        const int iw0 = (int)ow0 + p.OFFw + xpos;
        const bool c0 = (unsigned)iw0 < (unsigned)p.srcW, c1 = (unsigned)(iw0 + 1) < (unsigned)p.srcW,
                   c2 = (unsigned)(iw0 + 2) < (unsigned)p.srcW;
        const bool r0 = (unsigned)base_h < (unsigned)p.srcH, r1 = (unsigned)(base_h + 1) < (unsigned)p.srcH,
                   r2 = (unsigned)(base_h + 2) < (unsigned)p.srcH;
        const unsigned vbase = (img * (unsigned)(p.Csrc * p.srcS) + (unsigned)(base_h * p.srcW + iw0)) * 4u;
#pragma unroll
        for (int i = 0; i < XI_MAX; ++i) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                constexpr int dummy = 0;
                const int g = W + 4 * i, kk = g * 4 + s;           // constants after unrolling
                const int chan = kk / 9, kh = (kk % 9) / 3, kw = kk % 3;
                const bool live = g < NG_T && kk < NG_T * 4 && kk < 9 * 8;
                const bool ok = live && (kh == 0 ? r0 : kh == 1 ? r1 : r2) && (kw == 0 ? c0 : kw == 1 ? c1 : c2);
                const unsigned off = ok ? vbase + (unsigned)((chan * p.srcS + kh * p.srcW + kw) * 4) : 0xFFFFFFFFu;
                xr[i][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
                (void)dummy;
            }
        }
    };
    auto gather_split = [&](long long tile) __attribute__((always_inline)) {
        if (xg0 == 0) gather_fast(tile, std::integral_constant<int, 0>{});
        else if (xg0 == 1) gather_fast(tile, std::integral_constant<int, 1>{});
        else if (xg0 == 2) gather_fast(tile, std::integral_constant<int, 2>{});
        else gather_fast(tile, std::integral_constant<int, 3>{});
    };
    auto scatter = [&](int buf) __attribute__((always_inline)) {
        float* dst = Xs + (size_t)buf * NG * SK_TP * 4;
#pragma unroll
        for (int i = 0; i < XI_MAX; ++i)
            if (i < xi_n)
                *reinterpret_cast<float4*>(&dst[((size_t)(xg0 + 4 * i) * SK_TP + xpos) * 4]) =
                    make_float4(xr[i][0], xr[i][1], xr[i][2], xr[i][3]);
    };

    float s1[CT], s2[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    float bias_t[CT];                                      // SPLIT: bias of the lane's channel in each of its 2 QT tiles
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int cg = wave / PQ;
        const int ch = ((t % (2 * QT)) < QT ? cg * QT + (t % (2 * QT)) : CT / 2 + cg * QT + ((t % (2 * QT)) - QT)) * 16 + fr;
        bias_t[t] = (SPLIT && p.bias) ? p.bias[ch] : 0.f;
    }

    // experiment switches (wt is unused by the forward): 1 = no stores, 2 = no gathers, 4 = store after the LDS refill;
    // the compile-time-shaped instantiation keeps only bit 4
    const int dbg = NG_T ? (p.wt & 4) : p.wt;
    // Workgroups go round-robin to the 8 XCDs (one L2 each).  seq -> tile gives every XCD a contiguous range of position
    // tiles, so that the 64-position pieces of one 2 KB output row are written back by one L2 instead of eight.
    const bool xcd_ranges = ((ntiles | (long long)gridDim.x) & 7) == 0;
    auto tile_of = [&](long long s) { return xcd_ranges ? (s & 7) * (ntiles >> 3) + (s >> 3) : s; };
    long long seq = blockIdx.x;
    if (seq < ntiles) {
        if constexpr (SPLIT) gather_split(tile_of(seq)); else gather(tile_of(seq));
        scatter(0);
    }
    __syncthreads();

    int buf = 0;
    for (; seq < ntiles; seq += gridDim.x) {
        const long long tile = tile_of(seq);
        const bool has_next = seq + gridDim.x < ntiles;
        const long long next = has_next ? tile_of(seq + gridDim.x) : tile;
        if constexpr (SPLIT) gather_split(next);
        else if (NG_T) gather(next);                                 // unconditional: keeps the loop body one block
        else if (has_next && !(dbg & 2)) gather(next);

        floatx4 acc[CT];                                   // SPLIT: acc[sub * 2 QT + t], t < QT primal, t >= QT dual
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const float b0 = SPLIT ? bias_t[j % (2 * QT)] : 0.f;      // SPLIT: the bias rides in the accumulator
            acc[j] = (floatx4){b0, b0, b0, b0};
        }
        if constexpr (SPLIT) {
            const int half = wave % PQ, cgp = wave / PQ;
            const float* xb = Xs + (size_t)buf * NG * SK_TP * 4 + (size_t)(half * 32 + fr) * 4 + fk;
            const float* wp = Ws + (size_t)(cgp * QT * 16 + fr) * 4 + fk;                      // primal tiles of this wave
            const float* wd = Ws + (size_t)((CT / 2 + cgp * QT) * 16 + fr) * 4 + fk;             // dual tiles
            const float* wd2 = Ws + (size_t)(NGP_T * BC + cgp * QT * 16 + fr) * 4 + fk;         // dual tiles, groups >= NGP
#pragma unroll
            for (int g = 0; g < NG_T; ++g) {
                const float a0 = xb[(size_t)g * SK_TP * 4], a1 = xb[(size_t)g * SK_TP * 4 + 64];
                if (g < NGP_T) {
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        const float b = wp[((size_t)g * BC + t * 16) * 4];
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[t], 0, 0, 0);
                        acc[2 * QT + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[2 * QT + t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const float b = g < NGP_T ? wd[((size_t)g * BC + t * 16) * 4]
                                              : wd2[((size_t)(g - NGP_T) * (BC / 2) + t * 16) * 4];
                    acc[QT + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[QT + t], 0, 0, 0);
                    acc[3 * QT + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[3 * QT + t], 0, 0, 0);
                }
            }
        } else {
        const float* xb = Xs + (size_t)buf * NG * SK_TP * 4 + (size_t)(wave * 16 + fr) * 4 + fk;
        const float* wb = Ws + (size_t)fr * 4 + fk;
        // groups every channel tile needs
#pragma unroll
        for (int g = 0; g < NGP; ++g) {
            const float a = xb[(size_t)g * SK_TP * 4];
#pragma unroll
            for (int j = 0; j < CT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wb[((size_t)g * BC + j * 16) * 4], acc[j], 0, 0, 0);
        }
        // groups in the structurally-zero K half of the primal channels: upper half of the tiles only
#pragma unroll
        for (int g = NGP; g < NG; ++g) {
            const float a = xb[(size_t)g * SK_TP * 4];
#pragma unroll
            for (int j = CT / 2; j < CT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wb[((size_t)g * BC + j * 16) * 4], acc[j], 0, 0, 0);
        }
        }

        auto store_tile = [&]() __attribute__((always_inline)) {
            if constexpr (SPLIT) {
                // lane owns channel TJ*16+fr, positions tile*64 + half*32 + sub*16 + fk*4 .. +3 for sub = 0, 1
                const int half = wave % PQ, cgp = wave / PQ;
                const unsigned pos = (unsigned)(tile * SK_TP) + half * 32 + fk * 4;
                const unsigned img = pos / (unsigned)p.dstS;
                const unsigned rem = pos - img * (unsigned)p.dstS;
                float* drow = p.dst + (size_t)img * p.Cdst * p.dstS + rem;
                const bool ok0 = pos < (unsigned)p.Ptot, ok1 = pos + 16 < (unsigned)p.Ptot;      // dstS % 64 == 0: same image
#pragma unroll
                for (int t = 0; t < 2 * QT; ++t) {
                    const int ch = (t < QT ? cgp * QT + t : CT / 2 + cgp * QT + (t - QT)) * 16 + fr;
                    const floatx4 c0 = acc[t], c1 = acc[2 * QT + t];
                    const float4 o0 = make_float4(c0[0], c0[1], c0[2], c0[3]);
                    const float4 o1 = make_float4(c1[0], c1[1], c1[2], c1[3]);
                    if (ok0) {
                        if (!(dbg & 1)) *reinterpret_cast<float4*>(drow + (size_t)ch * p.dstS) = o0;
                        s1[t] += (o0.x + o0.y) + (o0.z + o0.w);
                        s2[t] += (o0.x * o0.x + o0.y * o0.y) + (o0.z * o0.z + o0.w * o0.w);
                    }
                    if (ok1) {
                        if (!(dbg & 1)) *reinterpret_cast<float4*>(drow + (size_t)ch * p.dstS + 16) = o1;
                        s1[t] += (o1.x + o1.y) + (o1.z + o1.w);
                        s2[t] += (o1.x * o1.x + o1.y * o1.y) + (o1.z * o1.z + o1.w * o1.w);
                    }
                }
                return;
            }
            // store: lane owns channel j*16+fr, positions tile*128 + wave*16 + fk*4 .. +3
            const unsigned pos = (unsigned)(tile * SK_TP) + wave * 16 + fk * 4;
            if (pos < (unsigned)p.Ptot) {
                const unsigned img = pos / (unsigned)p.dstS;
                const unsigned rem = pos - img * (unsigned)p.dstS;
                float* drow = p.dst + (size_t)img * p.Cdst * p.dstS + rem;
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const int ch = j * 16 + fr;
                    const float bvv = p.bias ? p.bias[ch] : 0.f;
                    const float4 o = make_float4(acc[j][0] + bvv, acc[j][1] + bvv, acc[j][2] + bvv, acc[j][3] + bvv);
                    if (!(dbg & 1)) *reinterpret_cast<float4*>(drow + (size_t)ch * p.dstS) = o;
                    s1[j] += (o.x + o.y) + (o.z + o.w);
                    s2[j] += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                }
            }
        };
        if (!(dbg & 4)) store_tile();
        // (SELD_SMALLK_DBG bit 4 stores AFTER the LDS refill instead: 656 vs 648 us.)  What keeps stores, gathers and MFMAs
        // from overlapping is visible in the ISA: the first LDS operand reads of the next tile reuse the VGPRs that held this
        // tile's store addresses / data, and the compiler guards that reuse with s_waitcnt vmcnt(..) -- the next tile's MFMAs
        // start only when this tile's stores are acknowledged.  Stores straight from a second, alternating accumulator set
        // with SGPR-base addressing would remove the reuse; not done yet.
        if (DB) {
            if (has_next) scatter(buf ^ 1);
            lds_barrier();
            buf ^= 1;
        } else {
            lds_barrier();                               // everyone is done reading the image
            if (has_next) scatter(0);
            lds_barrier();
        }

        if (dbg & 4) store_tile();
    }

    if (p.epilogue & SELD_EPI_STATS) {
        // lanes fr, fr+16, fr+32, fr+48 hold the same channel; 8 waves hold different positions
        float* red = Xs;                                   // free now: [SK_NW][BC][2]
        if constexpr (SPLIT) {
            for (int t = tid; t < SK_NW * BC * 2; t += SK_NT) red[t] = 0.f;       // a wave fills its own tiles only
            __syncthreads();
            const int cgp = wave / PQ;
#pragma unroll
            for (int t = 0; t < 2 * QT; ++t) {
                const int ch = (t < QT ? cgp * QT + t : CT / 2 + cgp * QT + (t - QT)) * 16 + fr;
                float a = s1[t], b = s2[t];
                a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
                b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
                if (fk == 0) {
                    red[(wave * BC + ch) * 2 + 0] = a;
                    red[(wave * BC + ch) * 2 + 1] = b;
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            float a = s1[j], b = s2[j];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (fk == 0) {
                red[(wave * BC + j * 16 + fr) * 2 + 0] = a;
                red[(wave * BC + j * 16 + fr) * 2 + 1] = b;
            }
        }
        }
        __syncthreads();
        float* rep = p.stats + (size_t)(blockIdx.x % SELD_STATS_REPLICAS) * 2 * p.Cdst;
        for (int t = tid; t < BC; t += SK_NT) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int wv = 0; wv < SK_NW; ++wv) { a += red[(wv * BC + t) * 2]; b += red[(wv * BC + t) * 2 + 1]; }
            atomicAdd(rep + t, a);
            atomicAdd(rep + p.Cdst + t, b);
        }
    }
}

template <int CT, int NW>
static int launch_smallk(const ConvP& p, int NG, int NGP, hipStream_t st) {
    constexpr int TP = SkGeom<NW>::TP, NT = SkGeom<NW>::NT;
    const long long ntiles = (p.Ptot + TP - 1) / TP;
    const size_t smem_generic = ((size_t)NG * CT * 16 * 4 + (size_t)(NW == 8 ? 2 : 1) * NG * TP * 4) * sizeof(float);
#define SELD_SK(KH_, KW_, NG_, NGP_)                                                                               \
    do {                                                                                                           \
        auto kern = hc_conv_smallk_kernel<CT, KH_, KW_, NW, NG_, NGP_>;                                            \
        constexpr bool split = NG_ && (NGP_ * 2 == NG_) && (CT % 4 == 0);                                          \
        const size_t smem = split ? ((size_t)(NGP_ * CT * 16 + (NG_ - NGP_) * CT * 8) * 4 + (size_t)NG_ * TP * 4) * sizeof(float) \
                                  : smem_generic;                                                                   \
        long long want = (split || NW == 4) ? 512 : 256;       /* two workgroups per CU, or one 8-wave workgroup */   \
        if (env().smallk_wgs) want = env().smallk_wgs;                                                             \
        const unsigned grid = (unsigned)(ntiles < want ? ntiles : want);                                           \
        if (smem > 64 * 1024 &&                                                                                    \
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) \
            return SELD_ELAUNCH;                                                                                   \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), smem, st, p, NG, NGP, ntiles);                              \
    } while (0)
    if (p.KH == 1 && p.KW == 3) SELD_SK(1, 3, 0, 0);
    else if (p.KH == 3 && p.KW == 3 && NG == 18 && NGP == 9 && p.Ktot == 72 && p.dstS % TP == 0 && p.dstW % TP == 0 &&
             p.Ptot % TP == 0 && p.KDh == 1 && p.KDw == 1) SELD_SK(3, 3, 18, 9);   // the 8-channel first layer
    else if (p.KH == 3 && p.KW == 3) SELD_SK(3, 3, 0, 0);
    else SELD_SK(0, 0, 0, 0);
#undef SELD_SK
    return check_launch();
}

// Returns the channel-tile count if the call is (dry_run: would be) handled here (rc in *rc), 0 if the general
// kernel must be used.
int hc_conv_smallk_try(const ConvP& p, hipStream_t st, int* rc, int dry_run) {
    if (env().conv_no_smallk) return 0;
    if (p.mode != MODE_FWD || p.SDh != 1 || p.SDw != 1 || p.SMh != 1 || p.SMw != 1) return 0;
    if (p.Ktot > 160 || (p.epilogue & ~SELD_EPI_STATS)) return 0;
    if (p.dstS % 4 || p.Ptot >= (1LL << 31) || p.src_elems * 4 >= (1LL << 32)) return 0;
    if (p.Ptot < 256 * 128) return 0;                         // not worth a persistent launch
    const int NG = (p.Ktot + 3) / 4;
    int NGP = NG;                                             // groups needed by every tile
    if (p.algebra == 8 && (p.Ktot / 2) % 4 == 0 && (p.Cdst / 2) % 16 == 0) NGP = p.Ktot / 8;
    int ct = 0;
    if (p.Cdst == 192) ct = 12;
    else if (p.Cdst == 128) ct = 8;
    else if (p.Cdst == 64) ct = 4;
    else return 0;
    const int nw = env().smallk_nw;        // 8 only in -DSELD_TUNING builds (untested tuning variant)
    const size_t smem = ((size_t)NG * ct * 16 * 4 + (size_t)(nw == 8 ? 2 : 1) * NG * nw * 16 * 4) * sizeof(float);
    if (smem * (8 / nw) > 156 * 1024) return 0;
    if (dry_run) return ct;
    if (nw == 8) {
        if (ct == 12) *rc = launch_smallk<12, 8>(p, NG, NGP, st);
        else if (ct == 8) *rc = launch_smallk<8, 8>(p, NG, NGP, st);
        else *rc = launch_smallk<4, 8>(p, NG, NGP, st);
    } else {
        if (ct == 12) *rc = launch_smallk<12, 4>(p, NG, NGP, st);
        else if (ct == 8) *rc = launch_smallk<8, 4>(p, NG, NGP, st);
        else *rc = launch_smallk<4, 4>(p, NG, NGP, st);
    }
    return ct;
}

}  // namespace seld
