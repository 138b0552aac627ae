// Hypercomplex convolution, weight gradient, gfx950.
//
//   dWfull[co][ci*KK + kidx] = sum_{img,pos} dy[img][co][pos] * x[img][ci][src(pos, kidx)]
//
// One implicit GEMM on the fp32 MFMA with the reduction over positions, split over workgroups.  Each
// workgroup folds its tile straight into the COMPONENT gradients with float atomics (sign applied), i.e.
// the transpose of the assembly at quaternion_ops.py:131-135 -- no expanded gradient, no workspace, and
// the tiles of the dual quaternion's structurally-zero quadrant are never computed.
// The atomic payload is nsplit * 0.75 * Cout * Cin * K floats per call (a few MB: far below the
// ~1.3 TB/s chip-wide float-atomic rate, guide Guideline 12).
// Staging follows hc_conv_fwd.hip: wave w stages k-group w (4 consecutive positions) of every row, the
// (image, row, column) of that group is tracked incrementally in SGPRs, per-row decodes are hoisted.
#include "hc_common.h"

namespace seld {

template <int WRW /*waves along rows*/, int RT /*row tiles per wave*/, int CTL /*col tiles per wave*/, int KH_T, int KW_T>
__global__ __launch_bounds__(256) void hc_wgrad_kernel(const WgradP p) {
    constexpr int WCW = 4 / WRW;
    constexpr int BM = WRW * RT * 16;
    constexpr int BN = WCW * CTL * 16;
    constexpr int AR = (BM + 63) / 64;
    constexpr int BR = (BN + 63) / 64;
    __shared__ __attribute__((aligned(16))) float As[2][4][BM][4];   // dy   [k-group][co][4 positions]
    __shared__ __attribute__((aligned(16))) float Bs[2][4][BN][4];   // xcol [k-group][col][4 positions]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr_ = wave / WCW, wc_ = wave % WCW;
    int tile_m, tile_n;
    wgrad_tile(p, &tile_m, &tile_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int split = blockIdx.x;
    const int KH = KH_T ? KH_T : p.KH;
    const int KW = KW_T ? KW_T : p.KW;
    const int KK = KH * KW;
    const int CK = p.IA * KK;

    // structural zero block of the dual-quaternion matrix: rows primal (co < Cout/2), cols dual (ci >= Cin/2)
    if (p.algebra == 8 && m0 + BM <= (p.Cout >> 1) && n0 >= (p.Ktot >> 1)) return;

    const long long pbeg = (long long)split * p.split_len;
    long long pend = pbeg + p.split_len;
    if (pend > p.Ptot) pend = p.Ptot;
    const int nchunks = pbeg < pend ? (int)((pend - pbeg + 15) >> 4) : 0;

    // hoisted per-lane decodes
    bool a_ok[AR];
    int a_co[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int r = lane + 64 * j;
        a_ok[j] = r < BM && (m0 + r) < p.Cout;
        a_co[j] = a_ok[j] ? m0 + r : 0;
    }
    bool b_ok[BR];
    int b_coff[BR], b_dh[BR], b_dw[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int c = lane + 64 * j;
        const int kk = n0 + c;
        b_ok[j] = c < BN && kk < p.Ktot;
        const int kkc = b_ok[j] ? kk : 0;
        const int ci = kkc / KK;
        const int kidx = kkc - ci * KK;
        const int kh = kidx / KW, kw = kidx - kh * KW;
        b_coff[j] = ci * p.inS;
        b_dh[j] = kh * p.dh - p.ph;
        b_dw[j] = kw * p.dw - p.pw;
    }

    // scalar tracker of this wave's first position  pbeg + 16*chunk + 4*wave = (img, oh, ow)
    long long t_img;
    int t_oh, t_ow;
    {
        const long long pos = pbeg + wave * 4;
        t_img = pos / p.outS;
        const int rem = (int)(pos - t_img * p.outS);
        t_oh = rem / p.outW;
        t_ow = rem - t_oh * p.outW;
    }
    const bool a_vec = (p.outS & 3) == 0;

    float ar[AR][4], br[BR][4];

    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
        const long long pos0 = pbeg + (long long)chunk * 16 + wave * 4;
        // the 4 positions of this wave's group, as (img, oh, ow) in scalar registers
        long long im[4];
        int oh[4], ow[4];
        {
            long long i = t_img;
            int h = t_oh, w = t_ow;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                im[s] = i; oh[s] = h; ow[s] = w;
                if (++w >= p.outW) { w = 0; if (++h >= p.outH) { h = 0; ++i; } }
            }
        }
        if (a_vec) {
            const bool pin = pos0 < pend;        // outS % 4 == 0: the four positions share validity and image
            const size_t base = (size_t)im[0] * p.Cout * p.outS + (size_t)oh[0] * p.outW + ow[0];
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pin && a_ok[j]) v = *reinterpret_cast<const float4*>(p.dy + base + (size_t)a_co[j] * p.outS);
                ar[j][0] = v.x; ar[j][1] = v.y; ar[j][2] = v.z; ar[j][3] = v.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool pin = (pos0 + s) < pend;
                const size_t base = (size_t)im[s] * p.Cout * p.outS + (size_t)oh[s] * p.outW + ow[s];
#pragma unroll
                for (int j = 0; j < AR; ++j) ar[j][s] = (pin && a_ok[j]) ? p.dy[base + (size_t)a_co[j] * p.outS] : 0.f;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool pin = (pos0 + s) < pend;
            const float* xb = p.x + (size_t)im[s] * p.Cin * p.inS;
            const int hh = oh[s] * p.sh, ww = ow[s] * p.sw;
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                const int ih = hh + b_dh[j], iw = ww + b_dw[j];
                float v = 0.f;
                if (pin && b_ok[j] && (unsigned)ih < (unsigned)p.inH && (unsigned)iw < (unsigned)p.inW)
                    v = xb[b_coff[j] + ih * p.inW + iw];
                br[j][s] = v;
            }
        }
        // advance the tracker by 16 positions
        t_ow += 16;
        while (t_ow >= p.outW) {
            t_ow -= p.outW;
            if (++t_oh >= p.outH) { t_oh = 0; ++t_img; }
        }
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const int r = lane + 64 * j;
            if (r < BM) *reinterpret_cast<float4*>(&As[buf][wave][r][0]) = make_float4(ar[j][0], ar[j][1], ar[j][2], ar[j][3]);
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int c = lane + 64 * j;
            if (c < BN) *reinterpret_cast<float4*>(&Bs[buf][wave][c][0]) = make_float4(br[j][0], br[j][1], br[j][2], br[j][3]);
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
        float av[RT][4], bv[CTL][4];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(&As[buf][fk][wr_ * (RT * 16) + i * 16 + fr][0]);
            av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < CTL; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(&Bs[buf][fk][wc_ * (CTL * 16) + j * 16 + fr][0]);
            bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CTL; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
        if (chunk + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    // D layout: col = lane&15 (B index = column kk), row = (lane>>4)*4 + r (A index = co).
    // Fold into the component gradients: block (p, q) -> sign * dw[comp][o][c*KK + kidx].
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
                atomicAdd(p.gw.p[comp] + (size_t)o * CK + ckl, neg ? -v : v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fast variant for stride-1-along-W convolutions whose output rows are a multiple of 4 wide (every shape
// of the SELD models).  Same GEMM, but the K step is 32 positions and the staging is COALESCED: 8 lanes
// cover the 128 contiguous bytes a row contributes to a step (the kernel above lets each lane walk its own
// row, i.e. 64 different cache lines per load instruction).  Each lane tracks the (image, row, column) of
// its own 4-position group incrementally.
// ------------------------------------------------------------------------------------------
template <int WRW, int RT, int CTL, int KH_T, int KW_T>
__global__ __launch_bounds__(256) void hc_wgrad32_kernel(const WgradP p) {
    constexpr int WCW = 4 / WRW;
    constexpr int BM = WRW * RT * 16;
    constexpr int BN = WCW * CTL * 16;
    constexpr int AR = (BM + 31) / 32;       // rows staged per thread (32 rows per pass)
    constexpr int BR = (BN + 31) / 32;
    // row pitch = 2 (mod 16) float4: the 8 position groups of a store then hit different LDS banks
    __shared__ __attribute__((aligned(16))) float As[2][8][(BM + 15) / 16 * 16 + 2][4];   // dy   [k-group][co][4 positions]
    __shared__ __attribute__((aligned(16))) float Bs[2][8][(BN + 15) / 16 * 16 + 2][4];   // xcol [k-group][col][4 positions]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr_ = wave / WCW, wc_ = wave % WCW;
    int tile_m, tile_n;
    wgrad_tile(p, &tile_m, &tile_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int split = blockIdx.x;
    const int KH = KH_T ? KH_T : p.KH;
    const int KW = KW_T ? KW_T : p.KW;
    const int KK = KH * KW;
    const int CK = p.IA * KK;

    if (p.algebra == 8 && m0 + BM <= (p.Cout >> 1) && n0 >= (p.Ktot >> 1)) return;   // zero quadrant

    const long long pbeg = (long long)split * p.split_len;
    long long pend = pbeg + p.split_len;
    if (pend > p.Ptot) pend = p.Ptot;
    const int nchunks = pbeg < pend ? (int)((pend - pbeg + 31) >> 5) : 0;

    const int g = tid & 7;                   // 4-position group inside the 32-position step
    const int rsub = tid >> 3;               // 0..31

    bool a_ok[AR];
    int a_off[AR];                           // co * outS
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int r = rsub + 32 * j;
        a_ok[j] = r < BM && (m0 + r) < p.Cout;
        a_off[j] = a_ok[j] ? (m0 + r) * p.outS : 0;
    }
    bool b_ok[BR];
    int b_coff[BR], b_dh[BR], b_dw[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int c = rsub + 32 * j;
        const int kk = n0 + c;
        b_ok[j] = c < BN && kk < p.Ktot;
        const int kkc = b_ok[j] ? kk : 0;
        const int ci = kkc / KK;
        const int kidx = kkc - ci * KK;
        const int kh = kidx / KW, kw = kidx - kh * KW;
        b_coff[j] = ci * p.inS;
        b_dh[j] = kh * p.dh - p.ph;
        b_dw[j] = kw * p.dw - p.pw;
    }

    // per-lane tracker of this lane's group start: position pbeg + 32*chunk + 4*g = (img, oh, ow)
    long long t_pos = pbeg + 4 * g;
    int t_img, t_oh, t_ow;
    {
        const long long im = t_pos / p.outS;
        const int rem = (int)(t_pos - im * p.outS);
        t_img = (int)im;
        t_oh = rem / p.outW;
        t_ow = rem - t_oh * p.outW;
    }
    const size_t dy_img = (size_t)p.Cout * p.outS;
    const size_t x_img = (size_t)p.Cin * p.inS;
    const int img_first = (int)(pbeg / p.outS);                    // scalar: first image of this split
    const long long x_remain = ((long long)p.N - img_first) * (long long)x_img * 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (size_t)img_first * x_img), 0, x_remain > 0xFFFFFFFFLL ? 0xFFFFFFFFu : (unsigned)x_remain, 0x00020000);

    float ar[AR][4], br[BR][4], am[AR];

    auto load_chunk = [&](int) __attribute__((always_inline)) {
        const bool pin = t_pos < pend;          // outW % 4 == 0: the group's 4 positions share row and validity
        // unconditional 16-byte loads (an inactive item re-reads the tensor's first words); validity is applied as a
        // 0/1 multiplier when the registers go to LDS, so no branch and no wait sits between loads and MFMAs
        const float* dyb = pin ? p.dy + (size_t)t_img * dy_img + (size_t)t_oh * p.outW + t_ow : p.dy;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const bool use = pin && a_ok[j];
            const float4 v = *reinterpret_cast<const float4*>(dyb + (use ? a_off[j] : 0));
            ar[j][0] = v.x; ar[j][1] = v.y; ar[j][2] = v.z; ar[j][3] = v.w;
            am[j] = use ? 1.f : 0.f;
        }
        // x gather through a buffer descriptor based at this split's first image: an invalid tap gets the
        // offset 0xFFFFFFFF, which the hardware range check turns into 0.0 (no exec-mask juggling)
        const int xoff = (t_img - img_first) * (int)x_img;
        const int hh = t_oh * p.sh, ww = t_ow;          // sw == 1
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int ih = hh + b_dh[j], iw = ww + b_dw[j];
            const bool rowok = pin && b_ok[j] && (unsigned)ih < (unsigned)p.inH;
            const int base = (xoff + b_coff[j] + ih * p.inW + iw) * 4;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned off = (rowok && (unsigned)(iw + s) < (unsigned)p.inW) ? (unsigned)(base + 4 * s) : 0xFFFFFFFFu;
                br[j][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, off, 0, 0));
            }
        }
        // advance by 32 positions (outW >= 32: at most one wrap per level -> selects, no loop)
        t_pos += 32;
        t_ow += 32;
        const bool wrap_w = t_ow >= p.outW;
        t_ow -= wrap_w ? p.outW : 0;
        t_oh += wrap_w ? 1 : 0;
        const bool wrap_h = t_oh >= p.outH;
        t_oh = wrap_h ? 0 : t_oh;
        t_img += wrap_h ? 1 : 0;
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const int r = rsub + 32 * j;
            if (r < BM) *reinterpret_cast<float4*>(&As[buf][g][r][0]) = make_float4(ar[j][0] * am[j], ar[j][1] * am[j], ar[j][2] * am[j], ar[j][3] * am[j]);
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int c = rsub + 32 * j;
            if (c < BN) *reinterpret_cast<float4*>(&Bs[buf][g][c][0]) = make_float4(br[j][0], br[j][1], br[j][2], br[j][3]);
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
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float av[RT][4], bv[CTL][4];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(&As[buf][half * 4 + fk][wr_ * (RT * 16) + i * 16 + fr][0]);
                av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
            }
#pragma unroll
            for (int j = 0; j < CTL; ++j) {
                const float4 t = *reinterpret_cast<const float4*>(&Bs[buf][half * 4 + fk][wc_ * (CTL * 16) + j * 16 + fr][0]);
                bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CTL; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
        }
        if (chunk + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
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
                atomicAdd(p.gw.p[comp] + (size_t)o * CK + ckl, neg ? -v : v);
            }
        }
    }
}

// per-channel sum over (N, S): dbias  (accumulates)
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
    if (threadIdx.x == 0) out[c] += red[0] + red[1] + red[2] + red[3];
}

struct ZeroP {
    float* p[9];
    long long n[9];
};
__global__ void zero_many_kernel(const ZeroP z) {
    float* dst = z.p[blockIdx.y];
    const long long n = z.n[blockIdx.y];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = 0.f;
}

// Number of position splits.  The kernels keep 2 workgroups per CU resident (4 for the 64 x 64 tile): a launch
// of W workgroups runs in ceil(W / slots) generations of equal length, so the split count is the LARGEST one whose
// active workgroups (tiles outside the dual-quaternion zero quadrant) still fit in one generation -- 21 tiles x 25
// splits = 525 workgroups on 512 slots take twice as long as 21 x 24 (measured: 2088 vs 1422 us on the 3x3 layer).
// Short reductions are split less (at least 512 positions per workgroup) and then run several generations deep.
static int wgrad_splits(const seld_conv_desc* d, int o[2], int bm, int bn, WgradP* p) {
    long long* split_len = &p->split_len;
    const long long Ptot = (long long)d->N * o[0] * o[1];
    const long long Ktot = (long long)d->Cin * d->k[0] * d->k[1];
    const long long mt = (d->Cout + bm - 1) / bm, nt = (Ktot + bn - 1) / bn;
    long long tiles = mt * nt;
    p->mz = 0; p->nact = (int)nt; p->nt = (int)nt;
    if (d->algebra == 8) {
        // tiles that lie wholly in the zero quadrant (rows < Cout/2, columns >= Ktot/2) are not launched
        const long long mz = (d->Cout / 2) / bm;                       // row tiles entirely primal
        const long long nz = nt - ((Ktot / 2) + bn - 1) / bn;          // column tiles entirely in the upper K half
        if (nz > 0 && mz > 0) { p->mz = (int)mz; p->nact = (int)(nt - nz); tiles -= mz * nz; }
    }
    // resident workgroups per CU (LDS-limited): 128x128 / 192x80 / 96x128 / 64x160 -> 2, 64x80 -> 3, 64x64 -> 4
    long long slots = (bm == 64 ? (bn == 64 ? 4 : (bn == 80 ? 3 : 2)) : 2) * 256;
    if (env().wgrad_wgs) slots = env().wgrad_wgs;
    tiles *= (p->nslots > 1 ? 2 : 1);                     // a pair launch carries two gradients
    long long want = slots / tiles;                       // floor: stay within one generation
    if (env().deterministic) want = 1;                    // one position range per tile: a single contribution per element
    const long long maxs = (Ptot + 511) / 512;            // at least 512 positions per split
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    if (want >= 16) want -= want % 8;                     // multiple of 8: the tiles of one split share an XCD (wgrad_tile)
    long long len = (Ptot + want - 1) / want;
    len = (len + 31) / 32 * 32;
    const int ns = (int)((Ptot + len - 1) / len);
    *split_len = len;
    return ns < 1 ? 1 : ns;
}

// tile configuration: 0 = 128 x 128 (waves 2 x 2), 1 = 192 x 80 for short K (first layer), 2 = 64 x 64 (small layers),
// 3 = 96 x 128, 4 = 64 x 80 (short K: a third of the float-atomic chain per gradient element of 192 x 80, x re-read 3x),
// 5 = 64 x 160 (the 16-channel first layer, K = 144: ONE column tile, so the 0.8-1.6 GB dy / y operand is read once --
// with 64 x 64 tiles it was read three times: 744 us at batch 16)
static int wgrad_cfg(const seld_conv_desc* d) {
    const int Ktot = d->Cin * d->k[0] * d->k[1];
    if (env().wgrad_cfg >= 0) return env().wgrad_cfg;     // tuning aid, validated 0..5
    if (Ktot <= 80 && d->Cout > 64) return (d->Cout % 64 == 0) ? 4 : 1;
    if (Ktot <= 160 && d->Cout > 64 && d->Cout % 64 == 0) return 5;
    if (d->Cout <= 64 || Ktot <= 64) return 2;
    // few 128 x 128 tiles and a short reduction (positions / 512 splits at most): take 64 x 64 tiles so that
    // the launch still has >= 2 workgroups per CU
    int o[2];
    hc_out_shape(d, o);
    const long long P = (long long)d->N * o[0] * o[1];
    const long long tiles128 = (long long)((d->Cout + 127) / 128) * ((Ktot + 127) / 128);
    const long long splits = (P + 511) / 512;
    if (tiles128 * (splits < 48 ? splits : 48) < 512) return 2;
    // dual quaternion: 96-row tiles end exactly on the primal/dual boundary when Cout/2 is a multiple of 96,
    // so the whole zero quadrant is skipped (128-row tiles straddle it)
    if (d->algebra == 8 && (d->Cout / 2) % 96 == 0) return 3;
    return 0;
}

bool hc_wgrad_row_ok(const WgradP& p);
void hc_wgrad_row_launch(const WgradP& p, int cfg, hipStream_t st);

static bool wgrad_fast_ok(const WgradP& p) {
    return (p.outW % 4 == 0) && p.outW >= 32 && p.sw == 1 && !env().wgrad_slow;
}

template <int WRW, int RT, int CTL>
static void launch_wgrad(const WgradP& p, hipStream_t st) {
    constexpr int BM = WRW * RT * 16, BN = (4 / WRW) * CTL * 16;
    dim3 grid(p.nsplit, 1, p.mz * p.nact + ((p.Cout + BM - 1) / BM - p.mz) * p.nt);
    const bool fast = wgrad_fast_ok(p);
#define SELD_WG(KH_, KW_)                                                                                         \
    do {                                                                                                          \
        if (fast) hipLaunchKernelGGL((hc_wgrad32_kernel<WRW, RT, CTL, KH_, KW_>), grid, dim3(256), 0, st, p);      \
        else hipLaunchKernelGGL((hc_wgrad_kernel<WRW, RT, CTL, KH_, KW_>), grid, dim3(256), 0, st, p);            \
    } while (0)
    if (p.KH == 1 && p.KW == 1) SELD_WG(1, 1);
    else if (p.KH == 1 && p.KW == 3) SELD_WG(1, 3);
    else if (p.KH == 3 && p.KW == 3) SELD_WG(3, 3);
    else SELD_WG(0, 0);
#undef SELD_WG
}

// One weight gradient, or (dy2 != nullptr) the two of a pair that shares x and the geometry.
static int wgrad_run2(const seld_conv_desc* d, const float* x, const float* dy, float* const dw[8], float* dbias,
                      const float* dy2, float* const dw2[8], float* dbias2, int accumulate, hipStream_t st) {
    int rc = hc_validate(d);
    if (rc) return rc;
    int o[2];
    hc_out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0 || !x || !dy || !dw) return SELD_EINVAL;
    if (dy2 && !dw2) return SELD_EINVAL;
    WgradP p{};
    p.algebra = d->algebra; p.N = d->N; p.Cin = d->Cin; p.Cout = d->Cout;
    p.inH = d->in[0]; p.inW = d->in[1]; p.outH = o[0]; p.outW = o[1];
    p.KH = d->k[0]; p.KW = d->k[1];
    p.sh = d->stride[0]; p.sw = d->stride[1]; p.ph = d->pad[0]; p.pw = d->pad[1]; p.dh = d->dil[0]; p.dw = d->dil[1];
    p.Ktot = d->Cin * p.KH * p.KW;
    p.OA = d->Cout / d->algebra; p.IA = d->Cin / d->algebra;
    p.inS = p.inH * p.inW; p.outS = p.outH * p.outW;
    p.Ptot = (long long)d->N * p.outS;
    p.x = x; p.dy = dy;
    p.nslots = dy2 ? 2 : 1;
    p.dy2 = dy2;
    for (int i = 0; i < 8; ++i) {
        p.gw.p[i] = (i < d->algebra) ? dw[i] : nullptr;
        p.gw2.p[i] = (dy2 && i < d->algebra) ? dw2[i] : nullptr;
    }
    if (!accumulate) {
        for (int sl = 0; sl < p.nslots; ++sl) {
            ZeroP z{};
            const long long per = (long long)p.OA * p.IA * p.KH * p.KW;
            float* const* dws = sl ? dw2 : dw;
            float* db = sl ? dbias2 : dbias;
            int n = 0;
            for (int i = 0; i < d->algebra; ++i) { z.p[n] = dws[i]; z.n[n] = per; ++n; }
            if (db) { z.p[n] = db; z.n[n] = d->Cout; ++n; }
            const long long blocks = (per + 255) / 256;
            hipLaunchKernelGGL(zero_many_kernel, dim3((unsigned)(blocks > 64 ? 64 : blocks), n), dim3(256), 0, st, z);
            rc = check_launch();
            if (rc) return rc;
        }
    }
    const int cfg = wgrad_cfg(d);
    static const int tile_m[6] = {128, 192, 64, 96, 64, 64}, tile_n[6] = {128, 80, 64, 128, 80, 160};
    p.nsplit = wgrad_splits(d, o, tile_m[cfg], tile_n[cfg], &p);
    if (hc_wgrad_row_ok(p)) hc_wgrad_row_launch(p, cfg, st);           // row-chunk staging (hc_wgrad_row.hip)
    else if (dy2) return SELD_EUNSUPPORTED;                            // pairs only on the row kernel
    else if (cfg == 0) launch_wgrad<2, 4, 4>(p, st);
    else if (cfg == 1) launch_wgrad<4, 3, 5>(p, st);
    else if (cfg == 3) launch_wgrad<2, 3, 4>(p, st);
    else if (cfg == 4) launch_wgrad<4, 1, 5>(p, st);
    else if (cfg == 5) launch_wgrad<4, 1, 10>(p, st);
    else launch_wgrad<2, 2, 2>(p, st);
    rc = check_launch();
    if (rc) return rc;
    for (int sl = 0; sl < p.nslots; ++sl) {
        float* db = sl ? dbias2 : dbias;
        if (db) {
            hipLaunchKernelGGL(channel_sum_kernel, dim3(d->Cout), dim3(256), 0, st, sl ? dy2 : dy, d->N, d->Cout, p.outS, db);
            rc = check_launch();
            if (rc) return rc;
        }
    }
    return rc;
}

static int wgrad_run(const seld_conv_desc* d, const float* x, const float* dy, float* const dw[8], float* dbias,
                     int accumulate, hipStream_t st) {
    return wgrad_run2(d, x, dy, dw, dbias, nullptr, nullptr, nullptr, accumulate, st);
}

// Would a pair launch run (row kernel eligibility)?
int hc_wgrad_pair_ok(const seld_conv_desc* d) {
    int o[2];
    hc_out_shape(d, o);
    WgradP p{};
    p.KH = d->k[0]; p.KW = d->k[1]; p.sw = d->stride[1]; p.outW = o[1]; p.split_len = 32;
    p.Cout = d->Cout; p.Cin = d->Cin; p.outS = o[0] * o[1]; p.inS = d->in[0] * d->in[1];
    return hc_wgrad_row_ok(p) ? 1 : 0;
}

int hc_wgrad_label(const seld_conv_desc* d, char* buf, int buflen) {
    int kh = d->k[0], kw = d->k[1];
    if (!((kh == 1 && kw == 1) || (kh == 1 && kw == 3) || (kh == 3 && kw == 3))) kh = kw = 0;
    const int cfg = wgrad_cfg(d);
    const char* t = cfg == 0 ? "2, 4, 4" : (cfg == 1 ? "4, 3, 5" : (cfg == 3 ? "2, 3, 4" : (cfg == 4 ? "4, 1, 5" : (cfg == 5 ? "4, 1, 10" : "2, 2, 2"))));
    int o[2];
    hc_out_shape(d, o);
    const bool fast = (o[1] % 4 == 0) && o[1] >= 32 && d->stride[1] == 1 && !env().wgrad_slow;
    WgradP p{};
    p.KH = d->k[0]; p.KW = d->k[1]; p.sw = d->stride[1]; p.outW = o[1]; p.split_len = 32;
    p.Cout = d->Cout; p.Cin = d->Cin; p.outS = o[0] * o[1]; p.inS = d->in[0] * d->in[1];
    const bool row = hc_wgrad_row_ok(p);
    if (row) snprintf(buf, buflen, "hc_wgrad_row_kernel<%s, %d, %d, 0>", t, kh, kw);     // last argument: fused BN/pool backward
    else snprintf(buf, buflen, "%s<%s, %d, %d>", fast ? "hc_wgrad32_kernel" : "hc_wgrad_kernel", t, kh, kw);
    return SELD_OK;
}

}  // namespace seld
using namespace seld;

// dw[c] += weight gradient of a convolution that is followed by BatchNorm2d -> ReLU -> MaxPool2d(ph, 1) and whose
// input needs no gradient: the gradient w.r.t. the conv output is formed from y, the pooled-size tensors and `coef`
// (seld_bn_relu_pool_bwd_coef) while the operand is staged.  SELD_EUNSUPPORTED when the shape does not qualify.
extern "C" int seld_hc_conv_bwd_weight_bnpool_drop_acc(const seld_conv_desc* d, const float* x, const float* y,
                                                       const float* pooled, const float* dpooled, const uint8_t* idx,
                                                       int32_t ph, const float* coef, float* const dw[8], float drop_p,
                                                       uint64_t seed, uint64_t offset, const uint64_t* state, void* stream);

extern "C" int seld_hc_conv_bwd_weight_bnpool_acc(const seld_conv_desc* d, const float* x, const float* y,
                                                  const float* pooled, const float* dpooled, const uint8_t* idx,
                                                  int32_t ph, const float* coef, float* const dw[8], void* stream) {
    return seld_hc_conv_bwd_weight_bnpool_drop_acc(d, x, y, pooled, dpooled, idx, ph, coef, dw, 0.f, 0, 0, nullptr, stream);
}

// ... with dpooled = the gradient BEHIND the stage's Dropout(drop_p): its mask is replayed while dpooled is loaded
extern "C" int seld_hc_conv_bwd_weight_bnpool_drop_acc(const seld_conv_desc* d, const float* x, const float* y,
                                                       const float* pooled, const float* dpooled, const uint8_t* idx,
                                                       int32_t ph, const float* coef, float* const dw[8], float drop_p,
                                                       uint64_t seed, uint64_t offset, const uint64_t* state, void* stream) {
    if (drop_p < 0.f || drop_p >= 1.f) return SELD_EINVAL;
    int rc = hc_validate(d);
    if (rc) return rc;
    int o[2];
    hc_out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0 || !x || !y || !pooled || !dpooled || !idx || !coef || !dw || ph <= 0) return SELD_EINVAL;
    if (d->k[0] != 3 || d->k[1] != 3 || o[0] % ph != 0 || (long long)d->Cout * (o[0] / ph) * o[1] >= (1LL << 29))
        return SELD_EUNSUPPORTED;
    WgradP p{};
    p.algebra = d->algebra; p.N = d->N; p.Cin = d->Cin; p.Cout = d->Cout;
    p.inH = d->in[0]; p.inW = d->in[1]; p.outH = o[0]; p.outW = o[1];
    p.KH = d->k[0]; p.KW = d->k[1];
    p.sh = d->stride[0]; p.sw = d->stride[1]; p.ph = d->pad[0]; p.pw = d->pad[1]; p.dh = d->dil[0]; p.dw = d->dil[1];
    p.Ktot = d->Cin * p.KH * p.KW;
    p.OA = d->Cout / d->algebra; p.IA = d->Cin / d->algebra;
    p.inS = p.inH * p.inW; p.outS = p.outH * p.outW;
    p.Ptot = (long long)d->N * p.outS;
    p.x = x; p.dy = y;
    p.nslots = 1;
    p.pooled = pooled; p.dpooled = dpooled; p.pidx = idx; p.coef = coef; p.poolh = ph;
    p.drop = DropP{drop_p, 1.0f / (1.0f - drop_p), seed, offset, state};
    for (int i = 0; i < 8; ++i) p.gw.p[i] = (i < d->algebra) ? dw[i] : nullptr;
    const int cfg = wgrad_cfg(d);
    static const int tile_m[6] = {128, 192, 64, 96, 64, 64}, tile_n[6] = {128, 80, 64, 128, 80, 160};
    p.nsplit = wgrad_splits(d, o, tile_m[cfg], tile_n[cfg], &p);
    if (!hc_wgrad_row_ok(p)) return SELD_EUNSUPPORTED;
    hc_wgrad_row_launch(p, cfg, (hipStream_t)stream);
    return check_launch();
}

// dwA[c] += wgrad(x, dyA), dwB[c] += wgrad(x, dyB) (+ bias gradients): one launch for two convolutions that read
// the same input with the same geometry.  SELD_EUNSUPPORTED when the shape does not qualify: call the single form twice.
extern "C" int seld_hc_conv_pair_bwd_weight_acc(const seld_conv_desc* d, const float* x, const float* dyA,
                                                const float* dyB, float* const dwA[8], float* const dwB[8],
                                                float* dbiasA, float* dbiasB, void* stream) {
    if (!dyB || !dwB) return SELD_EINVAL;
    return wgrad_run2(d, x, dyA, dwA, dbiasA, dyB, dwB, dbiasB, 1, (hipStream_t)stream);
}

extern "C" size_t seld_hc_conv_bwd_weight_workspace(const seld_conv_desc* d) {
    (void)d;
    return 0;      // the fold is done with atomics; kept for ABI stability
}

extern "C" int seld_hc_conv_bwd_weight(const seld_conv_desc* d, const float* x, const float* dy,
                                       float* const dw[8], float* dbias, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    return wgrad_run(d, x, dy, dw, dbias, 0, (hipStream_t)stream);
}

// Hamilton fold of a real (Cout, Cin, K) gradient into the component gradients: element (comp, o, i, k) += the signed sum of
// the blocks (po, qi) of the real matrix that hold it, in a fixed order
struct DetFoldP {
    const float* full;
    float* dw[8];
    int A, OA, IA, K;
};
__global__ __launch_bounds__(256) void hc_wgrad_det_fold_kernel(const DetFoldP p) {
    const int per = p.OA * p.IA * p.K;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p.A * per) return;
    const int comp = e / per;
    int rem = e - comp * per;
    const int o = rem / (p.IA * p.K);
    rem -= o * p.IA * p.K;
    const int i = rem / p.K, k = rem - i * p.K;
    float a = 0.f;
    for (int po = 0; po < p.A; ++po)
        for (int qi = 0; qi < p.A; ++qi) {
            float sign;
            if (block_comp(p.A, po, qi, &sign) != comp) continue;
            a += sign * p.full[((size_t)(po * p.OA + o) * (p.A * p.IA) + (qi * p.IA + i)) * p.K + k];
        }
    p.dw[comp][((size_t)o * p.IA + i) * p.K + k] += a;
}

/* Reproducible weight gradient of any convolution this library takes (SELD_DETERMINISTIC mode of the host mirror):
 * the layer is differentiated as the REAL convolution the reference assembles (quaternion_ops.py:131-147,
 * dual_quaternion_ops.py:122-153) with ONE position range per output tile -- every element of the (Cout, Cin, K) gradient
 * has a single contributor -- into `workspace` (Cout * Cin * K floats), and a fold adds the signed blocks to dw[c] in a fixed
 * order.  Slow (no split over positions); same values as seld_hc_conv_bwd_weight_acc up to summation order. */
extern "C" size_t seld_hc_conv_bwd_weight_det_workspace(const seld_conv_desc* d) {
    if (hc_validate(d) != SELD_OK) return 0;
    return (size_t)d->Cout * d->Cin * d->k[0] * d->k[1] * sizeof(float);
}
extern "C" int seld_hc_conv_bwd_weight_det(const seld_conv_desc* d, const float* x, const float* dy, float* const dw[8],
                                           float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = hc_validate(d);
    if (rc) return rc;
    if (!x || !dy || !dw || !workspace) return SELD_EINVAL;
    if (workspace_bytes < seld_hc_conv_bwd_weight_det_workspace(d)) return SELD_EWORKSPACE;
    if (!env().deterministic) return SELD_EUNSUPPORTED;           // the single-range launch plan is tied to the switch
    hipStream_t st = (hipStream_t)stream;
    seld_conv_desc real = *d;
    real.algebra = 1;
    float* full[8] = {(float*)workspace, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    rc = wgrad_run(&real, x, dy, full, nullptr, 0, st);           // zero-fills `full`, then one contribution per element
    if (rc) return rc;
    DetFoldP f{};
    f.full = (const float*)workspace; f.A = d->algebra; f.OA = d->Cout / d->algebra; f.IA = d->Cin / d->algebra;
    f.K = d->k[0] * d->k[1];
    for (int i = 0; i < d->algebra; ++i) {
        if (!dw[i]) return SELD_EINVAL;
        f.dw[i] = dw[i];
    }
    const int total = d->algebra * f.OA * f.IA * f.K;
    hipLaunchKernelGGL(hc_wgrad_det_fold_kernel, dim3((total + 255) / 256), dim3(256), 0, st, f);
    rc = check_launch();
    if (rc) return rc;
    if (dbias) {
        int o[2];
        hc_out_shape(d, o);
        hipLaunchKernelGGL(channel_sum_kernel, dim3(d->Cout), dim3(256), 0, st, dy, d->N, d->Cout, o[0] * o[1], dbias);
        rc = check_launch();
    }
    return rc;
}

// dw[c] += ..., dbias += ...  (gradient accumulation straight into the optimiser's flat gradient buffer)
extern "C" int seld_hc_conv_bwd_weight_acc(const seld_conv_desc* d, const float* x, const float* dy,
                                           float* const dw[8], float* dbias, void* stream) {
    return wgrad_run(d, x, dy, dw, dbias, 1, (hipStream_t)stream);
}
