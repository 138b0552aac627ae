// Weight gradient of the quaternion / dual-quaternion convolutions with the 8-multiplication Hamilton product (gfx950).
//
// For y = W (x) x the gradient is dW = sum over positions of dy (x) conj(x) -- again a Hamilton product per (output block
// channel, input block channel, tap), so the identities of hcq_conv.hip apply with a = dy, b = conj(x): 8 real GEMMs
//     P_m[o][c] = sum_pos F_m(dy)[o][pos] * G_m(conj x)[c][pos],      c = (input block channel, tap)
// instead of 16, recombined once at the end.  Dual quaternion (y_p = Q x_p, y_d = Q2 x_p + Q x_d):
//     dQ = dy_p (x) conj(x_p) + dy_d (x) conj(x_d)        dQ2 = dy_d (x) conj(x_p)
// i.e. three products, two of which share their accumulators: 24 sub-products instead of 48.
//
// Work decomposition: the reduction runs over N*H*W positions and the result is tiny, so the grid is
// (position splits) x (column groups) x (row tiles); one wave owns a 16 (block channels o) x 16 (columns c) tile of all
// 8 forms of dQ and dQ2 (16 accumulators), a workgroup = NW waves on NW neighbouring column tiles of one row tile.
// Per 32-position chunk a workgroup stages the RAW component rows it needs (dy: 16 rows, x: the block channels its
// columns touch, with halo) in LDS; per group of 4 positions a lane reads its 8 dy and 8 x component values, forms the
// 16 + 16 sums in registers (one group ahead of their use) and issues 24 MFMAs.  Partial results of the splits are
// combined with float atomics straight into the gradient tensors (FlatAdam's slices), after the 8 -> 4 recombination.
// Block-channel counts that are 8 (mod 16): the last 8 block channels of the primal and the dual half share one row
// tile (rows 0-7 read dy_p, rows 8-15 dy_d): two products instead of three, of which one is half used.
#include <type_traits>
#include "hc_common.h"

namespace seld {

struct HcqWgP {
    const float* x;
    const float* dy[2];          // one or two convolutions of the same input (pair)
    float* dw[2][8];             // component gradients, accumulated into
    int A;
    int N, Cin, Cout;
    int IB, OB;
    int H, W;
    int KH, KW, dil, dpad;
    int ncol;                    // IB * KH * KW
    int xp;                      // LDS pitch of an x row: 32 + 2*dpad + 2
    int nib_max;                 // x block channels staged per column group
    int row_tiles;               // row tiles per convolution in THIS launch (regular tiles; 1 for the mixed-tile launch)
    long long nchunks;           // 32-position chunks in total
    long long chunks_per_split;
};

typedef unsigned int uintx4w __attribute__((ext_vector_type(4)));

// sums of two components (see hcq_conv.hip): F of the left operand a = dy, G of the right operand b = conj(x)
__device__ __forceinline__ void fforms(const float a[4], float f[8]) {
    f[0] = a[3] + a[1];
    f[1] = a[0] - a[2];
    f[2] = a[0] + a[2];
    f[3] = a[3] - a[1];
    f[4] = a[3] - a[2];
    f[5] = a[1] + a[0];
    f[6] = a[0] - a[1];
    f[7] = a[3] + a[2];
}
__device__ __forceinline__ void gforms_conj(const float x[4], float g[8]) {
    // b = (x0, -x1, -x2, -x3):  b1+b2, b0+b3, b0-b3, b1-b2, b2-b3, b1+b0, b2+b3, b1-b0
    g[0] = -(x[1] + x[2]);
    g[1] = x[0] - x[3];
    g[2] = x[0] + x[3];
    g[3] = x[2] - x[1];
    g[4] = x[3] - x[2];
    g[5] = x[0] - x[1];
    g[6] = -(x[2] + x[3]);
    g[7] = -(x[1] + x[0]);
}

// NW waves per workgroup (column tiles per group); DI / XI: staging items per thread (dy / x), upper bounds;
// KIND 0: quaternion, 1: dual quaternion regular row tiles, 2: the mixed row tile (a launch of its own)
template <int KH, int KW, int NW, int DI, int XI, int KIND>
__global__ __launch_bounds__(64 * NW, 2) void hcq_wgrad_kernel(const HcqWgP p) {
    constexpr int NTH = 64 * NW;
    constexpr int TAPS = KH * KW;
    constexpr int DP = 34;                                       // dy row pitch: conflict-free b32 fragment reads
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    constexpr int A = KIND == 0 ? 4 : 8;
    constexpr int halves = A / 4;                                // 1 quaternion, 2 dual quaternion

    // ---- which tile -----------------------------------------------------------------------------------------------
    const int rt_all = blockIdx.z;
    const int conv = rt_all / p.row_tiles;                       // which convolution of a pair
    const int rt = rt_all - conv * p.row_tiles;
    constexpr bool mix = KIND == 2;
    const int o0 = mix ? p.OB - 8 : rt * 16;
    const int cg = blockIdx.y;
    const int c_lo = cg * (16 * NW);
    const int ib_lo = c_lo / TAPS;
    int ib_hi = (c_lo + 16 * NW + TAPS - 1) / TAPS;
    if (ib_hi > p.IB) ib_hi = p.IB;
    const int nib = ib_hi - ib_lo;
    const long long ch0 = (long long)blockIdx.x * p.chunks_per_split;
    long long ch1 = ch0 + p.chunks_per_split;
    if (ch1 > p.nchunks) ch1 = p.nchunks;
    if (ch0 >= ch1 || c_lo >= p.ncol) return;

    // ---- LDS: dy [comp A][16 rows][DP], x [comp A][nib][KH][xp] -----------------------------------------------------
    float* dys = lds;
    float* xs = lds + A * 16 * DP;
    const int xp = p.xp;
    const int xrow_quads = (32 + 2 * p.dpad) / 4;
    const int x_rows = A * nib * KH;
    const int x_items = x_rows * xrow_quads;
    const int dy_items = A * 16 * 8;

    const long long S = (long long)p.H * p.W;
    const unsigned OOB = 0xFFFFFFF0u;
    const long long xbytes = (long long)p.N * p.Cin * S * 4, dbytes = (long long)p.N * p.Cout * S * 4;
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, xbytes > (long long)OOB ? OOB : (unsigned)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.dy[conv], 0, dbytes > (long long)OOB ? OOB : (unsigned)dbytes, 0x00020000);

    // staging items: loop-invariant parts.  dy item = (comp, row, quad); x item = (comp, ib, kh, quad)
    unsigned d_inv[DI];
    int d_lds[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) {
        const int f = tid + NTH * i;
        const bool in = f < dy_items;
        const int ff = in ? f : 0;
        const int quad = ff & 7, row = (ff >> 3) & 15, comp = ff >> 7;
        int o = o0 + row;
        if (mix && row >= 8) o = p.OB - 1;                       // rows 8..15 of the mixed tile's image are not used
        if (o >= p.OB) o = p.OB - 1;
        d_inv[i] = (unsigned)((((long long)(comp * p.OB + o)) * S + 4 * quad) * 4);
        d_lds[i] = in ? (comp * 16 + row) * DP + 4 * quad : -1;
    }
    unsigned x_inv[XI];
    int x_pos[XI];                                               // LDS float index << 12 | (kh offset + 1) << 10 | (w offset + 512); -1: none
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int f = tid + NTH * i;
        const bool in = f < x_items;
        const int ff = in ? f : 0;
        const int row = ff / xrow_quads, quad = ff - row * xrow_quads;
        const int kh = row % KH;
        const int ci = row / KH;
        const int comp = ci / nib, ibl = ci - comp * nib;
        x_inv[i] = (unsigned)(((long long)(comp * p.IB + ib_lo + ibl)) * S * 4);
        x_pos[i] = in ? ((row * xp + 4 * quad) << 12) | ((kh - (KH - 1) / 2 + 1) << 10) | (4 * quad - p.dpad + 512) : -1;
    }

    // ---- operand addresses ---------------------------------------------------------------------------------------------
    // A (dy): row fr, position 4g + fk.  Mixed tile: rows 0-7 read the primal components, rows 8-15 the dual ones at row - 8
    const int arow = (mix && fr >= 8) ? fr - 8 : fr;
    const int a_half = (mix && fr >= 8) ? 1 : 0;
    const int a_off = arow * DP + fk;
    const int a_cs = 16 * DP;                                     // component stride
    // B (x): column c = c_lo + 16*wave + fr -> (ib, kh, kw); invalid columns read column ncol-1 (never stored)
    int c = c_lo + 16 * wave + fr;
    const bool c_ok = c < p.ncol;
    if (!c_ok) c = p.ncol - 1;
    const int ci_ = c / TAPS, tap = c - ci_ * TAPS;
    const int kh_ = tap / KW, kw_ = tap - kh_ * KW;
    const int b_off = ((ci_ - ib_lo) * KH + kh_) * xp + p.dpad + (kw_ - (KW - 1) / 2) * p.dil + fk;
    const int b_cs = nib * KH * xp;
    const bool wave_on = c_lo + 16 * wave < p.ncol;               // a wave whose column tile lies outside does no MFMAs

    floatx4 acc0[8], acc1[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) { acc0[m] = (floatx4){0.f, 0.f, 0.f, 0.f}; acc1[m] = (floatx4){0.f, 0.f, 0.f, 0.f}; }

    float fa[2][2][8], gb[2][2][8];                                // [stage][half][form]

    auto read_forms = [&](int g, int st) __attribute__((always_inline)) {
        float a[4], b[4];
        if (mix) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = dys[a_off + 4 * g + (a_half * 4 + q) * a_cs];
            fforms(a, fa[st][0]);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (h < halves) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] = dys[a_off + 4 * g + (h * 4 + q) * a_cs];
                    fforms(a, fa[st][h]);
                }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (h < halves) {
#pragma unroll
                for (int q = 0; q < 4; ++q) b[q] = xs[b_off + 4 * g + (h * 4 + q) * b_cs];
                gforms_conj(b, gb[st][h]);
            }
    };
    auto mfmas = [&](int st) __attribute__((always_inline)) {
        if (halves == 1) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc0[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st][0][m], gb[st][0][m], acc0[m], 0, 0, 0);
        } else if (mix) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                acc0[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st][0][m], gb[st][0][m], acc0[m], 0, 0, 0);   // . conj(x_p)
                acc1[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st][0][m], gb[st][1][m], acc1[m], 0, 0, 0);   // . conj(x_d)
            }
        } else {
            // three passes over the forms: an accumulator is not touched by two MFMAs in a row
#pragma unroll
            for (int m = 0; m < 8; ++m) acc0[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st][0][m], gb[st][0][m], acc0[m], 0, 0, 0);   // dQ  += dy_p conj(x_p)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc1[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st][1][m], gb[st][0][m], acc1[m], 0, 0, 0);   // dQ2 += dy_d conj(x_p)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc0[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[st][1][m], gb[st][1][m], acc0[m], 0, 0, 0);   // dQ  += dy_d conj(x_d)
        }
    };

    // ---- chunk loop -------------------------------------------------------------------------------------------------
    const int wq = p.W / 32;                                       // chunks per image row
    floatx4 dr[DI], xr[XI];
    auto load_chunk = [&](long long ch) __attribute__((always_inline)) {
        const long long rowi = ch / wq;                            // n * H + h
        const int w0 = (int)(ch - rowi * wq) * 32;
        const int n_img = (int)(rowi / p.H);
        const int h0 = (int)(rowi - (long long)n_img * p.H);
        const unsigned dbase = (unsigned)((((long long)n_img * p.Cout) * S + (long long)h0 * p.W + w0) * 4);
        const long long xbase = (((long long)n_img * p.Cin) * S + (long long)h0 * p.W + w0) * 4;
#pragma unroll
        for (int i = 0; i < DI; ++i) {
            const uintx4w v = __builtin_amdgcn_raw_buffer_load_b128(drs, d_lds[i] >= 0 ? dbase + d_inv[i] : OOB, 0, 0);
            dr[i][0] = __uint_as_float(v[0]); dr[i][1] = __uint_as_float(v[1]);
            dr[i][2] = __uint_as_float(v[2]); dr[i][3] = __uint_as_float(v[3]);
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int xkh = ((x_pos[i] >> 10) & 3) - 1, xw = (x_pos[i] & 1023) - 512;
            const int hh = h0 + xkh, ww = w0 + xw;
            const bool ok = x_pos[i] >= 0 && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W;
            const long long off = xbase + (long long)x_inv[i] + ((long long)xkh * p.W + xw) * 4;
            const uintx4w v = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? (unsigned)off : OOB, 0, 0);
            xr[i][0] = __uint_as_float(v[0]); xr[i][1] = __uint_as_float(v[1]);
            xr[i][2] = __uint_as_float(v[2]); xr[i][3] = __uint_as_float(v[3]);
        }
    };
    for (long long ch = ch0; ch < ch1; ++ch) {
        load_chunk(ch);
        __syncthreads();                                           // everyone is done reading the previous chunk
#pragma unroll
        for (int i = 0; i < DI; ++i)
            if (d_lds[i] >= 0) {
                *reinterpret_cast<float2*>(dys + d_lds[i]) = make_float2(dr[i][0], dr[i][1]);
                *reinterpret_cast<float2*>(dys + d_lds[i] + 2) = make_float2(dr[i][2], dr[i][3]);
            }
#pragma unroll
        for (int i = 0; i < XI; ++i)
            if (x_pos[i] >= 0) {
                *reinterpret_cast<float2*>(xs + (x_pos[i] >> 12)) = make_float2(xr[i][0], xr[i][1]);
                *reinterpret_cast<float2*>(xs + (x_pos[i] >> 12) + 2) = make_float2(xr[i][2], xr[i][3]);
            }
        __syncthreads();
        // (requesting the next chunk HERE, before the MFMAs, was measured slower on every shape -- 92 vs 88 us on the
        // TCN layer, 206 vs 164 us on the quaternion 3x3 layer: the 48 parked registers cost more than the exposed
        // latency, which the second workgroup of the CU already covers)
        if (wave_on) {
            read_forms(0, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 1 < 8) read_forms(g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(g & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- recombination + atomics -----------------------------------------------------------------------------------
    // lane: column c, rows 4*fk + r.  regular: acc0 -> dQ, acc1 -> dQ2 (rows o0 + row).  mixed: acc0 rows 0-7 -> dQ,
    // rows 8-15 -> dQ2; acc1 rows 8-15 -> dQ (rows 0-7 dropped).  quaternion: acc0 -> dW.
    if (!wave_on || !c_ok) return;
    auto combine = [&](const floatx4* ac, floatx4* out) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float h0_ = 0.5f * ac[0][r], h1 = 0.5f * ac[1][r], h2 = 0.5f * ac[2][r], h3 = 0.5f * ac[3][r];
            out[0][r] = (h3 - h0_) + (h1 + h2) + ac[4][r];
            out[1][r] = (h3 - h0_) - (h1 + h2) + ac[5][r];
            out[2][r] = (h3 + h0_) + (h2 - h1) + ac[6][r];
            out[3][r] = (h3 + h0_) + (h1 - h2) - ac[7][r];
        }
    };
    floatx4 c0[4], c1[4];
    combine(acc0, c0);
    if (halves == 2) combine(acc1, c1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * fk + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (halves == 1) {
                const int o = o0 + row;
                if (o < p.OB) atomicAdd(p.dw[conv][q] + (size_t)o * p.ncol + c, c0[q][r]);
            } else if (!mix) {
                const int o = o0 + row;
                if (o < p.OB) {
                    atomicAdd(p.dw[conv][q] + (size_t)o * p.ncol + c, c0[q][r]);
                    atomicAdd(p.dw[conv][4 + q] + (size_t)o * p.ncol + c, c1[q][r]);
                }
            } else {
                const int o = o0 + (row & 7);
                if (row < 8) atomicAdd(p.dw[conv][q] + (size_t)o * p.ncol + c, c0[q][r]);
                else {
                    atomicAdd(p.dw[conv][4 + q] + (size_t)o * p.ncol + c, c0[q][r]);
                    atomicAdd(p.dw[conv][q] + (size_t)o * p.ncol + c, c1[q][r]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct HcqWgPlan {
    int ok, KH, KW, NW, DI, XI, mix, npair;
    HcqWgP kp;
    dim3 grid;
    size_t smem;
};

static HcqWgPlan hcq_wgrad_plan(const seld_conv_desc* d, int npair) {
    HcqWgPlan pl{};
    if (env().conv_no_hcq) return pl;
    const int A = d->algebra;
    if (A != 4 && A != 8) return pl;
    // Measured against the 16/48-product weight-gradient kernels (tools/hcq_wgrad_check.py): the quaternion layers win
    // (3x3: 164 vs 214 us, 1x3: 18.6 vs 26.0), the dual-quaternion layers of config 3 do not yet (TCN 1x3 88 vs 75 us,
    // cnn.1 1400 vs 1159): the old kernels already run at 65-73 % of their roof, this one at ~35 % of its own.  The dual
    // quaternion therefore takes this kernel only on request (SELD_HCQ_WGRAD_DQ, a selection switch: same results).
    if (A == 8 && !env().hcq_wgrad_dq) return pl;
    if (d->stride[0] != 1 || d->stride[1] != 1 || d->dil[0] != 1) return pl;
    int o[2];
    hc_out_shape(d, o);
    if (o[0] != d->in[0] || o[1] != d->in[1]) return pl;
    const int KH = d->k[0], KW = d->k[1];
    if (!((KH == 1 && (KW == 1 || KW == 3)) || (KH == 3 && KW == 3))) return pl;
    if (2 * d->pad[1] != d->dil[1] * (KW - 1) || 2 * d->pad[0] != (KH - 1)) return pl;
    if (KW == 1 && d->dil[1] != 1) return pl;
    const int W = d->in[1], H = d->in[0];
    if (W % 32) return pl;
    const int IB = d->Cin / A, OB = d->Cout / A;
    if ((long long)d->N * d->Cin * H * W * 4 >= 0xFFFFFFF0ll || (long long)d->N * d->Cout * H * W * 4 >= 0xFFFFFFF0ll) return pl;
    if (A == 8 ? !(OB % 16 == 0 || (OB % 16 == 8 && OB > 8)) : (OB % 16 != 0)) return pl;
    const int taps = KH * KW, ncol = IB * taps;
    const int dil = KW == 3 ? d->dil[1] : 0, dpad = KW == 3 ? (dil + 3) / 4 * 4 : 0;
    const int xp = 32 + 2 * dpad + 2;
    const int coltiles = (ncol + 15) / 16;
    // waves per workgroup: as many neighbouring column tiles as fit the LDS budget (two workgroups per CU), preferring
    // group sizes that leave no wave idle
    int best_nw = 0;
    size_t best_smem = 0;
    int best_nib = 0;
    double best_score = -1;
    for (int nw = 5; nw >= 2; --nw) {
        int nib = (16 * nw + taps - 1) / taps + 1;
        if (nib > IB) nib = IB;
        const size_t smem = ((size_t)A * 16 * 34 + (size_t)A * nib * KH * xp) * sizeof(float);
        if (smem > 78 * 1024) continue;
        if ((long long)A * nib * KH * ((32 + 2 * dpad) / 4) > 8LL * 64 * nw) continue;      // staging items per thread <= 8
        if ((32 + 2 * dpad) / 4 - 0 > 128 || xp * (long long)A * nib * KH >= (1 << 19)) continue;   // packed item fields
        const int groups = (coltiles + nw - 1) / nw;
        const double use = (double)coltiles / (groups * nw);
        const double score = use + 0.02 * nw;
        if (score > best_score) { best_score = score; best_nw = nw; best_smem = smem; best_nib = nib; }
    }
    if (!best_nw) return pl;
    const int NW = best_nw;
    const int NTH = 64 * NW;
    const int dy_items = A * 16 * 8;
    const int x_items = A * best_nib * KH * ((32 + 2 * dpad) / 4);
    const int DI = (dy_items + NTH - 1) / NTH, XI = (x_items + NTH - 1) / NTH;
    HcqWgP& k = pl.kp;
    k.A = A; k.N = d->N; k.Cin = d->Cin; k.Cout = d->Cout; k.IB = IB; k.OB = OB; k.H = H; k.W = W;
    k.KH = KH; k.KW = KW; k.dil = dil; k.dpad = dpad; k.ncol = ncol; k.xp = xp; k.nib_max = best_nib;
    const int reg_tiles = OB / 16, mix = (A == 8 && OB % 16 == 8) ? 1 : 0;
    k.row_tiles = reg_tiles;
    pl.mix = mix;
    k.nchunks = (long long)d->N * H * W / 32;
    const int groups = (coltiles + NW - 1) / NW;
    const long long tiles = (long long)groups * (reg_tiles + mix) * npair;
    // position splits: about four workgroups per CU in flight, at least 4 chunks per split
    long long splits = (1024 + tiles - 1) / tiles;
    if (splits > k.nchunks / 4) splits = k.nchunks / 4;
    if (splits < 1) splits = 1;
    k.chunks_per_split = (k.nchunks + splits - 1) / splits;
    splits = (k.nchunks + k.chunks_per_split - 1) / k.chunks_per_split;
    pl.grid = dim3((unsigned)splits, (unsigned)groups, (unsigned)(k.row_tiles * npair));
    pl.npair = npair;
    pl.smem = best_smem;
    pl.KH = KH; pl.KW = KW; pl.NW = NW; pl.DI = DI; pl.XI = XI;
    pl.ok = 1;
    return pl;
}

template <int KH, int KW, int NW, int DI, int XI, int KIND>
static int hcq_wgrad_launch_kind(const HcqWgPlan& pl, const HcqWgP& kp, dim3 grid, hipStream_t st) {
    auto kern = hcq_wgrad_kernel<KH, KW, NW, DI, XI, KIND>;
    if (pl.smem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem) != hipSuccess)
        return SELD_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), pl.smem, st, kp);
    return check_launch();
}

template <int KH, int KW, int NW, int DI, int XI>
static int hcq_wgrad_launch_one(const HcqWgPlan& pl, hipStream_t st) {
    if (pl.kp.A == 4) return hcq_wgrad_launch_kind<KH, KW, NW, DI, XI, 0>(pl, pl.kp, pl.grid, st);
    int rc = SELD_OK;
    if (pl.kp.row_tiles > 0) rc = hcq_wgrad_launch_kind<KH, KW, NW, DI, XI, 1>(pl, pl.kp, pl.grid, st);
    if (rc == SELD_OK && pl.mix) {                    // the mixed row tile: a launch of its own (other program)
        HcqWgP kp = pl.kp;
        kp.row_tiles = 1;
        rc = hcq_wgrad_launch_kind<KH, KW, NW, DI, XI, 2>(pl, kp, dim3(pl.grid.x, pl.grid.y, (unsigned)pl.npair), st);
    }
    return rc;
}

// (DI, XI) buckets per wave count: DI = ceil(A*128 / (64 NW)), XI bounded by the LDS budget
template <int KH, int KW, int NW>
static int hcq_wgrad_launch_nw(const HcqWgPlan& pl, hipStream_t st) {
    constexpr int DIM = (8 * 128 + 64 * NW - 1) / (64 * NW);
    if (pl.DI > DIM) return SELD_EUNSUPPORTED;
    if (pl.XI <= 4) return hcq_wgrad_launch_one<KH, KW, NW, DIM, 4>(pl, st);
    if (pl.XI <= 8) return hcq_wgrad_launch_one<KH, KW, NW, DIM, 8>(pl, st);
    return SELD_EUNSUPPORTED;
}

template <int KH, int KW>
static int hcq_wgrad_launch_k(const HcqWgPlan& pl, hipStream_t st) {
    switch (pl.NW) {
        case 5: return hcq_wgrad_launch_nw<KH, KW, 5>(pl, st);
        case 4: return hcq_wgrad_launch_nw<KH, KW, 4>(pl, st);
        case 3: return hcq_wgrad_launch_nw<KH, KW, 3>(pl, st);
        case 2: return hcq_wgrad_launch_nw<KH, KW, 2>(pl, st);
    }
    return SELD_EUNSUPPORTED;
}

static bool hcq_wgrad_launchable(const HcqWgPlan& pl) {
    const int dim = (8 * 128 + 64 * pl.NW - 1) / (64 * pl.NW);
    return pl.ok && pl.DI <= dim && pl.XI <= 8;
}

}  // namespace seld

using namespace seld;

/* 1 if seld_hcq_wgrad_acc takes (desc, npair), else 0 (use seld_hc_conv_bwd_weight_acc). */
extern "C" int seld_hcq_wgrad_supported(const seld_conv_desc* d, int32_t npair) {
    if (hc_validate(d) != SELD_OK || npair < 1 || npair > 2) return 0;
    return hcq_wgrad_launchable(hcq_wgrad_plan(d, npair)) ? 1 : 0;
}

extern "C" int seld_hcq_wgrad_label(const seld_conv_desc* d, int32_t npair, char* buf, int32_t buflen) {
    if (hc_validate(d) != SELD_OK || !buf || buflen < 64) return SELD_EINVAL;
    const HcqWgPlan pl = hcq_wgrad_plan(d, npair);
    if (!hcq_wgrad_launchable(pl)) return SELD_EUNSUPPORTED;
    const int dim = (8 * 128 + 64 * pl.NW - 1) / (64 * pl.NW);
    snprintf(buf, buflen, "hcq_wgrad_kernel<%d, %d, %d, %d, %d, %d>", pl.KH, pl.KW, pl.NW, dim, pl.XI <= 4 ? 4 : 8,
             pl.kp.A == 4 ? 0 : 1);
    return SELD_OK;
}

/* dwA[c] += weight gradient of conv(x; W_A) given dyA (and, npair == 2, dwB[c] += that of a second convolution of the
 * same input given dyB): the accumulating form of seld_hc_conv_bwd_weight_acc on the fast-product kernel.  No bias. */
extern "C" int seld_hcq_wgrad_acc(const seld_conv_desc* d, int32_t npair, const float* x, const float* dyA, const float* dyB,
                                  float* const dwA[8], float* const dwB[8], void* stream) {
    if (hc_validate(d) != SELD_OK || !x || !dyA || !dwA || npair < 1 || npair > 2) return SELD_EINVAL;
    if (npair == 2 && (!dyB || !dwB)) return SELD_EINVAL;
    HcqWgPlan pl = hcq_wgrad_plan(d, npair);
    if (!hcq_wgrad_launchable(pl)) return SELD_EUNSUPPORTED;
    pl.kp.x = x;
    pl.kp.dy[0] = dyA;
    pl.kp.dy[1] = npair == 2 ? dyB : nullptr;
    for (int i = 0; i < 8; ++i) {
        pl.kp.dw[0][i] = i < d->algebra ? dwA[i] : nullptr;
        pl.kp.dw[1][i] = (npair == 2 && i < d->algebra) ? dwB[i] : nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    if (pl.KH == 1 && pl.KW == 1) return hcq_wgrad_launch_k<1, 1>(pl, st);
    if (pl.KH == 1 && pl.KW == 3) return hcq_wgrad_launch_k<1, 3>(pl, st);
    return hcq_wgrad_launch_k<3, 3>(pl, st);
}
