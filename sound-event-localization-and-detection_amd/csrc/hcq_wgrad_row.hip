// Dual-quaternion weight gradient by the 8-multiplication Hamilton product, on the row-chunk GEMM of hc_wgrad_row.hip.
//
//     dQ  = dy_p (x) conj(x_p) + dy_d (x) conj(x_d)        dQ2 = dy_d (x) conj(x_p)
//
// (dual_quaternion_ops.py:122-153 differentiated; quaternion_ops.py:131-147 for the product).  Each (x) is a Hamilton
// product of matrix-valued components and has rank 8: with F_m(a) / G_m(b) the sums of two components listed in
// hcq_conv.hip, P_m = F_m(dy) G_m(conj x)^T are 8 independent real GEMMs over the positions, and the four component
// gradients are signed sums of the P_m.  24 block products instead of the 48 of the block-matrix kernel.
//
// The block-matrix kernel's structure is kept -- a step is 32 positions inside one output row, operands staged with
// 16-byte buffer loads through loop-invariant offsets, LDS images [k-group][row][4 positions], one ds_read_b128 feeds
// four MFMAs -- but the staged rows are FORMS: a loader thread fetches the two component rows of its form and stores
// their signed sum, so the MFMA loop is exactly the plain GEMM loop.  A workgroup owns one form m and
//   family A: rows [F_m(dy_p); F_m(dy_d)] x columns G_m(x_p)   -> P_m of dQ (primal part) and of dQ2 in ONE tile,
//   family B: rows  F_m(dy_d)             x columns G_m(x_d)   -> P_m of dQ (dual part),
// i.e. tiles of 2*OA (or OA) rows x up to 96 columns whose operands are all useful (the block-matrix tiles of the
// TCN layers are 64 x 64 with a quarter of them structurally zero).  The waves of a workgroup split the 32 positions
// of a step in two and the column tiles in two.  Partial P_m go to a small fp32 workspace with float atomics (a third of
// the block-matrix kernel's atomic count); hcq_wr_fold_kernel recombines the eight forms, adds the result to the
// gradient slots and hands the workspace back zeroed.
#include <string.h>
#include "hc_common.h"

namespace seld {

typedef unsigned int uintx4r __attribute__((ext_vector_type(4)));

struct HcqWrP {
    const float* x;
    const float* dy[2];          // pair launch: blockIdx.y selects the gradient
    float* pws;                  // [slot][form][3*OA rows][colpad]
    int IB, OA, Cin, Cout;
    int inH, inW, outH, outW, inS, outS;
    int ph, pw, dh, dw;
    int ncol, colpad, ncg;       // IB*KK, ncg*COLS, column groups
    long long Ptot, split_len;
};

// The two tile families of one launch (blockIdx.z / (8 * ncg)):
//   A: rows [F(dy_p); F(dy_d)] x columns G(x_p) -> workspace rows [0, 2*OA)      B: rows F(dy_d) x columns G(x_d) -> [2*OA, 3*OA)
struct HcqWrFam {
    int nrows;                   // valid rows of the tile (2*OA or OA)
    int row_half[2];             // source half (0 primal, 1 dual) of rows [0, OA) / [OA, 2*OA)
    int col_half;                // source half of the columns
    int row_base;                // first row in the workspace's 3*OA rows
    int split_mul;               // positions per workgroup = split_mul * split_len
};

// F_m(a) = a[c1] + s2 * a[c2]   (first sign always +)
__device__ __forceinline__ void f_form(int m, int* c1, int* c2, float* s2) {
    const int C1[8] = {3, 0, 0, 3, 3, 1, 0, 3}, C2[8] = {1, 2, 2, 1, 2, 0, 1, 2};
    const float S2[8] = {1.f, -1.f, 1.f, -1.f, -1.f, 1.f, -1.f, 1.f};
    *c1 = C1[m]; *c2 = C2[m]; *s2 = S2[m];
}
// G_m(conj x) = s1 * x[c1] + s2 * x[c2]
__device__ __forceinline__ void g_form(int m, int* c1, int* c2, float* s1, float* s2) {
    const int C1[8] = {1, 0, 0, 2, 3, 0, 2, 1}, C2[8] = {2, 3, 3, 1, 2, 1, 3, 0};
    const float S1[8] = {-1.f, 1.f, 1.f, 1.f, 1.f, 1.f, -1.f, -1.f}, S2[8] = {-1.f, -1.f, 1.f, -1.f, -1.f, -1.f, -1.f, -1.f};
    *c1 = C1[m]; *c2 = C2[m]; *s1 = S1[m]; *s2 = S2[m];
}

// RT row tiles (all of them in every wave), CTW column tiles per wave (the workgroup has 2 * CTW)
template <int RT, int CTW>
struct HcqWrSmem {
    static constexpr int AR = (RT * 16 + 31) / 32, BR = (2 * CTW * 16 + 31) / 32;
    static constexpr int APITCH = AR * 32 + 2, BPITCH = BR * 32 + 2;            // rows per k-group (+2: bank spread)
    static constexpr int A_FLOATS = 2 * 8 * APITCH * 4, B_FLOATS = 2 * 8 * BPITCH * 4;
};

template <int RT, int CTW, int KH_T, int KW_T>
__device__ __forceinline__ void hcq_wr_body(const HcqWrP& p, const HcqWrFam& fam, int zt, float* smem) {
    constexpr int ROWS = RT * 16, COLS = 2 * CTW * 16;
    using SM = HcqWrSmem<RT, CTW>;
    constexpr int AR = SM::AR, BR = SM::BR;
    constexpr int KK = KH_T * KW_T;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    float* const As = smem;                              // forms of dy   [buf][k-group][row][4 positions]
    float* const Bs = smem + SM::A_FLOATS;               // forms of xcol [buf][k-group][col][4 positions]
    auto a_at = [&](int buf, int kg, int row) { return As + ((buf * 8 + kg) * SM::APITCH + row) * 4; };
    auto b_at = [&](int buf, int kg, int col) { return Bs + ((buf * 8 + kg) * SM::BPITCH + col) * 4; };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh_w = wave & 1;               // which half of a step's 32 positions this wave multiplies
    const int cg_w = wave >> 1;              // which half of the workgroup's column tiles
    const int m = zt & 7;                    // the form
    const int cg = zt >> 3;                  // column group
    const float* const dyz = p.dy[blockIdx.y];

    int ca1, ca2, cb1, cb2;
    float sa2, sb1, sb2;
    f_form(m, &ca1, &ca2, &sa2);
    g_form(m, &cb1, &cb2, &sb1, &sb2);

    // family B's tile has half the rows: it takes twice the positions per workgroup (half as many workgroups), so that
    // all workgroups of the launch -- one resident generation -- finish together
    const long long slen = p.split_len * fam.split_mul;
    const long long pbeg = (long long)blockIdx.x * slen;
    if (pbeg >= p.Ptot) return;
    long long pend = pbeg + slen;
    if (pend > p.Ptot) pend = p.Ptot;
    const int nchunks = pbeg < pend ? (int)((pend - pbeg) >> 5) : 0;

    const int g = tid & 7;                   // 4-position group inside the 32-position step
    const int rsub = tid >> 3;               // 0..31

    unsigned a_v1[AR], a_v2[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int r = rsub + 32 * j;
        const bool ok = r < ROWS && r < fam.nrows;
        const int seg = (ok && r >= p.OA) ? 1 : 0;
        const int o = ok ? r - seg * p.OA : 0;
        const int half = fam.row_half[seg];
        a_v1[j] = ok ? (unsigned)((((half * 4 + ca1) * p.OA + o) * p.outS + 4 * g) * 4) : OOB;
        a_v2[j] = ok ? (unsigned)((((half * 4 + ca2) * p.OA + o) * p.outS + 4 * g) * 4) : OOB;
    }
    unsigned b_v1[BR], b_v2[BR];
    int b_row[BR], b_col[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int c = rsub + 32 * j;
        const int col = cg * COLS + c;
        const bool ok = c < COLS && col < p.ncol;
        const int colc = ok ? col : 0;
        const int ib = colc / KK;
        const int tap = colc - ib * KK;
        const int kh = tap / KW_T, kw = tap - kh * KW_T;
        const unsigned inner = (unsigned)(kh * p.dh * p.inW + kw * p.dw + 4 * g);
        b_v1[j] = ok ? (unsigned)((((fam.col_half * 4 + cb1) * p.IB + ib) * p.inS + inner) * 4) : OOB;
        b_v2[j] = ok ? (unsigned)((((fam.col_half * 4 + cb2) * p.IB + ib) * p.inS + inner) * 4) : OOB;
        b_row[j] = kh * p.dh - p.ph;
        b_col[j] = kw * p.dw - p.pw + 4 * g;
    }
    const bool rows_trivial = (KH_T == 1) && (p.ph == 0);

    int t_img, t_oh, t_ow;
    {
        const long long im = pbeg / p.outS;
        const int rem = (int)(pbeg - im * p.outS);
        t_img = (int)im;
        t_oh = rem / p.outW;
        t_ow = rem - t_oh * p.outW;
    }
    const long long dy_img = (long long)p.Cout * p.outS;
    const long long x_img = (long long)p.Cin * p.inS;
    const unsigned nrec_a = (unsigned)(dy_img * 4 > (long long)OOB ? (long long)OOB : dy_img * 4);
    const long long xb = (x_img + (long long)(KH_T * p.dh + p.ph + 1) * p.inW + KW_T * p.dw + 64) * 4;
    const unsigned nrec_b = (unsigned)(xb > (long long)OOB ? (long long)OOB : xb);
    const int wspan = (KW_T - 1) * p.dw - p.pw;

    floatx4 ar[AR], br[BR];
    auto ldx4 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) __attribute__((always_inline)) {
        const uintx4r v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        return (floatx4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    };
    auto load_chunk = [&](bool advance) __attribute__((always_inline)) {
        const float* abase = dyz + (long long)t_img * dy_img + (long long)t_oh * p.outW + t_ow;
        const float* bbase = p.x + (long long)t_img * x_img + (long long)(t_oh - p.ph) * p.inW + (t_ow - p.pw);
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, nrec_a, 0x00020000);
        const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)bbase, 0, nrec_b, 0x00020000);
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const floatx4 v1 = ldx4(arsrc, a_v1[j]), v2 = ldx4(arsrc, a_v2[j]);
            ar[j] = v1 + sa2 * v2;
        }
        const bool interior = (t_ow - p.pw >= 0) && (t_ow + 31 + wspan < p.inW);      // scalar
        if (interior) {
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                unsigned o1 = b_v1[j], o2 = b_v2[j];
                if (!rows_trivial && !((unsigned)(t_oh + b_row[j]) < (unsigned)p.inH)) { o1 = OOB; o2 = OOB; }
                const floatx4 v1 = ldx4(brsrc, o1), v2 = ldx4(brsrc, o2);
                br[j] = sb1 * v1 + sb2 * v2;
            }
        } else {
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                const bool rowok = (unsigned)(t_oh + b_row[j]) < (unsigned)p.inH;
                const int iw = t_ow + b_col[j];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bool ok = rowok && (unsigned)(iw + s) < (unsigned)p.inW && b_v1[j] != OOB;
                    const float v1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brsrc, ok ? b_v1[j] + 4u * s : OOB, 0, 0));
                    const float v2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brsrc, ok ? b_v2[j] + 4u * s : OOB, 0, 0));
                    br[j][s] = sb1 * v1 + sb2 * v2;
                }
            }
        }
        if (advance) {
            t_ow += 32;
            if (t_ow >= p.outW) {
                t_ow = 0;
                if (++t_oh >= p.outH) { t_oh = 0; ++t_img; }
            }
        }
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < AR; ++j) *reinterpret_cast<floatx4*>(a_at(buf, g, rsub + 32 * j)) = ar[j];
#pragma unroll
        for (int j = 0; j < BR; ++j) *reinterpret_cast<floatx4*>(b_at(buf, g, rsub + 32 * j)) = br[j];
    };

    floatx4 acc[RT][CTW];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CTW; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fk = lane >> 4;
    if (nchunks > 0) { load_chunk(nchunks > 1); store_chunk(0); }
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        load_chunk(chunk + 2 < nchunks);
        __builtin_amdgcn_sched_barrier(0);
        floatx4 av[RT], bv[CTW];
#pragma unroll
        for (int i = 0; i < RT; ++i) av[i] = *reinterpret_cast<const floatx4*>(a_at(buf, kh_w * 4 + fk, i * 16 + fr));
#pragma unroll
        for (int j = 0; j < CTW; ++j) bv[j] = *reinterpret_cast<const floatx4*>(b_at(buf, kh_w * 4 + fk, (cg_w * CTW + j) * 16 + fr));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s == 3) __builtin_amdgcn_sched_barrier(0);      // the LDS stores may mix with the last k-step only
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CTW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
        }
        store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- partial P_m into the workspace ------------------------------------------------------------------------------
    float* const pw = p.pws + ((size_t)(blockIdx.y * 8 + m) * (3 * p.OA) + fam.row_base) * p.colpad;
#pragma unroll
    for (int j = 0; j < CTW; ++j) {
        const int col = cg * COLS + (cg_w * CTW + j) * 16 + fr;
        if (col >= p.ncol) continue;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + fk * 4 + r;
                if (row < fam.nrows) atomicAdd(pw + (size_t)row * p.colpad + col, acc[i][j][r]);
            }
    }
}

// One launch, both tile families: blockIdx.z = family * (8 * ncg) + column group * 8 + form.  Family A's tile has RTA
// row tiles, family B's half of them (rounded up).
template <int RTA, int CTW, int KH_T, int KW_T>
__global__ __launch_bounds__(256) void hcq_wgrad_row_kernel(const HcqWrP p, const HcqWrFam famA, const HcqWrFam famB) {
    constexpr int RTB = (RTA + 1) / 2;
    using SMA = HcqWrSmem<RTA, CTW>;
    __shared__ __attribute__((aligned(16))) float smem[SMA::A_FLOATS + SMA::B_FLOATS];
    const int per = 8 * p.ncg;
    if ((int)blockIdx.z < per) hcq_wr_body<RTA, CTW, KH_T, KW_T>(p, famA, (int)blockIdx.z, smem);
    else hcq_wr_body<RTB, CTW, KH_T, KW_T>(p, famB, (int)blockIdx.z - per, smem);
}

struct HcqWrFoldP {
    float* pws;
    float* dw[2][8];
    int OA, ncol, colpad, nslots;
};

// one thread per (slot, Q / Q2, output block channel, column): the eight forms -> four component gradients
__global__ __launch_bounds__(256) void hcq_wr_fold_kernel(const HcqWrFoldP f) {
    const int per = f.OA * f.ncol;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= f.nslots * 2 * per) return;
    const int slot = idx / (2 * per);
    int rem = idx - slot * 2 * per;
    const int set = rem / per;
    rem -= set * per;
    const int o = rem / f.ncol, col = rem - o * f.ncol;
    float P[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        float* base = f.pws + (size_t)(slot * 8 + m) * (3 * f.OA) * f.colpad + col;
        if (set == 0) {
            float* a = base + (size_t)o * f.colpad;
            float* b = base + (size_t)(2 * f.OA + o) * f.colpad;
            P[m] = *a + *b;
            *a = 0.f;
            *b = 0.f;
        } else {
            float* a = base + (size_t)(f.OA + o) * f.colpad;
            P[m] = *a;
            *a = 0.f;
        }
    }
    const float h0 = 0.5f * P[0], h1 = 0.5f * P[1], h2 = 0.5f * P[2], h3 = 0.5f * P[3];
    const float c[4] = {(h3 - h0) + (h1 + h2) + P[4], (h3 - h0) - (h1 + h2) + P[5], (h3 + h0) + (h2 - h1) + P[6],
                        (h3 + h0) + (h1 - h2) - P[7]};
#pragma unroll
    for (int q = 0; q < 4; ++q) f.dw[slot][set * 4 + q][(size_t)o * f.ncol + col] += c[q];
}

// ---------------------------------------------------------------------------------------------------------------------
struct HcqWrPlan {
    int ok;
    int KH, KW, RTA, RTB, CTW;
    HcqWrP kp;
    int nsplit;
    size_t ws_bytes;
};

static long long hcq_wr_slots() { return env().wgrad_wgs ? env().wgrad_wgs : 768; }

static HcqWrPlan hcq_wr_plan(const seld_conv_desc* d, int npair) {
    HcqWrPlan pl{};
    if (!env().hcq_wgrad_row || env().conv_no_hcq) return pl;    // opt-in (SELD_HCQ_WGRAD_ROW=1): not yet faster, see DESIGN 4a
    if (d->algebra != 8 || npair < 1 || npair > 2) return pl;
    if (d->stride[0] != 1 || d->stride[1] != 1) return pl;
    const int KH = d->k[0], KW = d->k[1];
    if (!((KH == 1 && (KW == 1 || KW == 3)) || (KH == 3 && KW == 3))) return pl;
    int o[2];
    hc_out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0 || o[1] % 32) return pl;
    const int OA = d->Cout / 8, IB = d->Cin / 8;
    if (OA != 24 && OA != 48) return pl;                       // row tiles of 2*OA and OA rows: 3 + 2 or 6 + 3
    const int outS = o[0] * o[1], inS = d->in[0] * d->in[1];
    if ((long long)d->Cout * outS >= (1LL << 29) || (long long)d->Cin * inS >= (1LL << 29)) return pl;
    const long long Ptot = (long long)d->N * outS;
    if (Ptot < 4096) return pl;                                // tiny problems: the block-matrix kernels
    const int ncol = IB * KH * KW;
    if (ncol < 32) return pl;                                  // the first layers (9 / 18 columns) stay on the fused kernel
    // column groups of 64 or 96: the less padding wins, ties to the wider tile
    const int g64 = (ncol + 63) / 64, g96 = (ncol + 95) / 96;
    const int CTW = (g96 * 96 <= g64 * 64) ? 3 : 2;
    const int COLS = 32 * CTW, ncg = CTW == 3 ? g96 : g64;
    HcqWrP& k = pl.kp;
    k.IB = IB; k.OA = OA; k.Cin = d->Cin; k.Cout = d->Cout;
    k.inH = d->in[0]; k.inW = d->in[1]; k.outH = o[0]; k.outW = o[1]; k.inS = inS; k.outS = outS;
    k.ph = d->pad[0]; k.pw = d->pad[1]; k.dh = d->dil[0]; k.dw = d->dil[1];
    k.ncol = ncol; k.colpad = ncg * COLS; k.ncg = ncg;
    k.Ptot = Ptot;
    // splits: fill ~768 workgroup slots (2 families x 8 forms x column groups x gradients), >= 256 positions each,
    // a multiple of 8 so that the tiles of one split share an XCD (hc_common.h wgrad_tile)
    const long long tiles = 16LL * ncg * npair;
    long long want = (hcq_wr_slots() + tiles - 1) / tiles;
    const long long maxs = (Ptot + 255) / 256;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want >= 8) want -= want % 8;
    if (want > 1 && (want & 1)) --want;                        // even: family B pairs the splits up
    long long len = (Ptot + want - 1) / want;
    len = (len + 31) / 32 * 32;
    pl.nsplit = (int)((Ptot + len - 1) / len);
    k.split_len = len;
    pl.KH = KH; pl.KW = KW; pl.CTW = CTW;
    pl.RTA = 2 * OA / 16;
    pl.RTB = (OA + 15) / 16;
    pl.ws_bytes = (size_t)npair * 8 * 3 * OA * k.colpad * sizeof(float);
    pl.ok = 1;
    return pl;
}

template <int KH, int KW>
static int hcq_wr_launch(const HcqWrPlan& pl, int npair, hipStream_t st) {
    const HcqWrP& k = pl.kp;
    HcqWrFam a{}, b{};
    a.nrows = 2 * k.OA; a.row_half[0] = 0; a.row_half[1] = 1; a.col_half = 0; a.row_base = 0; a.split_mul = 1;
    b.nrows = k.OA; b.row_half[0] = 1; b.row_half[1] = 1; b.col_half = 1; b.row_base = 2 * k.OA; b.split_mul = 1;      // (2 = equal work per workgroup: measured slower, 152 vs 101 us)
    const dim3 grid((unsigned)pl.nsplit, (unsigned)npair, (unsigned)(16 * k.ncg));
    if (pl.CTW == 3) {
        if (pl.RTA == 6) hipLaunchKernelGGL((hcq_wgrad_row_kernel<6, 3, KH, KW>), grid, dim3(256), 0, st, k, a, b);
        else hipLaunchKernelGGL((hcq_wgrad_row_kernel<3, 3, KH, KW>), grid, dim3(256), 0, st, k, a, b);
    } else {
        if (pl.RTA == 6) hipLaunchKernelGGL((hcq_wgrad_row_kernel<6, 2, KH, KW>), grid, dim3(256), 0, st, k, a, b);
        else hipLaunchKernelGGL((hcq_wgrad_row_kernel<3, 2, KH, KW>), grid, dim3(256), 0, st, k, a, b);
    }
    return check_launch();
}

}  // namespace seld
using namespace seld;

/* Bytes of the fp32 workspace seld_hcq_wgrad_row_acc needs for (desc, npair); 0 = shape not taken (use
 * seld_hc_conv[_pair]_bwd_weight_acc / seld_hcq_wgrad_acc).  The workspace must be ZERO on the first call; every call
 * hands it back zeroed, so one buffer per stream serves every layer. */
extern "C" size_t seld_hcq_wgrad_row_workspace(const seld_conv_desc* d, int32_t npair) {
    if (hc_validate(d) != SELD_OK) return 0;
    const HcqWrPlan pl = hcq_wr_plan(d, npair);
    return pl.ok ? pl.ws_bytes : 0;
}

extern "C" int seld_hcq_wgrad_row_label(const seld_conv_desc* d, int32_t npair, char* buf, int32_t buflen) {
    if (hc_validate(d) != SELD_OK || !buf || buflen < 64) return SELD_EINVAL;
    const HcqWrPlan pl = hcq_wr_plan(d, npair);
    if (!pl.ok) return SELD_EUNSUPPORTED;
    snprintf(buf, buflen, "hcq_wgrad_row_kernel<%d, %d, %d, %d>", pl.RTA, pl.CTW, pl.KH, pl.KW);
    return SELD_OK;
}

/* dwA[c] += weight gradient of the dual-quaternion convolution (x; W_A) given dyA (npair == 2: and dwB[c] += that of a
 * second convolution of the same input given dyB), 24 instead of 48 block products.  Two launches on `stream`: the tiles
 * and the fold. */
extern "C" int seld_hcq_wgrad_row_acc(const seld_conv_desc* d, int32_t npair, const float* x, const float* dyA,
                                      const float* dyB, float* const dwA[8], float* const dwB[8], void* workspace,
                                      size_t workspace_bytes, void* stream) {
    if (hc_validate(d) != SELD_OK || !x || !dyA || !dwA || npair < 1 || npair > 2 || !workspace) return SELD_EINVAL;
    if (npair == 2 && (!dyB || !dwB)) return SELD_EINVAL;
    HcqWrPlan pl = hcq_wr_plan(d, npair);
    if (!pl.ok) return SELD_EUNSUPPORTED;
    if (workspace_bytes < pl.ws_bytes) return SELD_EWORKSPACE;
    pl.kp.x = x;
    pl.kp.dy[0] = dyA;
    pl.kp.dy[1] = npair == 2 ? dyB : dyA;
    pl.kp.pws = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (pl.KH == 1 && pl.KW == 1) rc = hcq_wr_launch<1, 1>(pl, npair, st);
    else if (pl.KH == 1) rc = hcq_wr_launch<1, 3>(pl, npair, st);
    else rc = hcq_wr_launch<3, 3>(pl, npair, st);
    if (rc) return rc;
    HcqWrFoldP f{};
    f.pws = pl.kp.pws; f.OA = pl.kp.OA; f.ncol = pl.kp.ncol; f.colpad = pl.kp.colpad; f.nslots = npair;
    for (int i = 0; i < 8; ++i) {
        f.dw[0][i] = dwA[i];
        f.dw[1][i] = npair == 2 ? dwB[i] : nullptr;
    }
    const int total = npair * 2 * pl.kp.OA * pl.kp.ncol;
    hipLaunchKernelGGL(hcq_wr_fold_kernel, dim3((total + 255) / 256), dim3(256), 0, st, f);
    return check_launch();
}
