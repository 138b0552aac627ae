// Dual-quaternion weight gradients of MANY layers in one persistent launch (gfx950), 24 block products per layer.
//
//     dQ  = dy_p (x) conj(x_p) + dy_d (x) conj(x_d)        dQ2 = dy_d (x) conj(x_p)
//
// (dual_quaternion_ops.py:122-153 differentiated; quaternion_ops.py:131-147 for the Hamilton product (x)).  Each product
// has rank 8 (hcq_conv.hip): P_m = F_m(dy) G_m(conj x)^T, m = 0..7, are independent real GEMMs over the positions and the
// component gradients are signed sums of the P_m.
//
// Why a new kernel: the row-chunk kernels (hc_wgrad_row.hip, hcq_wgrad_row.hip) sit at ~4.4 us per
// 32-position step whatever their MFMA count -- 36-72 MFMAs per wave and step are shorter than the staging round trip,
// and every layer's launch is one resident generation with its own ramp, drain and atomic fold.  Here the shape of the
// problem is turned round:
//
//   * A workgroup (8 waves, ONE PER FORM m) keeps the WHOLE output of a layer in registers -- all of
//     [F_m(dy_p); F_m(dy_d)] x [G_m(x_p) | G_m(x_d)] minus the structurally zero block: 42 16x16 tiles per wave for the
//     TCN 1x3 layers -- so a step of 16 positions is 168 MFMAs per wave (4.5 us per workgroup) against 70 KB of RAW
//     operands: 5-10 bytes per cycle and CU, which L2 / the Infinity Cache deliver without effort.
//   * The raw component rows go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, no VGPR hop, no ds_write), two
//     stages, one barrier per step; the loads of step s+1 are issued a whole step (thousands of cycles) before their
//     vmcnt(0).  Every wave forms ITS sums F_m / G_m at LDS -> register time (one fma per operand dword).
//   * Such a tile only pays when a workgroup stays on it for hundreds of positions, and the reduction length of one layer
//     (16384 positions) spread over 256 CUs is 64: so the layers are GROUPED.  The backward pass defers the weight
//     gradients of all residual blocks (hip_ops.DeferredWgrads) and hands them over as one job list; the concatenated
//     step space is dealt evenly to the persistent workgroups, a workgroup flushes its tile when it crosses into the next
//     job.  Flushes are plain 16-byte stores of accumulator fragments into a scratch slot (wg + job): no atomics.
//   * hcq_gw_sum_kernel adds a job's slots IN INDEX ORDER, hcq_gw_fold_kernel recombines the eight forms into the
//     component gradients and adds them to the gradient slots.  The result is run-to-run reproducible.
//
// Geometry per job ("sub-job"): a 1 x KW slice of a 'same' stride-1 convolution whose rows are W = 16k positions long;
// a 3x3 layer is three sub-jobs (one per kernel row, the input row shifted), a wide 1x3 layer three single-tap ones.
#include <string.h>
#include <type_traits>
#include <vector>
#include "hc_common.h"

namespace seld {

typedef int int4v __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

constexpr int GW_MAXJ = 56;
constexpr unsigned GW_OOB = 0xFFFFFFF0u;
constexpr int GW_BIAS = 64;            // the x descriptor is based 64 positions before the step's first output column

struct GwJob {
    const float* x;
    const float* dy;
    int H;                 // rows per image (output = input: 'same', stride 1)
    int woff0, wstep;      // input column of tap t = output column + woff0 + t * wstep
    int hoff;              // input row = output row + hoff
    int step0;             // first step of this job in the launch's concatenated step space
    int W;                 // row length (a multiple of 16)
};

struct GwP {
    GwJob job[GW_MAXJ];
    float* part;           // [nwg + njobs][8 forms][NT tiles][256]
    int njobs, per;        // per = steps per workgroup
    int total;             // all steps
};

// The forms (hcq_conv.hip; the tables of hcq_wgrad_row.hip), one nibble / bit per m:
//   F_m(a)      = a[c1] + s2 * a[c2]          c1 = {3,0,0,3,3,1,0,3}  c2 = {1,2,2,1,2,0,1,2}  s2 = {+,-,+,-,-,+,-,+}
//   G_m(conj x) = s1 * (x[c1] + t * x[c2])    c1 = {1,0,0,2,3,0,2,1}  c2 = {2,3,3,1,2,1,3,0}
//                 s1 = {-,+,+,+,+,+,-,-}, s2 = {-,-,+,-,-,-,-,-}, t = s2 * s1 = {+,-,+,-,-,-,+,+}; s1 is applied by the fold
constexpr unsigned GW_FC1 = 0x30133003u, GW_FC2 = 0x21021221u, GW_FS2 = 0x5Au;
constexpr unsigned GW_GC1 = 0x12032001u, GW_GC2 = 0x03121332u, GW_GT = 0x3Au, GW_GS1 = 0xC1u;
constexpr bool gw_nibbles_are(unsigned v, int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7) {
    return ((v >> 0) & 15) == (unsigned)a0 && ((v >> 4) & 15) == (unsigned)a1 && ((v >> 8) & 15) == (unsigned)a2 &&
           ((v >> 12) & 15) == (unsigned)a3 && ((v >> 16) & 15) == (unsigned)a4 && ((v >> 20) & 15) == (unsigned)a5 &&
           ((v >> 24) & 15) == (unsigned)a6 && ((v >> 28) & 15) == (unsigned)a7;
}
static_assert(gw_nibbles_are(GW_FC1, 3, 0, 0, 3, 3, 1, 0, 3) && gw_nibbles_are(GW_FC2, 1, 2, 2, 1, 2, 0, 1, 2), "F_m components");
static_assert(gw_nibbles_are(GW_GC1, 1, 0, 0, 2, 3, 0, 2, 1) && gw_nibbles_are(GW_GC2, 2, 3, 3, 1, 2, 1, 3, 0), "G_m components");
static_assert(GW_FS2 == ((1u << 1) | (1u << 3) | (1u << 4) | (1u << 6)), "F_m second sign negative at m = 1, 3, 4, 6");
static_assert(GW_GS1 == ((1u << 0) | (1u << 6) | (1u << 7)), "G_m first sign negative at m = 0, 6, 7");
static_assert(GW_GT == ((1u << 1) | (1u << 3) | (1u << 4) | (1u << 5)), "G_m sign ratio negative at m = 1, 3, 4, 5");
__device__ __forceinline__ void gw_f_form(int m, int* c1, int* c2, float* s2) {
    *c1 = (GW_FC1 >> (4 * m)) & 3;
    *c2 = (GW_FC2 >> (4 * m)) & 3;
    *s2 = ((GW_FS2 >> m) & 1) ? -1.f : 1.f;
}
__device__ __forceinline__ void gw_g_form(int m, int* c1, int* c2, float* t) {
    *c1 = (GW_GC1 >> (4 * m)) & 3;
    *c2 = (GW_GC2 >> (4 * m)) & 3;
    *t = ((GW_GT >> m) & 1) ? -1.f : 1.f;
}
__host__ __device__ __forceinline__ float gw_g_sign(int m) { return ((GW_GS1 >> m) & 1) ? -1.f : 1.f; }

// Two accumulator layouts per form m:
//   SPLIT (OA not a multiple of 16, e.g. 24): rows [F(dy_p); F(dy_d)] (2*OA, the middle tile mixes the halves) x columns
//         [G(x_p) | G(x_d)] -- every 16x16 tile is full; tiles of primal rows x dual columns (the structural zero block) are
//         not computed.  The fold adds the (p, p) and (d, d) blocks into dQ and takes (d, p) as dQ2.
//   COMB  (OA a multiple of 16): dQ and dQ2 are accumulated directly,
//             accQ [o][col] += F(dy_p) G(x_p)^T + F(dy_d) G(x_d)^T        accQ2[o][col] += F(dy_d) G(x_p)^T
//         with row tiles of o and column tiles of (tap, ib) -- a third fewer accumulators (120 instead of 168 registers for
//         the 192 -> 384 1x3 layers, which is what lets them fit two waves per SIMD) for a last column tile that is half
//         empty when IB*KW is not a multiple of 16 (45 instead of 42 MFMAs per k-step there).
template <int OA, int IB, int KW, int XP>
struct GwShape {
    static constexpr bool COMB = (OA % 16) == 0;
    static constexpr int RT = COMB ? OA / 16 : 2 * OA / 16;
    static constexpr int CT = COMB ? (IB * KW + 15) / 16 : 2 * IB * KW / 16;
    static constexpr int NPR = OA / 16;                    // SPLIT: row tiles that hold primal rows only: they skip ...
    static constexpr int CTP = (IB * KW + 15) / 16;        // ... the column tiles behind the first CTP (dual columns only)
    static constexpr int NT = COMB ? 2 * RT * CT : NPR * CTP + (RT - NPR) * CT;
    static constexpr int XC = XP / 4;                      // 16-byte chunks per staged x row
    static constexpr int DYB = 8 * OA * 64;                // bytes of one stage's dy image   [8*OA rows][16]
    static constexpr int XTB = 8 * IB * XP * 4;            // bytes of one tap's x image      [8*IB rows][XP]
    static constexpr int XB = KW * XTB;
    static constexpr int NDYI = DYB / 1024, XTI = XTB / 1024, NINSTR = NDYI + KW * XTI;
    static constexpr int LDS_BYTES = 2 * (DYB + XB);
    static_assert((2 * OA) % 16 == 0 && (2 * IB * KW) % 16 == 0, "tile structure");
    static_assert(DYB % 1024 == 0 && XTB % 1024 == 0, "an LDS-DMA instruction fills 1 KiB of one image");
    static_assert(LDS_BYTES <= 160 * 1024, "two stages must fit the CU's LDS");
    // SPLIT
    __host__ __device__ static constexpr int tile(int i, int j) { return i < NPR ? i * CTP + j : NPR * CTP + (i - NPR) * CT + j; }
    __host__ __device__ static constexpr bool active(int i, int j) { return i >= NPR || j < CTP; }
    // COMB: set 0 = dQ, 1 = dQ2
    __host__ __device__ static constexpr int tileq(int set, int i, int j) { return (set * RT + i) * CT + j; }
};

// One 16-byte-per-lane LDS-DMA: LDS[lds_addr + 16 * lane ..] <- buffer[voff ..] (zero when voff is out of range).  Inline
// asm: the compiler's waitcnt pass would otherwise put vmcnt(0) in front of every LDS read that follows (it cannot tell
// the two stages apart); the kernel counts these loads itself (gw_wait_dma).  M0 is saved and restored in the statement.
__device__ __forceinline__ void gw_dma16(unsigned lds_addr, unsigned voff, int4v rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void gw_wait_dma_and_barrier() {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ int4v gw_rsrc(const float* base) {
    const unsigned long long a = (unsigned long long)base;
    return (int4v){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)0x80000000u, 0x00020000};
}

template <int OA, int IB, int KW, int XP>
__global__ __launch_bounds__(512, 2) void hcq_wgrad_grp_kernel(const GwP p) {
    using S = GwShape<OA, IB, KW, XP>;
    constexpr int RT = S::RT, CT = S::CT, NT = S::NT;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[S::LDS_BYTES];   // [dy stage 0][dy stage 1][x stage 0][x stage 1]
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int m = wave;                                    // this wave's form

    int ca1, ca2, cb1, cb2;
    float sa2, tb;
    gw_f_form(m, &ca1, &ca2, &sa2);
    gw_g_form(m, &cb1, &cb2, &tb);

    // ---- this workgroup's range of the concatenated step space -------------------------------------------------------
    const int g0 = (int)blockIdx.x * p.per;
    int g1 = g0 + p.per;
    if (g1 > p.total) g1 = p.total;
    if (g0 >= g1) return;
    int jb = 0;
    while (jb + 1 < p.njobs && p.job[jb + 1].step0 <= g0) ++jb;

    // ---- lane constants of the fragment reads ------------------------------------------------------------------------
    // A step's 16 positions are multiplied in two halves of 8 (two MFMA k-steps each): in half u, lane (fr, fk) supplies
    // positions 8*u + 2*fk + {0, 1} of row / column 16*tile + fr -- 8-byte reads, half the operand registers of a
    // 16-position pass (the 42-tile layers have 168 accumulators per lane).
    // dy image: row (half*4 + c)*OA + o, 64 bytes per row
    // (OA a multiple of 16: no row tile mixes primal and dual rows, the tiles differ by compile-time offsets from ONE lane
    // register; otherwise one register per tile)
    constexpr bool COMB = S::COMB;
    constexpr int NAR = COMB ? 1 : RT;
    unsigned arow_[NAR];
#pragma unroll
    for (int i = 0; i < NAR; ++i) {
        const int r = 16 * i + fr;
        const int half = r >= OA ? 1 : 0;
        arow_[i] = (unsigned)(((half * 4 * OA + (r - half * OA)) * 16 + 2 * fk) * 4);
    }
    // SPLIT: row tile i of [dy_p; dy_d];  COMB: row tile i of the half `half`
    auto arow = [&](int i, int half) __attribute__((always_inline)) -> unsigned {
        if constexpr (COMB) return arow_[0] + (unsigned)((half * 4 * OA + 16 * i) * 64);
        else return arow_[i];
    };
    const unsigned a_c1 = (unsigned)(ca1 * OA * 64), a_c2 = (unsigned)(ca2 * OA * 64);
    const unsigned b_c1 = (unsigned)(cb1 * IB * XP * 4), b_c2 = (unsigned)(cb2 * IB * XP * 4);

    for (int g = g0; g < g1;) {
        const GwJob& J = p.job[jb];
        const int jend = (jb + 1 < p.njobs) ? p.job[jb + 1].step0 : p.total;
        const int s_end = (g1 < jend ? g1 : jend) - J.step0;          // steps of this job are [s_beg, s_end)
        const int s_beg = g - J.step0;
        const int H = J.H, W = J.W;
        const int spr = W >> 4;                               // steps per row
        const long long dy_img = (long long)(8 * OA) * H * W, x_img = (long long)(8 * IB) * H * W;
        const unsigned dy_rs = (unsigned)(H * W * 4), x_rs = dy_rs;        // bytes between channel rows

        // x image of tap t: window of XP floats starting at column  w0 + al[t],  al[t] = floor4(woff0 + t*wstep);  the
        // fragment reads start ofs[t] = (woff0 + t*wstep) - al[t] floats into it.  Column (half, tap, ib) of lane fr:
        // (COMB: the CT column tiles of the primal half; the dual half lies 4*IB image rows further; columns past IB*KW in
        // the last tile read column 0 -- their products land in output columns the fold never looks at)
        unsigned bcol[CT];
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            int c = 16 * j + fr;
            if (COMB && c >= IB * KW) c = 0;
            const int half = c >= IB * KW ? 1 : 0;
            const int cc = c - half * IB * KW;
            const int t = cc / IB, ib = cc - t * IB;
            const int wo = J.woff0 + t * J.wstep;
            const int ofs = wo & 3;
            bcol[j] = (unsigned)((((t * 8 + half * 4) * IB + ib) * XP + ofs + 2 * fk) * 4);
        }

        floatx4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (floatx4){0.f, 0.f, 0.f, 0.f};

        // wave-uniform tracker of the step to LOAD next
        int l_wi, l_h, l_n;
        {
            const int row = s_beg / spr;
            l_wi = s_beg - row * spr;
            l_n = row / H;
            l_h = row - l_n * H;
        }
        auto issue = [&](auto stage_c) __attribute__((always_inline)) {
            constexpr int STG = decltype(stage_c)::value;
            const int w0 = l_wi << 4;
            const float* dyb = J.dy + (long long)l_n * dy_img + (long long)l_h * W + w0;
            const int hi = l_h + J.hoff;
            const bool rowok = (unsigned)hi < (unsigned)H;
            const float* xb = J.x + (long long)l_n * x_img + (long long)hi * W + (w0 - GW_BIAS);
            const int4v dyr = gw_rsrc(dyb), xr = gw_rsrc(xb);
            // the per-lane offsets below are loop-invariant, and hoisted out of the step loop they would cost ~30 registers
            // this kernel does not have: recomputed per step (a dozen VALU per DMA) from a lane id the compiler cannot see through
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int k = 0; k < (S::NINSTR + 7) / 8; ++k) {
                const int q = k * 8 + wave;                           // instruction index, wave-uniform
                if (q < S::NINSTR) {
                    if (q < S::NDYI) {
                        const int c = q * 64 + ln;                    // chunk: row c >> 2, 16-byte piece c & 3
                        const unsigned voff = (unsigned)(c >> 2) * dy_rs + (unsigned)(c & 3) * 16u;
                        gw_dma16(lds0 + STG * S::DYB + q * 1024, voff, dyr);
                    } else {
                        const int qx = q - S::NDYI;
                        const int t = qx / S::XTI;                    // tap, wave-uniform (XTI instructions per tap image)
                        const int rr = (qx - t * S::XTI) * 64 + ln;
                        const int row = rr / S::XC, ch = rr - row * S::XC;
                        const int wo = J.woff0 + t * J.wstep;
                        const int rel = (wo & ~3) + 4 * ch;           // first column of the piece, relative to w0
                        const bool ok = rowok && (unsigned)(w0 + rel) <= (unsigned)(W - 4);
                        const unsigned voff = ok ? (unsigned)row * x_rs + (unsigned)((GW_BIAS + rel) * 4) : GW_OOB;
                        gw_dma16(lds0 + 2 * S::DYB + STG * S::XB + qx * 1024, voff, xr);
                    }
                }
            }
            if (++l_wi == spr) { l_wi = 0; if (++l_h == H) { l_h = 0; ++l_n; } }
        };
        auto compute = [&](auto stage_c) __attribute__((always_inline)) {
            constexpr int STG = decltype(stage_c)::value;
            const unsigned char* dyi = smem + STG * S::DYB;
            const unsigned char* xi = smem + 2 * S::DYB + STG * S::XB;
            if constexpr (!COMB) {
                // Column tiles in groups of GC: the raw reads + sums of group g+1 are issued in front of group g's MFMAs (one
                // scheduling region), so a wave holds 2 * GC column operands, not CT of them.
                constexpr int GC = 3, NG = CT / GC;
                static_assert(COMB || CT % GC == 0, "column tiles come in groups of three");
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    floatx2 F[RT], G[2][GC];
                    auto read_g = [&](auto grp_c) __attribute__((always_inline)) {
                        constexpr int GRP = decltype(grp_c)::value;
#pragma unroll
                        for (int jj = 0; jj < GC; ++jj) {
                            const float* q1 = reinterpret_cast<const float*>(xi + bcol[GRP * GC + jj] + b_c1 + u * 32);
                            const float* q2 = reinterpret_cast<const float*>(xi + bcol[GRP * GC + jj] + b_c2 + u * 32);
                            G[GRP & 1][jj][0] = q1[0] + tb * q2[0];
                            G[GRP & 1][jj][1] = q1[1] + tb * q2[1];
                        }
                    };
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        const floatx2 v1 = *reinterpret_cast<const floatx2*>(dyi + arow(i, 0) + a_c1 + u * 32);
                        const floatx2 v2 = *reinterpret_cast<const floatx2*>(dyi + arow(i, 0) + a_c2 + u * 32);
                        F[i] = v1 + sa2 * v2;
                    }
                    read_g(std::integral_constant<int, 0>{});
                    __builtin_amdgcn_sched_barrier(0);
                    auto group = [&](auto grp_c) __attribute__((always_inline)) {
                        constexpr int GRP = decltype(grp_c)::value;
                        if constexpr (GRP + 1 < NG) read_g(std::integral_constant<int, GRP + 1>{});
#pragma unroll
                        for (int s = 0; s < 2; ++s)
#pragma unroll
                            for (int i = 0; i < RT; ++i)
#pragma unroll
                                for (int jj = 0; jj < GC; ++jj)
                                    if (S::active(i, GRP * GC + jj))
                                        acc[S::tile(i, GRP * GC + jj)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                            F[i][s], G[GRP & 1][jj][s], acc[S::tile(i, GRP * GC + jj)], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    group(std::integral_constant<int, 0>{});
                    if constexpr (NG > 1) group(std::integral_constant<int, 1>{});
                    if constexpr (NG > 2) group(std::integral_constant<int, 2>{});
                    static_assert(NG <= 3, "unrolled by hand up to three groups");
                }
            } else {
                // One column tile at a time: its primal and dual operands (G_p, G_d) of tile j+1 are read in front of tile
                // j's 3 * RT * 2 MFMAs.  Per k-step and row tile: accQ += F_p G_p, accQ += F_d G_d, accQ2 += F_d G_p -- the
                // two products into the same accumulator are RT MFMAs apart (dependent latency 40 cycles, issue 32).
                constexpr unsigned DHALF = (unsigned)(4 * IB * XP * 4);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    floatx2 Fp[RT], Fd[RT], Gp[2], Gd[2];
                    auto read_g = [&](auto j_c) __attribute__((always_inline)) {
                        constexpr int JJ = decltype(j_c)::value;
                        const float* p1 = reinterpret_cast<const float*>(xi + bcol[JJ] + b_c1 + u * 32);
                        const float* p2 = reinterpret_cast<const float*>(xi + bcol[JJ] + b_c2 + u * 32);
                        const float* d1 = reinterpret_cast<const float*>(xi + bcol[JJ] + b_c1 + DHALF + u * 32);
                        const float* d2 = reinterpret_cast<const float*>(xi + bcol[JJ] + b_c2 + DHALF + u * 32);
                        Gp[JJ & 1][0] = p1[0] + tb * p2[0];
                        Gp[JJ & 1][1] = p1[1] + tb * p2[1];
                        Gd[JJ & 1][0] = d1[0] + tb * d2[0];
                        Gd[JJ & 1][1] = d1[1] + tb * d2[1];
                    };
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        const floatx2 p1 = *reinterpret_cast<const floatx2*>(dyi + arow(i, 0) + a_c1 + u * 32);
                        const floatx2 p2 = *reinterpret_cast<const floatx2*>(dyi + arow(i, 0) + a_c2 + u * 32);
                        const floatx2 d1 = *reinterpret_cast<const floatx2*>(dyi + arow(i, 1) + a_c1 + u * 32);
                        const floatx2 d2 = *reinterpret_cast<const floatx2*>(dyi + arow(i, 1) + a_c2 + u * 32);
                        Fp[i] = p1 + sa2 * p2;
                        Fd[i] = d1 + sa2 * d2;
                    }
                    read_g(std::integral_constant<int, 0>{});
                    __builtin_amdgcn_sched_barrier(0);
                    auto column = [&](auto j_c) __attribute__((always_inline)) {
                        constexpr int JJ = decltype(j_c)::value;
                        if constexpr (JJ + 1 < CT) read_g(std::integral_constant<int, JJ + 1>{});
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
#pragma unroll
                            for (int i = 0; i < RT; ++i)
                                acc[S::tileq(0, i, JJ)] = __builtin_amdgcn_mfma_f32_16x16x4f32(Fp[i][s], Gp[JJ & 1][s], acc[S::tileq(0, i, JJ)], 0, 0, 0);
#pragma unroll
                            for (int i = 0; i < RT; ++i)
                                acc[S::tileq(1, i, JJ)] = __builtin_amdgcn_mfma_f32_16x16x4f32(Fd[i][s], Gp[JJ & 1][s], acc[S::tileq(1, i, JJ)], 0, 0, 0);
#pragma unroll
                            for (int i = 0; i < RT; ++i)
                                acc[S::tileq(0, i, JJ)] = __builtin_amdgcn_mfma_f32_16x16x4f32(Fd[i][s], Gd[JJ & 1][s], acc[S::tileq(0, i, JJ)], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    column(std::integral_constant<int, 0>{});
                    if constexpr (CT > 1) column(std::integral_constant<int, 1>{});
                    if constexpr (CT > 2) column(std::integral_constant<int, 2>{});
                    if constexpr (CT > 3) column(std::integral_constant<int, 3>{});
                    if constexpr (CT > 4) column(std::integral_constant<int, 4>{});
                    static_assert(CT <= 5, "unrolled by hand up to five column tiles");
                }
            }
        };
        using ST0 = std::integral_constant<int, 0>;
        using ST1 = std::integral_constant<int, 1>;

        // every wave has left the previous job's last compute before its LDS images are overwritten
        asm volatile("s_barrier" ::: "memory");
        issue(ST0{});
        for (int s = s_beg; s < s_end; s += 2) {
            gw_wait_dma_and_barrier();                      // step s landed (all waves); everybody is done with stage 1
            if (s + 1 < s_end) issue(ST1{});
            compute(ST0{});
            if (s + 1 < s_end) {
                gw_wait_dma_and_barrier();
                if (s + 2 < s_end) issue(ST0{});
                compute(ST1{});
            }
        }

        // ---- flush: accumulator fragments as they stand, 16 bytes per lane, slot = workgroup + job ----------------------
        float* out = p.part + ((size_t)((int)blockIdx.x + jb) * 8 + m) * (size_t)(NT * 256) + lane * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<floatx4*>(out + t * 256) = acc[t];

        g = J.step0 + s_end;
        ++jb;
    }
}

// ---- sum of a job's slots, in slot order (deterministic): red[job][form][tile][256] ----------------------------------
struct GwSumP {
    const float* part;
    float* red;
    int njobs, per, total, tile_floats;      // tile_floats = 8 * NT * 256
    int step0[GW_MAXJ + 1];
};
__global__ __launch_bounds__(256) void hcq_gw_sum_kernel(const GwSumP p) {
    const int jb = blockIdx.y;
    const int i4 = blockIdx.x * 256 + threadIdx.x;                  // float4 index inside the job's tile set
    if (i4 * 4 >= p.tile_floats) return;
    const int s0 = p.step0[jb], s1 = p.step0[jb + 1];
    if (s0 >= s1) return;
    const int w0 = s0 / p.per, w1 = (s1 - 1) / p.per;               // workgroups that hold steps of this job
    floatx4 a = (floatx4){0.f, 0.f, 0.f, 0.f};
    for (int w = w0; w <= w1; ++w)
        a += *reinterpret_cast<const floatx4*>(p.part + (size_t)(w + jb) * p.tile_floats + (size_t)i4 * 4);
    *reinterpret_cast<floatx4*>(p.red + (size_t)jb * p.tile_floats + (size_t)i4 * 4) = a;
}

// ---- the eight forms -> the component gradients of one convolution (all its sub-jobs) --------------------------------
struct GwFoldConv {
    float* dw[8];
    int job0;              // first sub-job (index into red)
    int nsub;              // sub-jobs: KHs * KWs
    int KH, KWfull;        // kernel of the convolution (weights are [OA][IB][KH][KWfull])
    int subKW;             // taps per sub-job (KWfull or 1)
};
struct GwFoldP {
    const float* red;
    int tile_floats;
    int nconv;
    GwFoldConv conv[24];
};
template <int OA, int IB, int KW, int XP>
__global__ __launch_bounds__(256) void hcq_gw_fold_kernel(const GwFoldP p) {
    using S = GwShape<OA, IB, KW, XP>;
    const GwFoldConv& cv = p.conv[blockIdx.y];
    const int per_sub = 2 * OA * IB * KW;                            // (set, o, ib, tap) elements per sub-job
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cv.nsub * per_sub) return;
    const int sub = idx / per_sub;
    int rem = idx - sub * per_sub;
    const int set = rem / (OA * IB * KW);
    rem -= set * (OA * IB * KW);
    const int o = rem / (IB * KW);
    rem -= o * (IB * KW);
    const int t = rem / IB, ib = rem - t * IB;
    const float* red = p.red + (size_t)(cv.job0 + sub) * p.tile_floats;
    auto at = [&](int mform, int rhalf, int chalf) {
        const int r = rhalf * OA + o, c = chalf * IB * KW + t * IB + ib;
        const int ti = S::tile(r >> 4, c >> 4);
        const int ln = ((r & 15) >> 2) * 16 + (c & 15);
        return red[((size_t)mform * S::NT + ti) * 256 + ln * 4 + (r & 3)];
    };
    auto atq = [&](int mform) {
        const int c = t * IB + ib;
        const int ti = S::tileq(set, o >> 4, c >> 4);
        const int ln = ((o & 15) >> 2) * 16 + (c & 15);
        return red[((size_t)mform * S::NT + ti) * 256 + ln * 4 + (o & 3)];
    };
    float P[8];
#pragma unroll
    for (int mf = 0; mf < 8; ++mf) {
        float v;
        if constexpr (S::COMB) v = atq(mf);
        else v = set == 0 ? at(mf, 0, 0) + at(mf, 1, 1) : at(mf, 1, 0);
        P[mf] = gw_g_sign(mf) * v;
    }
    const float h0 = 0.5f * P[0], h1 = 0.5f * P[1], h2 = 0.5f * P[2], h3 = 0.5f * P[3];
    const float c[4] = {(h3 - h0) + (h1 + h2) + P[4], (h3 - h0) - (h1 + h2) + P[5], (h3 + h0) + (h2 - h1) + P[6],
                        (h3 + h0) + (h1 - h2) - P[7]};
    // sub-job -> (kh, first kw) of the convolution's kernel
    const int kws = cv.KWfull / cv.subKW;                            // sub-jobs per kernel row
    const int kh = sub / kws, kw = (sub - kh * kws) * cv.subKW + t;
    const size_t e = (((size_t)o * IB + ib) * cv.KH + kh) * cv.KWfull + kw;
#pragma unroll
    for (int q = 0; q < 4; ++q) cv.dw[set * 4 + q][e] += c[q];
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
enum { GW_KIND_NONE = -1, GW_K_48_24_3 = 0, GW_K_24_48_1 = 1, GW_K_24_24_3 = 2, GW_K_48_48_1S = 3, GW_NKIND = 4 };

struct GwKindInfo { int OA, IB, KW, XP, NT; };
static const GwKindInfo kGwKinds[GW_NKIND] = {
    {48, 24, 3, 20, GwShape<48, 24, 3, 20>::NT},      // TCN dilated 1x3 (192 -> 384), tcn.conv1
    {24, 48, 1, 16, GwShape<24, 48, 1, 16>::NT},      // TCN 1x1 (384 -> 192)
    {24, 24, 3, 20, GwShape<24, 24, 3, 20>::NT},      // one kernel row of a 3x3 layer (192 -> 192)
    {48, 48, 1, 20, GwShape<48, 48, 1, 20>::NT},      // one tap of a 1x3 layer 384 -> 384 (tcn.conv2)
};

// kind of a convolution and how it splits into sub-jobs; GW_KIND_NONE = not taken
static int gw_kind(const seld_conv_desc* d, int* nsub) {
    *nsub = 0;
    if (d->algebra != 8 || d->groups != 1) return GW_KIND_NONE;
    if (d->stride[0] != 1 || d->stride[1] != 1) return GW_KIND_NONE;
    const int KH = d->k[0], KW = d->k[1];
    int o[2];
    hc_out_shape(d, o);
    if (o[0] != d->in[0] || o[1] != d->in[1]) return GW_KIND_NONE;                      // 'same'
    if (d->in[1] % 16 || d->in[1] < 128) return GW_KIND_NONE;
    if ((long long)d->Cin * d->in[0] * d->in[1] * 4 >= (1LL << 31) || (long long)d->Cout * d->in[0] * d->in[1] * 4 >= (1LL << 31))
        return GW_KIND_NONE;
    const int OA = d->Cout / 8, IB = d->Cin / 8;
    // tap columns must stay inside the 64-position bias of the x descriptor and the staged window
    if (d->pad[1] > GW_BIAS - 4 || (KW - 1) * d->dil[1] - d->pad[1] > GW_BIAS - 4) return GW_KIND_NONE;
    if (KH == 1 && KW == 3 && OA == 48 && IB == 24) { *nsub = 1; return GW_K_48_24_3; }
    if (KH == 1 && KW == 1 && OA == 24 && IB == 48 && d->pad[1] == 0) { *nsub = 1; return GW_K_24_48_1; }
    if (KH == 3 && KW == 3 && OA == 24 && IB == 24) { *nsub = 3; return GW_K_24_24_3; }
    if (KH == 1 && KW == 3 && OA == 48 && IB == 48) { *nsub = 3; return GW_K_48_48_1S; }
    return GW_KIND_NONE;
}

static int gw_num_wgs() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = prop.multiProcessorCount;
        else
            n = 256;
    }
    return env().wgrad_wgs ? (int)env().wgrad_wgs : n;
}

struct GwPlanKind {
    std::vector<int> convs;          // indices into the caller's job array
    int nsubjobs = 0;
    long long total = 0;
    int nwg = 0, per = 0;
    size_t part_off = 0, red_off = 0, bytes = 0;     // byte offsets into the workspace
};

static size_t gw_plan(const seld_wgrad_job* jobs, int njobs, GwPlanKind plan[GW_NKIND]) {
    for (int i = 0; i < njobs; ++i) {
        if (hc_validate(&jobs[i].desc) != SELD_OK) return 0;
        int nsub;
        const int k = gw_kind(&jobs[i].desc, &nsub);
        if (k == GW_KIND_NONE) return 0;
        const seld_conv_desc& d = jobs[i].desc;
        plan[k].convs.push_back(i);
        plan[k].nsubjobs += nsub;
        plan[k].total += (long long)nsub * d.N * d.in[0] * (d.in[1] / 16);
    }
    size_t off = 0;
    const int G = gw_num_wgs();
    for (int k = 0; k < GW_NKIND; ++k) {
        GwPlanKind& pk = plan[k];
        if (pk.convs.empty()) continue;
        if (pk.nsubjobs > GW_MAXJ || pk.convs.size() > 24 || pk.total >= (1LL << 30)) return 0;
        pk.nwg = (int)(pk.total < G ? pk.total : G);
        pk.per = (int)((pk.total + pk.nwg - 1) / pk.nwg);
        pk.nwg = (int)((pk.total + pk.per - 1) / pk.per);
        const size_t tile_bytes = (size_t)8 * kGwKinds[k].NT * 256 * sizeof(float);
        pk.part_off = off;
        off += (size_t)(pk.nwg + pk.nsubjobs) * tile_bytes;
        pk.red_off = off;
        off += (size_t)pk.nsubjobs * tile_bytes;
        pk.bytes = off - pk.part_off;
    }
    return off;
}

template <int OA, int IB, int KW, int XP>
static int gw_launch_kind(const seld_wgrad_job* jobs, const GwPlanKind& pk, int subKW, unsigned char* ws, hipStream_t st) {
    using S = GwShape<OA, IB, KW, XP>;
    GwP p{};
    GwSumP sp{};
    GwFoldP fp{};
    p.per = pk.per; p.total = (int)pk.total;
    p.part = (float*)(ws + pk.part_off);
    int nj = 0, step = 0;
    fp.nconv = (int)pk.convs.size();
    for (size_t ci = 0; ci < pk.convs.size(); ++ci) {
        const seld_wgrad_job& jb = jobs[pk.convs[ci]];
        const seld_conv_desc& d = jb.desc;
        const int KH = d.k[0], KWf = d.k[1];
        GwFoldConv& fc = fp.conv[ci];
        for (int q = 0; q < 8; ++q) fc.dw[q] = jb.dw[q];
        fc.job0 = nj; fc.KH = KH; fc.KWfull = KWf; fc.subKW = subKW; fc.nsub = KH * (KWf / subKW);
        const int steps = d.N * d.in[0] * (d.in[1] / 16);
        for (int kh = 0; kh < KH; ++kh)
            for (int kw0 = 0; kw0 < KWf; kw0 += subKW) {
                GwJob& j = p.job[nj];
                j.x = jb.x; j.dy = jb.dy; j.H = d.in[0]; j.W = d.in[1];
                j.woff0 = kw0 * d.dil[1] - d.pad[1];
                j.wstep = d.dil[1];
                j.hoff = kh * d.dil[0] - d.pad[0];
                j.step0 = step;
                sp.step0[nj] = step;
                step += steps;
                ++nj;
            }
    }
    sp.step0[nj] = step;
    p.njobs = nj;
    hipLaunchKernelGGL((hcq_wgrad_grp_kernel<OA, IB, KW, XP>), dim3(pk.nwg), dim3(512), 0, st, p);
    int rc = check_launch();
    if (rc) return rc;
    sp.part = p.part; sp.red = (float*)(ws + pk.red_off); sp.njobs = nj; sp.per = pk.per; sp.total = (int)pk.total;
    sp.tile_floats = 8 * S::NT * 256;
    hipLaunchKernelGGL(hcq_gw_sum_kernel, dim3((sp.tile_floats / 4 + 255) / 256, nj), dim3(256), 0, st, sp);
    rc = check_launch();
    if (rc) return rc;
    fp.red = sp.red; fp.tile_floats = sp.tile_floats;
    int maxel = 0;
    for (int ci = 0; ci < fp.nconv; ++ci) {
        const int el = fp.conv[ci].nsub * 2 * OA * IB * KW;
        if (el > maxel) maxel = el;
    }
    hipLaunchKernelGGL((hcq_gw_fold_kernel<OA, IB, KW, XP>), dim3((maxel + 255) / 256, fp.nconv), dim3(256), 0, st, fp);
    return check_launch();
}

}  // namespace seld
using namespace seld;

/* Shape family of a convolution in the grouped weight-gradient kernels: 0 = (48, 24, 1x3), 1 = (24, 48, 1x1),
 * 2 = (24, 24, 3x3), 3 = (48, 48, 1x3) in (Cout/8, Cin/8, kernel); -1 = not taken.  One call of seld_hcq_wgrad_group
 * makes one persistent launch per family present in its list, so a caller batches by family. */
extern "C" int seld_hcq_wgrad_group_family(const seld_conv_desc* d) {
    if (hc_validate(d) != SELD_OK || env().conv_no_hcq) return -1;
    int nsub;
    return gw_kind(d, &nsub);
}

/* Bytes of scratch seld_hcq_wgrad_group needs for this job list; 0 = a job is not a shape these kernels take (the
 * caller then uses the per-layer entry points).  The scratch need not be initialised and is not kept between calls. */
extern "C" size_t seld_hcq_wgrad_group_workspace(const seld_wgrad_job* jobs, int32_t njobs) {
    if (!jobs || njobs <= 0 || env().conv_no_hcq) return 0;
    GwPlanKind plan[GW_NKIND];
    return gw_plan(jobs, njobs, plan);
}

/* dw[c] += weight gradient of every listed dual-quaternion convolution (jobs[i].desc; x, dy device tensors; dw the 8
 * component gradient tensors) -- dual_quaternion_ops.py:111-153 differentiated w.r.t. the weights, for a whole list of
 * layers in (per shape family) three launches.  Reproducible: no atomics, fixed summation order. */
extern "C" int seld_hcq_wgrad_group(const seld_wgrad_job* jobs, int32_t njobs, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    if (!jobs || njobs <= 0 || !workspace) return SELD_EINVAL;
    for (int i = 0; i < njobs; ++i) {
        if (!jobs[i].x || !jobs[i].dy) return SELD_EINVAL;
        for (int q = 0; q < 8; ++q)
            if (!jobs[i].dw[q]) return SELD_EINVAL;
    }
    GwPlanKind plan[GW_NKIND];
    const size_t need = gw_plan(jobs, njobs, plan);
    if (!need) return SELD_EUNSUPPORTED;
    if (workspace_bytes < need) return SELD_EWORKSPACE;
    if ((uintptr_t)workspace & 15) return SELD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)workspace;
    int rc = SELD_OK;
    if (!plan[GW_K_48_24_3].convs.empty()) rc = gw_launch_kind<48, 24, 3, 20>(jobs, plan[GW_K_48_24_3], 3, ws, st);
    if (!rc && !plan[GW_K_24_48_1].convs.empty()) rc = gw_launch_kind<24, 48, 1, 16>(jobs, plan[GW_K_24_48_1], 1, ws, st);
    if (!rc && !plan[GW_K_24_24_3].convs.empty()) rc = gw_launch_kind<24, 24, 3, 20>(jobs, plan[GW_K_24_24_3], 3, ws, st);
    if (!rc && !plan[GW_K_48_48_1S].convs.empty()) rc = gw_launch_kind<48, 48, 1, 20>(jobs, plan[GW_K_48_48_1S], 1, ws, st);
    return rc;
}
