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

// GW_DBG: timing experiments of tools/gw_micro.cpp ONLY (parts of the kernel switched off, WRONG results); the library is
// always built with 0.   1 = no LDS-DMA   2 = no LDS fragment reads   4 = no MFMAs   8 = no barriers
#ifndef GW_DBG
#define GW_DBG 0
#endif

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
    // x image of one tap: [8*IB rows][XP floats].  One LDS-DMA instruction covers XR whole rows (lane -> row lane / XC,
    // piece lane % XC: the same for every instruction); with 5 pieces per row that is 12 rows = 60 lanes, and lanes 60..63
    // carry the first four pieces of the NEXT 12 rows to where they belong (the next instruction writes the same values
    // there) -- behind the last rows they fall into 64 bytes of padding.
    static constexpr int XR = 64 / XC;                     // rows per instruction
    static constexpr int XTB = 8 * IB * XP * 4 + (64 % XC ? 64 : 0);
    static constexpr int XB = KW * XTB;
    static constexpr int NDYI = DYB / 1024, XTI = 8 * IB / XR, NINSTR = NDYI + KW * XTI;
    static_assert((8 * IB) % XR == 0 && NDYI % 4 == 0 && XTI % 4 == 0, "four loader waves, whole instructions per image");
    static_assert((2 * OA) % 16 == 0 && (2 * IB * KW) % 16 == 0, "tile structure");
    static_assert(DYB % 1024 == 0, "an LDS-DMA instruction fills 1 KiB of the dy image");
    // SPLIT
    __host__ __device__ static constexpr int tile(int i, int j) { return i < NPR ? i * CTP + j : NPR * CTP + (i - NPR) * CT + j; }
    __host__ __device__ static constexpr bool active(int i, int j) { return i >= NPR || j < CTP; }
    // COMB: set 0 = dQ, 1 = dQ2
    __host__ __device__ static constexpr int tileq(int set, int i, int j) { return (set * RT + i) * CT + j; }
};

// One 16-byte-per-lane LDS-DMA: LDS[lds_addr + 16 * lane ..] <- buffer[voff ..] (zero when voff is out of range).  Inline
// asm: the compiler's waitcnt pass would otherwise put vmcnt(0) in front of every LDS read that follows (it cannot tell
// the two stages apart); the kernel counts these loads itself (gw_wait_dma).  M0 is saved and restored in the statement.
__device__ __forceinline__ void gw_dma16(unsigned lds_addr, unsigned voff, int4v rsrc, unsigned soff) {
    if (GW_DBG & 1) return;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void gw_wait_dma_and_barrier() {
    if (GW_DBG & 8) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(N) : "memory");
}
__device__ __forceinline__ floatx4 gw_mfma(float a, float b, floatx4 c) {
    if (GW_DBG & 4) { c[0] += a; c[1] += b; return c; }
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// a fragment read; with GW_DBG & 2 the value comes from a register (no LDS instruction)
// the lane's two floats of half-step u from an x image row: XP == 16 -> swizzled 64-byte rows, one aligned 8-byte read;
// else the window starts at any column: two 4-byte aligned floats, u * 32 bytes on
template <int XP>
__device__ __forceinline__ floatx2 gw_ldx(const unsigned char* base, unsigned off, int u, const unsigned (&sw)[2]) {
    if (GW_DBG & 2) return (floatx2){__uint_as_float(off), __uint_as_float(off + 1)};
    if constexpr (XP == 16) {
        return *reinterpret_cast<const floatx2*>(base + off + sw[u]);
    } else {
        const float* q = reinterpret_cast<const float*>(base + off + u * 32);
        return (floatx2){q[0], q[1]};
    }
}
template <typename T>
__device__ __forceinline__ T gw_lds(const unsigned char* base, unsigned off) {
    if (GW_DBG & 2) { T v; for (int i = 0; i < (int)(sizeof(T) / 4); ++i) v[i] = __uint_as_float(off + i); return v; }
    return *reinterpret_cast<const T*>(base + off);
}
__device__ __forceinline__ int4v gw_rsrc(const float* base) {
    const unsigned long long a = (unsigned long long)base;
    return (int4v){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)0x80000000u, 0x00020000};
}

// NSTG = stages of the LDS ring (2, or more where a stage is small: the 1x1 layers' steps are too short for one stage of
// lead to cover the memory latency).
template <int OA, int IB, int KW, int XP, int NSTG>
__global__ __launch_bounds__(512, 2) void hcq_wgrad_grp_kernel(const GwP p) {
    using S = GwShape<OA, IB, KW, XP>;
    constexpr int RT = S::RT, CT = S::CT, NT = S::NT;
    constexpr bool COMB = S::COMB;
    // Waves 0..3 are the LOADERS (one per SIMD: waves w and w + 4 share one): KD LDS-DMA instructions each per stage.  The
    // SIMD's other wave loads nothing, so the two partners -- same program, one barrier per step -- do not march in
    // lockstep through their non-MFMA stretches with the matrix pipe idle (r3d counters: 54 % MFMA busy, 64 % of wave
    // cycles issue-stalled with every wave loading).
    constexpr int KD = S::NINSTR / 4;
    static_assert(NSTG >= 2 && (NSTG - 2) * KD <= 63, "vmcnt is a 6-bit count");
    static_assert(NSTG * (S::DYB + S::XB) <= 160 * 1024, "the ring must fit the CU's LDS");
    // [dy stage 0 .. NSTG-1][x stage 0 .. NSTG-1]
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTG * (S::DYB + S::XB)];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    constexpr unsigned XBASE = NSTG * S::DYB;

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
    // 16-position pass.  dy image: row (half*4 + c)*OA + o, 64 bytes per row.  COMB: the row tiles differ by compile-time
    // offsets from ONE lane register; SPLIT: one register per tile (the middle tile mixes primal and dual rows).
    // Bank conflicts: with 64-byte rows the 16 rows a read touches fall on 4 bank groups (r3g: the LDS array was busy with
    // conflicts 72 % of the 1x1 kernel's time, the LDS phase cost as much as the MFMAs).  The 16-byte pieces of a row are
    // therefore XOR-swizzled: logical piece c of image row R is stored at piece  c ^ ((R >> 2) & 3).  The LDS image stays
    // lane-linear for the DMA -- the loader swaps which piece of its row a lane FETCHES (R >> 2 & 3 = lane >> 4 & 3 there,
    // whatever the instruction) -- and a ds_read_b64 of 16 rows x {2 pieces halves} covers all 64 banks once.
    // In half u, lane (fr, fk) reads logical piece 2u + (fk >> 1), bytes 8 * (fk & 1) .. +7 of it.
    // COMB (OA a multiple of 16): R >> 2 & 3 = fr >> 2 & 3 for every row tile and component: ONE register per half.
    // SPLIT: R = (half*4 + comp)*OA + o depends on the component: one register per (row tile, component slot, half).
    const unsigned a_c1 = (unsigned)(ca1 * OA * 64), a_c2 = (unsigned)(ca2 * OA * 64);
    unsigned sw[2];                                                      // the swizzled piece + byte offset for sigma = fr >> 2 & 3
#pragma unroll
    for (int u = 0; u < 2; ++u) sw[u] = (unsigned)((((2 * u + (fk >> 1)) ^ ((fr >> 2) & 3)) << 4) + ((fk & 1) << 3));
    constexpr int NAR = COMB ? 1 : RT;
    unsigned arow_[NAR][2][2];
#pragma unroll
    for (int i = 0; i < NAR; ++i) {
        const int r = 16 * i + fr;
        const int half = r >= OA ? 1 : 0;
        const int o = r - half * OA;
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) {
            const int R = (half * 4 + (cs ? ca2 : ca1)) * OA + o;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                arow_[i][cs][u] = COMB ? (unsigned)(fr * 64) + sw[u]
                                       : (unsigned)(R * 64 + (((2 * u + (fk >> 1)) ^ ((R >> 2) & 3)) << 4) + ((fk & 1) << 3));
        }
    }
    // byte address (inside a stage's dy image) of this lane's 8 bytes: row tile i (COMB: of half `half`), component slot cs, half-step u
    auto arow = [&](int i, int half, int cs, int u) __attribute__((always_inline)) -> unsigned {
        if constexpr (COMB) return arow_[0][0][u] + (unsigned)((half * 4 * OA + 16 * i) * 64) + (cs ? a_c2 : a_c1);
        else return arow_[i][cs][u];
    };
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

        // x image of tap t: window of XP floats starting at column  w0 + floor4(woff0 + t*wstep);  the fragment reads start
        // (woff0 + t*wstep) & 3 floats into it.  Column (half, tap, ib) of lane fr.  (COMB: the CT column tiles of the primal
        // half; the dual half lies 4*IB image rows further; columns past IB*KW in the last tile read column 0 -- their
        // products land in output columns the fold never looks at.)
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
            // XP == 16 (aligned windows, 64-byte rows): the same swizzle as the dy image, sigma = ib >> 2 & 3 = fr >> 2 & 3
            // (IB and IB*KW multiples of 16 there), added per half-step from sw[]
            if constexpr (XP == 16) bcol[j] = (unsigned)(t * S::XTB + (half * 4 * IB + ib) * 64);
            else bcol[j] = (unsigned)(t * S::XTB + ((half * 4 * IB + ib) * XP + ofs + 2 * fk) * 4);
        }

        floatx4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (floatx4){0.f, 0.f, 0.f, 0.f};

        // ---- the loader: wave-uniform tracker of the step to LOAD next, and that step's descriptors -----------------------
        const bool loader = wave < 4;
        int l_wi, l_h, l_n;
        {
            const int row = s_beg / spr;
            l_wi = s_beg - row * spr;
            l_n = row / H;
            l_h = row - l_n * H;
        }
        // lane parts of the buffer offsets (bytes): dy piece (row lane >> 2, piece lane & 3); x piece (row lane / XC, piece
        // lane % XC), and the same with lanes >= XR * XC switched off for the last instruction of an image
        // (swizzle: the lane that WRITES piece lane & 3 of its row fetches piece (lane & 3) ^ (lane >> 4 & 3))
        const unsigned ldy = (unsigned)(lane >> 2) * dy_rs + (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;
        const int lch = S::XC == 4 ? ((lane & 3) ^ ((lane >> 4) & 3)) : lane % S::XC;
        const unsigned lx = (unsigned)(lane / S::XC) * x_rs + (unsigned)lch * 16u;
        const unsigned lx_last = lane < S::XR * S::XC ? lx : GW_OOB;
        const unsigned x_blk = (unsigned)S::XR * x_rs;                 // bytes between the row blocks of two instructions
        int al[KW];                                                     // first column of tap t's window, relative to w0
#pragma unroll
        for (int t = 0; t < KW; ++t) al[t] = (J.woff0 + t * J.wstep) & ~3;
        int4v dyr, xr;
        int lo[KW], hi[KW];                                             // valid pieces of tap t's rows in the step being loaded
        auto load_begin = [&]() __attribute__((always_inline)) {
            const int w0 = l_wi << 4;
            const float* dyb = J.dy + (long long)l_n * dy_img + (long long)l_h * W + w0;
            const int hi_ = l_h + J.hoff;
            const bool rowok = (unsigned)hi_ < (unsigned)H;
            const float* xb = J.x + (long long)l_n * x_img + (long long)hi_ * W + (w0 - GW_BIAS);
            dyr = gw_rsrc(dyb);
            xr = gw_rsrc(xb);
#pragma unroll
            for (int t = 0; t < KW; ++t) {
                const int c0 = w0 + al[t];                            // a multiple of 4; piece ch covers columns c0 + 4ch .. +3
                int l = c0 < 0 ? (-c0) >> 2 : 0;
                int h = (W - 4 - c0) >> 2;                            // last piece that ends inside the row (may be < 0)
                if (h > S::XC - 1) h = S::XC - 1;
                // piece ch is valid iff (unsigned)(ch - lo) <= (unsigned)hi, hi >= 0; an EMPTY range (window wholly outside
                // the row, or the input row outside the image) gets lo beyond every ch: the difference wraps to a huge value
                const bool none = !rowok || h < l;
                lo[t] = none ? (1 << 20) : l;
                hi[t] = none ? 0 : h - l;
            }
            if (++l_wi == spr) { l_wi = 0; if (++l_h == H) { l_h = 0; ++l_n; } }
        };
        // DMA number K (of KD) of a loader wave for the step load_begin() described, into stage STG: instruction 4K + wave.
        // (NDYI and XTI are multiples of 4: whether it is a dy or an x instruction, and of which tap, is compile-time.)
        auto dma = [&](auto stage_c, auto k_c) __attribute__((always_inline)) {
            constexpr int STG = decltype(stage_c)::value;
            constexpr int K = decltype(k_c)::value;
            constexpr int Q0 = 4 * K;
            if constexpr (Q0 < S::NDYI) {
                const unsigned q = (unsigned)(Q0 + wave);
                gw_dma16(lds0 + STG * S::DYB + q * 1024u, ldy, dyr, q * 16u * dy_rs);
            } else {
                constexpr int T = (Q0 - S::NDYI) / S::XTI;
                const unsigned e = (unsigned)(Q0 - S::NDYI - T * S::XTI + wave);        // row block inside the tap image
                unsigned v = (e == S::XTI - 1) ? lx_last : lx;
                v = ((unsigned)(lch - lo[T]) <= (unsigned)hi[T]) ? v : GW_OOB;
                gw_dma16(lds0 + XBASE + STG * S::XB + T * S::XTB + e * (unsigned)(S::XR * S::XC * 16), v, xr,
                         e * x_blk + (unsigned)((GW_BIAS + al[T]) * 4));
            }
        };
        // DMAs [K0, K1) -- compile-time range
        auto dma_range = [&](auto stage_c, auto k0_c, auto k1_c) __attribute__((always_inline)) {
            constexpr int K0 = decltype(k0_c)::value, K1 = decltype(k1_c)::value;
            if constexpr (K0 < K1) {
                dma(stage_c, std::integral_constant<int, K0>{});
                if constexpr (K0 + 1 < K1) dma(stage_c, std::integral_constant<int, K0 + 1>{});
                if constexpr (K0 + 2 < K1) dma(stage_c, std::integral_constant<int, K0 + 2>{});
                if constexpr (K0 + 3 < K1) dma(stage_c, std::integral_constant<int, K0 + 3>{});
                if constexpr (K0 + 4 < K1) dma(stage_c, std::integral_constant<int, K0 + 4>{});
                if constexpr (K0 + 5 < K1) dma(stage_c, std::integral_constant<int, K0 + 5>{});
                static_assert(K1 - K0 <= 6, "at most six DMAs per slot");
            }
        };
        auto load_all = [&](auto stage_c) __attribute__((always_inline)) {
            if (loader) {
                load_begin();
                dma_range(stage_c, std::integral_constant<int, 0>{}, std::integral_constant<int, (KD < 6 ? KD : 6)>{});
                dma_range(stage_c, std::integral_constant<int, (KD < 6 ? KD : 6)>{}, std::integral_constant<int, (KD < 12 ? KD : 12)>{});
                dma_range(stage_c, std::integral_constant<int, (KD < 12 ? KD : 12)>{}, std::integral_constant<int, KD>{});
                static_assert(KD <= 18, "load_all is unrolled for up to eighteen DMAs per loader wave");
            }
        };

        // ---- one step: the MFMAs of stage STG; the DMAs of the step NSTG-1 ahead are spread over its MFMA blocks ------------
        // Block b of NB gets DMAs [b*KD/NB, (b+1)*KD/NB): issued right behind the block's MFMAs they cost ~60 cycles each
        // instead of a burst of KD at the top of the step with the matrix pipe idle.
        auto compute = [&](auto stage_c, auto lstage_c, bool more) __attribute__((always_inline)) {
            constexpr int STG = decltype(stage_c)::value;
            const unsigned char* dyi = smem + STG * S::DYB;
            const unsigned char* xi = smem + XBASE + STG * S::XB;
            more = more && loader;
            if (more) load_begin();
            if constexpr (!COMB) {
                // SPLIT.  Column tiles in groups of GC: the raw reads of group g+1 are issued in front of group g's MFMAs and
                // summed behind them, so their LDS latency lies under the MFMAs.
                constexpr int GC = 3, NG = CT / GC, NB = 2 * NG;
                static_assert(COMB || CT % GC == 0, "column tiles come in groups of three");
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    floatx2 F[RT], G[GC], r1[GC], r2[GC];
                    auto read_raw = [&](auto grp_c) __attribute__((always_inline)) {
                        constexpr int GRP = decltype(grp_c)::value;
#pragma unroll
                        for (int jj = 0; jj < GC; ++jj) {
                            r1[jj] = gw_ldx<XP>(xi, bcol[GRP * GC + jj] + b_c1, u, sw);
                            r2[jj] = gw_ldx<XP>(xi, bcol[GRP * GC + jj] + b_c2, u, sw);
                        }
                    };
                    auto form_g = [&]() __attribute__((always_inline)) {
#pragma unroll
                        for (int jj = 0; jj < GC; ++jj) G[jj] = r1[jj] + tb * r2[jj];
                    };
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        const floatx2 v1 = gw_lds<floatx2>(dyi, arow(i, 0, 0, u));
                        const floatx2 v2 = gw_lds<floatx2>(dyi, arow(i, 0, 1, u));
                        F[i] = v1 + sa2 * v2;
                    }
                    read_raw(std::integral_constant<int, 0>{});
                    form_g();
                    auto group = [&](auto grp_c) __attribute__((always_inline)) {
                        constexpr int GRP = decltype(grp_c)::value;
                        constexpr int BLK = 0;
                        floatx2 Gc[GC];
#pragma unroll
                        for (int jj = 0; jj < GC; ++jj) Gc[jj] = G[jj];
                        if constexpr (GRP + 1 < NG) read_raw(std::integral_constant<int, GRP + 1>{});
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s = 0; s < 2; ++s)
#pragma unroll
                            for (int i = 0; i < RT; ++i)
#pragma unroll
                                for (int jj = 0; jj < GC; ++jj)
                                    if (S::active(i, GRP * GC + jj))
                                        acc[S::tile(i, GRP * GC + jj)] = gw_mfma(
                                            F[i][s], Gc[jj][s], acc[S::tile(i, GRP * GC + jj)]);
                        (void)BLK;
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (GRP + 1 < NG) form_g();
                    };
                    // with two stages the step's DMAs all go into the FIRST half-step: the next step opens with vmcnt(0), and a
                    // DMA issued behind the last MFMA block would expose its whole memory latency there, every step
                    constexpr int NSLOT = NSTG > 2 ? NB : NG;
                    auto slot = [&](auto b_c) __attribute__((always_inline)) {
                        constexpr int B = decltype(b_c)::value;
                        if constexpr (B < NSLOT)
                            if (more) dma_range(lstage_c, std::integral_constant<int, B * KD / NSLOT>{}, std::integral_constant<int, (B + 1) * KD / NSLOT>{});
                    };
                    group(std::integral_constant<int, 0>{});
                    if (u == 0) slot(std::integral_constant<int, 0>{}); else slot(std::integral_constant<int, NG>{});
                    if constexpr (NG > 1) {
                        group(std::integral_constant<int, 1>{});
                        if (u == 0) slot(std::integral_constant<int, 1>{}); else slot(std::integral_constant<int, NG + 1>{});
                    }
                    if constexpr (NG > 2) {
                        group(std::integral_constant<int, 2>{});
                        if (u == 0) slot(std::integral_constant<int, 2>{}); else slot(std::integral_constant<int, NG + 2>{});
                    }
                    static_assert(NG <= 3, "unrolled by hand up to three groups");
                }
            } else {
                // COMB.  One column tile at a time: its primal and dual operands of tile j+1 are read in front of tile j's
                // 3 * RT * 2 MFMAs and summed behind them.  Per k-step and row tile: accQ += F_p G_p, accQ2 += F_d G_p,
                // accQ += F_d G_d -- the two products into the same accumulator are 2 * RT MFMAs apart.
                constexpr unsigned DHALF = (unsigned)(4 * IB * XP * 4);
                constexpr int NB = 2 * CT;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    floatx2 Fp[RT], Fd[RT], Gp, Gd, rp1, rp2, rd1, rd2;
                    auto read_raw = [&](auto j_c) __attribute__((always_inline)) {
                        constexpr int JJ = decltype(j_c)::value;
                        rp1 = gw_ldx<XP>(xi, bcol[JJ] + b_c1, u, sw);
                        rp2 = gw_ldx<XP>(xi, bcol[JJ] + b_c2, u, sw);
                        rd1 = gw_ldx<XP>(xi, bcol[JJ] + b_c1 + DHALF, u, sw);
                        rd2 = gw_ldx<XP>(xi, bcol[JJ] + b_c2 + DHALF, u, sw);
                    };
                    auto form_g = [&]() __attribute__((always_inline)) {
                        Gp = rp1 + tb * rp2;
                        Gd = rd1 + tb * rd2;
                    };
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        const floatx2 p1 = gw_lds<floatx2>(dyi, arow(i, 0, 0, u));
                        const floatx2 p2 = gw_lds<floatx2>(dyi, arow(i, 0, 1, u));
                        const floatx2 d1 = gw_lds<floatx2>(dyi, arow(i, 1, 0, u));
                        const floatx2 d2 = gw_lds<floatx2>(dyi, arow(i, 1, 1, u));
                        Fp[i] = p1 + sa2 * p2;
                        Fd[i] = d1 + sa2 * d2;
                    }
                    read_raw(std::integral_constant<int, 0>{});
                    form_g();
                    auto column = [&](auto j_c) __attribute__((always_inline)) {
                        constexpr int JJ = decltype(j_c)::value;
                        const floatx2 gp = Gp, gd = Gd;
                        if constexpr (JJ + 1 < CT) read_raw(std::integral_constant<int, JJ + 1>{});
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
#pragma unroll
                            for (int i = 0; i < RT; ++i)
                                acc[S::tileq(0, i, JJ)] = gw_mfma(Fp[i][s], gp[s], acc[S::tileq(0, i, JJ)]);
#pragma unroll
                            for (int i = 0; i < RT; ++i)
                                acc[S::tileq(1, i, JJ)] = gw_mfma(Fd[i][s], gp[s], acc[S::tileq(1, i, JJ)]);
#pragma unroll
                            for (int i = 0; i < RT; ++i)
                                acc[S::tileq(0, i, JJ)] = gw_mfma(Fd[i][s], gd[s], acc[S::tileq(0, i, JJ)]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (JJ + 1 < CT) form_g();
                    };
                    constexpr int NSLOT = NSTG > 2 ? NB : CT;          // two stages: all DMAs in the first half-step (see SPLIT)
                    auto slot = [&](auto b_c) __attribute__((always_inline)) {
                        constexpr int B = decltype(b_c)::value;
                        if constexpr (B < NSLOT)
                            if (more) dma_range(lstage_c, std::integral_constant<int, B * KD / NSLOT>{}, std::integral_constant<int, (B + 1) * KD / NSLOT>{});
                    };
                    column(std::integral_constant<int, 0>{});
                    if (u == 0) slot(std::integral_constant<int, 0>{}); else slot(std::integral_constant<int, CT>{});
                    if constexpr (CT > 1) {
                        column(std::integral_constant<int, 1>{});
                        if (u == 0) slot(std::integral_constant<int, 1>{}); else slot(std::integral_constant<int, CT + 1>{});
                    }
                    if constexpr (CT > 2) {
                        column(std::integral_constant<int, 2>{});
                        if (u == 0) slot(std::integral_constant<int, 2>{}); else slot(std::integral_constant<int, CT + 2>{});
                    }
                    if constexpr (CT > 3) {
                        column(std::integral_constant<int, 3>{});
                        if (u == 0) slot(std::integral_constant<int, 3>{}); else slot(std::integral_constant<int, CT + 3>{});
                    }
                    if constexpr (CT > 4) {
                        column(std::integral_constant<int, 4>{});
                        if (u == 0) slot(std::integral_constant<int, 4>{}); else slot(std::integral_constant<int, CT + 4>{});
                    }
                    static_assert(CT <= 5, "unrolled by hand up to five column tiles");
                }
            }
        };

        // every wave has left the previous job's last compute before its LDS images are overwritten
        asm volatile("s_barrier" ::: "memory");
        // prologue: the first NSTG-1 steps' loads
        if (s_beg + 0 < s_end) load_all(std::integral_constant<int, 0>{});
        if constexpr (NSTG > 2) { if (s_beg + 1 < s_end) load_all(std::integral_constant<int, 1>{}); }
        if constexpr (NSTG > 3) { if (s_beg + 2 < s_end) load_all(std::integral_constant<int, 2>{}); }
        static_assert(NSTG <= 4, "prologue unrolled for up to four stages");
        auto step = [&](int s, auto r_c) __attribute__((always_inline)) {
            constexpr int R = decltype(r_c)::value;
            // the loads of steps s .. s+NSTG-2 are in flight (fewer at the end of the range): wait for the oldest stage
            if (s_end - s >= NSTG - 1) gw_wait_dma_and_barrier<(NSTG - 2) * KD>();
            else gw_wait_dma_and_barrier<0>();
            compute(r_c, std::integral_constant<int, (R + NSTG - 1) % NSTG>{}, s + NSTG - 1 < s_end);
        };
        for (int s = s_beg; s < s_end; s += NSTG) {
            step(s, std::integral_constant<int, 0>{});
            if (s + 1 < s_end) step(s + 1, std::integral_constant<int, 1>{});
            if constexpr (NSTG > 2) { if (s + 2 < s_end) step(s + 2, std::integral_constant<int, 2>{}); }
            if constexpr (NSTG > 3) { if (s + 3 < s_end) step(s + 3, std::integral_constant<int, 3>{}); }
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
    int w = w0;
    for (; w + 3 <= w1; w += 4) {                 // four requests in flight, added in workgroup order
        floatx4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const floatx4*>(p.part + (size_t)(w + k + jb) * p.tile_floats + (size_t)i4 * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) a += v[k];
    }
    for (; w <= w1; ++w)
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

template <int OA, int IB, int KW, int XP, int NSTG>
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
    hipLaunchKernelGGL((hcq_wgrad_grp_kernel<OA, IB, KW, XP, NSTG>), dim3(pk.nwg), dim3(512), 0, st, p);
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
    if (!plan[GW_K_48_24_3].convs.empty()) rc = gw_launch_kind<48, 24, 3, 20, 2>(jobs, plan[GW_K_48_24_3], 3, ws, st);
    if (!rc && !plan[GW_K_24_48_1].convs.empty()) rc = gw_launch_kind<24, 48, 1, 16, 4>(jobs, plan[GW_K_24_48_1], 1, ws, st);
    if (!rc && !plan[GW_K_24_24_3].convs.empty()) rc = gw_launch_kind<24, 24, 3, 20, 2>(jobs, plan[GW_K_24_24_3], 3, ws, st);
    if (!rc && !plan[GW_K_48_48_1S].convs.empty()) rc = gw_launch_kind<48, 48, 1, 20, 2>(jobs, plan[GW_K_48_48_1S], 1, ws, st);
    return rc;
}
