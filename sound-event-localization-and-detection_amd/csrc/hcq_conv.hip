// Quaternion / dual-quaternion convolution with the 8-multiplication Hamilton product (gfx950), forward and data
// gradient of the 1-D layers (1x3 dilated, 1x1) and the 3x3 layers.
//
// A Hamilton product w (x) x costs 16 real sub-products when it is evaluated as the 4 x 4 block matrix the reference
// assembles (quaternion_ops.py:131-135).  It is a bilinear map of rank 8: with
//
//     P0 = (a3 + a1)(b1 + b2)   P1 = (a0 - a2)(b0 + b3)   P2 = (a0 + a2)(b0 - b3)   P3 = (a3 - a1)(b1 - b2)
//     P4 = (a3 - a2)(b2 - b3)   P5 = (a1 + a0)(b1 + b0)   P6 = (a0 - a1)(b2 + b3)   P7 = (a3 + a2)(b1 - b0)
//
//     c0 = (-P0 + P1 + P2 + P3)/2 + P4      c1 = (-P0 - P1 - P2 + P3)/2 + P5
//     c2 = ( P0 - P1 + P2 + P3)/2 + P6      c3 = ( P0 + P1 - P2 + P3)/2 - P7
//
// (a = weight components r,i,j,k; b = input components; c = output components; the a's always stand on the left, so
// the identities hold for matrix-valued components, i.e. for convolutions).  The convolution therefore splits into 8
// INDEPENDENT real GEMMs  P_m = F_m(W) * G_m(X)  of one quarter of the K extent each -- half the MFMA work of the block
// matrix -- plus sums of two components on the way in and a signed sum of the eight accumulators on the way out.
// In fp32 the result differs from the 16-product evaluation by rounding only (measured against fp64 on the TCN layer:
// 2.5e-7 of max|y| against 1.8e-7 for the 16-product form).  The dual quaternion [[Q, 0], [Q2, Q]] is three such
// products, y_p = Q x_p,  y_d = Q2 x_p + Q x_d: 24 sub-products instead of 48.
//
// Kernel structure (one workgroup = 4 waves = 64 positions x one channel tile, 2 workgroups per CU):
//   * the weight forms F_m are precomputed ONCE per step by hcq_pack_kernel, already in MFMA B-fragment order
//     ([chunk][range][k-group pair][m][lane]); a wave reads its fragments with coalesced 8/16-byte loads straight from
//     L2 -- no LDS, no sign logic, no component switching in the loop;
//   * the input is staged RAW (no im2col): per K chunk of IBC block channels the 4 / 8 component rows of the tile plus
//     their halo, [component][channel][64 + 2*dpad] floats, 16-byte aligned loads; the taps are offsets into the rows;
//   * per k-group a lane reads its 4 component values, forms the 8 sums G_m (8 VALU) and issues 8 x tiles MFMAs
//     (v_mfma_f32_16x16x4_f32) into accumulators indexed [tile][m];
//   * two K ranges for the dual quaternion: range 0 reads the source half every output needs, range 1 the half only
//     one half of the outputs needs (the structural zero block is never touched);
//   * epilogue: the signed sums above, bias / addend / BatchNorm statistics, 16-byte stores.
#include <string.h>
#include <type_traits>
#include "hc_common.h"

// Phase ablation for timing experiments only (tools/hcq_ablate.sh builds variants into tools/_bin; results are WRONG with
// any bit set): 1 = weight fragments loaded once, 2 = LDS operand reads + sums once, 4 = input staged once,
// 8 = no per-chunk barrier.  The shipped library is built with 0.
// Round-3 readings on cnn.1 (forward / data gradient, us): as shipped 774 / 732; 1: 636 / 591; 4: 718 / 668; 8: no change;
// 1+2+4: 550 / 517 (the MFMA floor is 481).  The fragment loads are the largest term, and it is not their bytes, their
// request count or their latency window that costs: loading the 4 raw components per (pair, tile) and forming the 8 sums
// in registers (half the bytes and requests, two register stages = a whole pair of k-groups ahead) ran 773 / 745, with
// three-wave occupancy lost on the TCN layers (177 VGPRs) -- and again 757 against 701 once both versions had their requests
// pinned (HCQ_PIN), for the three-tile shapes only; raised wave priority (s_setprio 1) around each k-group's MFMAs 786 / 786; rotating the chunk order per workgroup (so that workgroups in
// step do not ask the L2 for the same lines) 751 / 723 against 752 / 722; staging the input from cache-resident
// addresses 712 against 723.  Left as it is.
#ifndef HCQ_DBG
#define HCQ_DBG 0
#endif
#ifndef HCQ_PIN
#define HCQ_PIN 1
#endif
#ifndef HCQ_XDMA
#define HCQ_XDMA 1          // input chunks global -> LDS with LDS-DMA (no staging registers, no ds_write), hcq_conv_kernel only
#endif

namespace seld {

struct HcqP {
    const float* src;
    const float* src2;           // data gradient of a pair: dst = dgrad(src, W_0) + dgrad(src2, W_1), the K loop runs over both
    int nsrc;                    // 1 or 2
    int mix_ytile;               // index of the channel tile that holds ONLY the mixed 8 + 8 tile (dual quaternion with
                                 // 24 block channels), or -1
    const float* wpack;
    float* dst[2];               // per weight set (a pair launch computes two convolutions of the same input)
    const float* bias[2];
    const float* addend[2];
    float* stats[2];
    int epilogue[2];
    int A;                       // 4 or 8
    int N, Csrc, Cdst;           // channels of src / of ONE dst
    int IB, OB;                  // Csrc / A, Cdst / A
    int W;                       // row length (T of a 1-D layer, image width of a 2-D layer)
    int Himg;                    // image height (1 for 1-D)
    int dil, dpad;               // dilation along W, padded up to a multiple of 4 (the halo each side of a tile)
    int wext;                    // 64 + 2 * dpad
    int nch;                     // K chunks per range (IB / IBC)
    int half_src[2];             // source half (0 primal, 1 dual) read by range 0 / range 1
    int ytiles;                  // channel tiles per weight set
    int ob_step;                 // block channels a channel tile advances by (16, or 0 for the single-tile layout)
    int tile_half[3][2];         // tile t, 8-channel group g: destination half ...
    int tile_ob[3][2];           // ... and first block channel (-1: padding, nothing stored); tile 2 = the mixed tile
    long long range_stride[2];   // floats of one (chunk, range) block of wpack
    long long ytile_stride;      // floats of one regular channel tile of wpack (all chunks of all sources)
    long long set_stride;        // floats of one weight set (regular tiles + the mixed tile's block)
};

typedef unsigned int uintx4h __attribute__((ext_vector_type(4)));
typedef int int4h __attribute__((ext_vector_type(4)));

// One 16-byte-per-lane LDS-DMA: LDS[lds_addr + 16 * lane ..] <- buffer[voff ..] (zeros when voff is out of range); lanes that
// are switched off write nothing.  M0 is saved and restored in the statement.  (csrc/hcq_wgrad_grp.hip uses the same.)
__device__ __forceinline__ void hcq_dma16(unsigned lds_addr, unsigned voff, int4h rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ int4h hcq_rsrc(const float* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    return (int4h){(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}

// n / d for 0 <= n < 2^20 with a precomputed 1.0f / d (exact: see hc_conv_vec.hip)
__device__ __forceinline__ int small_div_h(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }

// X-form m from the 4 component values (see the header)
__device__ __forceinline__ void xforms(const float b[4], float g[8]) {
    g[0] = b[1] + b[2];
    g[1] = b[0] + b[3];
    g[2] = b[0] - b[3];
    g[3] = b[1] - b[2];
    g[4] = b[2] - b[3];
    g[5] = b[1] + b[0];
    g[6] = b[2] + b[3];
    g[7] = b[1] - b[0];
}

// KH x KW taps (1x1, 1x3, 3x3), IBC block channels per K chunk, NT1 tiles active in range 0 only + NT2 tiles active
// in both ranges (NR = 1: quaternion, one range), XI staging items per thread.
// The channel tile p.mix_ytile (dual quaternion with 24 block channels, split layout) runs the same program with other
// descriptors: its range-0-only tile is padding, its both-range tile is descriptor slot 2 (8 channels of each half).
// MX: the launch has mixed-tile workgroups, whose copy of the K loop leaves the padding slots out (a second copy of the
// loop costs registers -- 186 instead of 160 for the TCN 1x3 shape -- so launches without such workgroups keep MX = false).
template <int KH, int KW, int IBC, int NT1, int NT2, int NR, int XI, bool MX = false>
__global__ __launch_bounds__(256, 2) void hcq_conv_kernel(const HcqP p) {
    constexpr int TAPS = KH * KW;
    constexpr int NT = NT1 + NT2;
    constexpr int KQ = IBC * TAPS;
    constexpr int NG = (KQ + 3) / 4;                  // the first layers (1 or 2 block channels x 9 taps) pad the last k-group:
                                                      // packed weights are zero there, the lane re-reads k = 0
    constexpr int NPAIR = (NG + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int A = p.A;
    const int ROWS = A * IBC * KH;                    // staged rows per chunk
    const int wext = p.wext, qw = wext >> 2;
    const int buf_floats = ROWS * wext;

    // ---- tile position (32-bit arithmetic throughout: the host checked that both tensors are below 4 GB) ------------
    const unsigned tiles_per_row = (unsigned)p.W >> 6;    // 64 consecutive positions inside ONE row (W % 64 == 0)
    // workgroups go round-robin to the 8 XCDs: give each XCD (its own L2) a contiguous range of position tiles, so that
    // the tiles sharing halo rows / output lines meet in one L2
    const unsigned bx = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const unsigned row_g = bx / tiles_per_row;            // n * Himg + h
    const int w0 = (int)(bx - row_g * tiles_per_row) * 64;
    const int n_img = (int)(row_g / (unsigned)p.Himg);
    const int h0 = (int)row_g - n_img * p.Himg;
    const int yt = blockIdx.y;                        // channel tile over all weight sets
    const int set = yt >= p.ytiles ? 1 : 0;           // at most two weight sets
    const int ytile = yt - set * p.ytiles;
    const bool mix_wg = ytile == p.mix_ytile;         // workgroup-uniform

    // ---- staging items: (row, quad) of the raw image of a chunk, everything but the chunk advance is invariant ----
    const unsigned S = (unsigned)(p.Himg * p.W);
    const unsigned src_bytes = (unsigned)p.N * (unsigned)p.Csrc * S * 4u;
    const unsigned OOB = 0xFFFFFFF0u;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, src_bytes, 0x00020000);
    unsigned xoff[XI];
    int xlds[XI];
    const unsigned xadv = (unsigned)IBC * S * 4u;
    const int total_items = ROWS * qw;
    {
        // item f = tid + 256 i  ->  (row, quad) = (f / qw, f % qw): one division, then increments
        int row = small_div_h(tid, 1.0f / (float)qw);
        int quad = tid - row * qw;
        const int drow = 256 / qw, dquad = 256 - drow * qw;       // wave-uniform
        const unsigned img_base = (unsigned)n_img * (unsigned)p.Csrc * S;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const bool in = tid + 256 * i < total_items;
            // row = (comp * IBC + ibl) * KH + kh, all divisors compile-time
            const int kh = row % KH;
            const int ci = row / KH;
            const int comp = ci / IBC;
            const int ibl = ci - comp * IBC;
            const int hh = h0 + (kh - (KH - 1) / 2);
            const int ww = w0 - p.dpad + 4 * quad;
            const bool ok = in && (unsigned)hh < (unsigned)p.Himg && (unsigned)ww < (unsigned)p.W;
            const unsigned e = img_base + ((unsigned)(comp * p.IB + ibl) * (unsigned)p.Himg + (unsigned)hh) * (unsigned)p.W + (unsigned)ww;
            xoff[i] = ok ? e * 4u : OOB;
            xlds[i] = in ? (row * qw + quad) * 4 : -1;     // float index of the quad in the LDS image; -1: no store
            row += drow;
            quad += dquad;
            if (quad >= qw) { quad -= qw; ++row; }
        }
    }
#if HCQ_XDMA
    // Input chunks go global -> LDS directly (buffer_load ... lds): the LDS image is linear in the item index (item f at byte
    // 16 f), so instruction i of wave w fills the contiguous KiB at 16 (256 i + 64 w); no staging registers (28 of them for
    // the 3x3 layers), no ds_write.  load_x(buf) requests the next chunk into `buf`; store_x waits for the wave's requests.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    int4h xrs = hcq_rsrc(p.src, src_bytes);
    int xchunk = 0;                                    // chunks requested so far
    int xbuf = 0;                                      // buffer the next request fills
    auto load_x = [&]() __attribute__((always_inline)) {
        if (xchunk == p.nch) {                         // second source of a pair: same offsets, other tensor
            xrs = hcq_rsrc(p.src2, src_bytes);
#pragma unroll
            for (int i = 0; i < XI; ++i) xoff[i] = xoff[i] == OOB ? OOB : xoff[i] - (unsigned)p.nch * xadv;
        }
        ++xchunk;
        const unsigned base = lds0 + (unsigned)xbuf * (unsigned)buf_floats * 4u + 1024u * (unsigned)wave;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            if (xlds[i] >= 0) hcq_dma16(base + 4096u * (unsigned)i, xoff[i], xrs);
            xoff[i] = xoff[i] == OOB ? OOB : xoff[i] + xadv;
        }
        xbuf ^= 1;
    };
    auto store_x = [&](int) __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
#else
    floatx4 xr[XI];
    int xchunk = 0;                                    // chunks requested so far
    auto load_x = [&]() __attribute__((always_inline)) {
        if (xchunk == p.nch) {                         // second source of a pair: same offsets, other tensor
            rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.src2, 0, src_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < XI; ++i) xoff[i] = xoff[i] == OOB ? OOB : xoff[i] - (unsigned)p.nch * xadv;
        }
        ++xchunk;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const uintx4h v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, xoff[i], 0, 0);
            xr[i][0] = __uint_as_float(v[0]); xr[i][1] = __uint_as_float(v[1]);
            xr[i][2] = __uint_as_float(v[2]); xr[i][3] = __uint_as_float(v[3]);
            xoff[i] = xoff[i] == OOB ? OOB : xoff[i] + xadv;
        }
    };
    auto store_x = [&](int buf) __attribute__((always_inline)) {
        float* b = lds + buf * buf_floats;
#pragma unroll
        for (int i = 0; i < XI; ++i)
            if (xlds[i] >= 0) *reinterpret_cast<float4*>(b + xlds[i]) = make_float4(xr[i][0], xr[i][1], xr[i][2], xr[i][3]);
    };
#endif

    // ---- A-operand addresses: k-group g, this lane's k = 4g + fk -> (ibl, kh, kw); component stride = IBC*KH*wext ----
    int aoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int kq = (4 * g + fk) < KQ ? 4 * g + fk : 0;
        const int ibl = kq / TAPS;
        const int tap = kq - ibl * TAPS;
        const int kh = tap / KW;
        const int kw = tap - kh * KW;
        aoff[g] = (ibl * KH + kh) * wext + p.dpad + wave * 16 + fr + (kw - (KW - 1) / 2) * p.dil;
    }
    const int comp_stride = IBC * KH * wext;

    // ---- weight fragments ------------------------------------------------------------------------------------------
    const float* wbase = p.wpack + (long long)set * p.set_stride + (long long)ytile * p.ytile_stride;
    const long long chunk_stride = p.range_stride[0] + (NR > 1 ? p.range_stride[1] : 0);

    floatx4 acc[NT][8];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[t][m] = (floatx4){0.f, 0.f, 0.f, 0.f};

    // ---- one K chunk = range 0's NG k-groups, then range 1's.  Software pipeline, forced with scheduling barriers
    // (left alone, the compiler sinks every fragment load to just before its MFMA and waits for it there: 43 % MFMA busy):
    //   * weight fragments come in pairs of k-groups, one register pair per form; a form's registers are re-requested
    //     for the next pair (or the next chunk's first pair) right after their last MFMA, so every request has the other
    //     seven forms' MFMAs and the next group's first ones in front of its use -- one stage instead of two keeps the
    //     kernel at 3 waves per SIMD;
    //   * the raw component values of k-group g+1 are read from LDS at the start of group g and turned into the 8 sums
    //     under the MFMAs of group g, so no VALU result feeds the very next MFMA.
    float2 bfr[8][NT];                                       // ONE stage: form m's pair is re-requested right after its last use
    float gm[2][8];
    float raw[4];
    constexpr int NPC = NR * NPAIR;                          // fragment pairs per chunk

    bool dbg_started = false;
    // T0: first tile that is loaded (a mixed-tile workgroup's range-0-only slots are padding: neither loaded nor multiplied)
    auto load_b1 = [&](const float* blk, int j, int m, auto ntrc, auto t0c) __attribute__((always_inline)) {
        constexpr int NTR = decltype(ntrc)::value, T0 = decltype(t0c)::value;
        if ((HCQ_DBG & 1) && dbg_started) return;
        const float* q = blk + ((long long)(j * 8 + m) * 64 + lane) * (2 * NTR);
#pragma unroll
        for (int t = T0; t < NTR; ++t) bfr[m][t] = *reinterpret_cast<const float2*>(q + 2 * t);
    };
    auto read_raw = [&](const float* xs, int g) __attribute__((always_inline)) {
        if ((HCQ_DBG & 2) && dbg_started) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) raw[q] = xs[aoff[g] + q * comp_stride];
    };
    using I0 = std::integral_constant<int, 0>;
    using INT = std::integral_constant<int, NT>;
    using INT1 = std::integral_constant<int, NT1>;
    using INT2 = std::integral_constant<int, NT2>;

    // ---- K loop ------------------------------------------------------------------------------------------------
    // MIX: the workgroup of the mixed 8 + 8 tile (descriptor slot 2 in accumulator slot NT1).  Its range-0-only slots are
    // padding; round 2 ran the same program on zero weights for them (a third of such a workgroup's MFMAs), now they are
    // compiled out of its copy of the loop.
    const int nchunks = p.nch * p.nsrc;
    auto kloop = [&](auto mixc) __attribute__((always_inline)) {
    constexpr bool MIX = decltype(mixc)::value;
    using IT0 = std::integral_constant<int, MIX ? NT1 : 0>;
    load_x();
    store_x(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) load_b1(wbase, 0, m, INT{}, IT0{});
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks && !(HCQ_DBG & 4)) load_x();
        const float* xb = lds + buf * buf_floats;
        const float* wc = wbase + (long long)ch * chunk_stride;
        const bool more = ch + 1 < nchunks;
        const float* wn = wc + chunk_stride;
        const float* xs0 = xb + p.half_src[0] * 4 * comp_stride;
        const float* xs1 = xb + p.half_src[1] * 4 * comp_stride;
        read_raw(xs0, 0);
        if (!(HCQ_DBG & 2) || !dbg_started) xforms(raw, gm[0]);
        if ((HCQ_DBG & 2) && !dbg_started) xforms(raw, gm[1]);
        dbg_started = true;
#pragma unroll
        for (int s = 0; s < NR * NG; ++s) {                   // k-groups of the chunk, both ranges
            const int r = s / NG, g = s - r * NG;
            const int pc = r * NPAIR + g / 2;                // fragment pair of this group within the chunk
            const int gst = s & 1;
            const bool last = s + 1 == NR * NG;
            const bool pair_ends = (g & 1) == 1 || g + 1 == NG;   // last k-group that uses the current fragments
            if (!last) {
                const int rn = (s + 1) / NG, gn = (s + 1) - rn * NG;
                read_raw(rn == 0 ? xs0 : xs1, gn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (r == 0) {
#pragma unroll
                    for (int t = MIX ? NT1 : 0; t < NT; ++t)
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(gm[gst][m], (g & 1) ? bfr[m][t].y : bfr[m][t].x,
                                                                          acc[t][m], 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < NT2; ++t)
                        acc[NT1 + t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(gm[gst][m], (g & 1) ? bfr[m][t].y : bfr[m][t].x,
                                                                                acc[NT1 + t][m], 0, 0, 0);
                }
                if (pair_ends) {                              // form m's fragments are free: request the next pair's
                    const int pn = pc + 1;
                    if (pn < NPC) {
                        const int rn = pn / NPAIR, jn = pn - rn * NPAIR;
                        if (rn == 0) load_b1(wc, jn, m, INT{}, IT0{});
                        else load_b1(wc + p.range_stride[0], jn, m, INT2{}, I0{});
                    } else if (more) {
                        load_b1(wn, 0, m, INT{}, IT0{});
                    }
                    // keep the request HERE: between two scheduling barriers the compiler moves loads towards their use
                    // (register pressure), i.e. to the end of this k-group -- a few MFMAs ahead of the wait instead of
                    // seven forms
                    if (HCQ_PIN && (NT >= 3 || HCQ_XDMA)) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!last && !(HCQ_DBG & 2)) xforms(raw, gm[gst ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch + 1 < nchunks && !(HCQ_DBG & 4)) store_x(buf ^ 1);
        if (!(HCQ_DBG & 8)) __syncthreads();
    }
    };
    if constexpr (MX) {
        static_assert(NR == 2 && NT1 == 1 && NT2 == 1, "the only layout with mixed-tile workgroups");
        if (mix_wg) kloop(std::true_type{});
        else kloop(std::false_type{});
    } else {
        kloop(std::false_type{});
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------
    float* const dst = p.dst[set];
    const float* const bias = p.bias[set];
    const float* const addend = p.addend[set];
    float* const stats = p.stats[set];
    const int epi = p.epilogue[set];
    const int grp = fr >> 3;
    const unsigned pos_off = (unsigned)h0 * (unsigned)p.W + (unsigned)(w0 + wave * 16 + fk * 4);
    const unsigned img_off = (unsigned)n_img * (unsigned)p.Cdst * S;
    float* redbuf = lds;                       // the K loop is over (last barrier passed): staging buffers are free
    // PLAIN: no bias / addend / accumulate / statistics (every data gradient, the plain forward): stores only
    auto epilogue = [&](auto plainc) __attribute__((always_inline)) {
        constexpr bool PLAIN = decltype(plainc)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // a mixed-tile workgroup: accumulator slot NT1 is descriptor slot 2, the other slots are padding
            const int ds = mix_wg ? 2 : t;
            const int ob0 = (mix_wg && t != NT1) ? -1 : p.tile_ob[ds][grp];
            const int half = p.tile_half[ds][grp];
            const bool chok = ob0 >= 0;
            const int ob = (chok ? ob0 : 0) + (mix_wg ? 0 : ytile * p.ob_step) + (fr & 7);
            floatx4 c[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h0_ = 0.5f * acc[t][0][r], h1 = 0.5f * acc[t][1][r], h2 = 0.5f * acc[t][2][r], h3 = 0.5f * acc[t][3][r];
                c[0][r] = (h3 - h0_) + (h1 + h2) + acc[t][4][r];
                c[1][r] = (h3 - h0_) - (h1 + h2) + acc[t][5][r];
                c[2][r] = (h3 + h0_) + (h2 - h1) + acc[t][6][r];
                c[3][r] = (h3 + h0_) + (h1 - h2) - acc[t][7][r];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int chn = (half * 4 + q) * p.OB + ob;           // component-major channel index
                const unsigned off = img_off + (unsigned)chn * S + pos_off;
                if (PLAIN) {
                    if (chok) *reinterpret_cast<float4*>(dst + off) = make_float4(c[q][0], c[q][1], c[q][2], c[q][3]);
                    continue;
                }
                float s1 = 0.f, s2 = 0.f;
                if (chok) {
                    const float bv = bias ? bias[chn] : 0.f;
                    float4 o = make_float4(c[q][0] + bv, c[q][1] + bv, c[q][2] + bv, c[q][3] + bv);
                    if (epi & SELD_EPI_ADD) {
                        const float4 ad = *reinterpret_cast<const float4*>(addend + off);
                        o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                    }
                    if (epi & SELD_EPI_ACCUMULATE) {
                        const float4 old = *reinterpret_cast<const float4*>(dst + off);
                        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                    }
                    *reinterpret_cast<float4*>(dst + off) = o;
                    s1 = o.x + o.y + o.z + o.w;
                    s2 = o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
                }
                if (epi & SELD_EPI_STATS) {
                    s1 += __shfl_xor(s1, 16, 64);
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 16, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    if (fk == 0) {
                        const int slot = ((wave * NT + t) * 4 + q) * 16 + fr;
                        redbuf[slot * 2 + 0] = s1;
                        redbuf[slot * 2 + 1] = s2;
                    }
                }
            }
        }
    };
    if (epi == 0 && !bias) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if (epi & SELD_EPI_STATS) {
        __syncthreads();
        float* rep = stats + (size_t)(blockIdx.x % SELD_STATS_REPLICAS) * 2 * p.Cdst;
        for (int e = tid; e < NT * 4 * 16; e += 256) {
            const int fr_ = e & 15, q = (e >> 4) & 3, t = e >> 6;
            const int g_ = fr_ >> 3;
            const int ds = mix_wg ? 2 : t;
            const int ob0 = (mix_wg && t != NT1) ? -1 : p.tile_ob[ds][g_];
            if (ob0 < 0) continue;
            const int chn = (p.tile_half[ds][g_] * 4 + q) * p.OB + ob0 + (mix_wg ? 0 : ytile * p.ob_step) + (fr_ & 7);
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                const int slot = ((wv * NT + t) * 4 + q) * 16 + fr_;
                a1 += redbuf[slot * 2 + 0];
                a2 += redbuf[slot * 2 + 1];
            }
            atomicAdd(rep + chn, a1);
            atomicAdd(rep + p.Cdst + chn, a2);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// First layers (3x3, one or two input block channels): the K loop is ONE short chunk (9 or 18 k of 12 / 20 slots), so
// hcq_conv_kernel above is all prologue and epilogue there (120 MFMAs per workgroup: 688 us for the 8-channel layer at
// batch 32, the block-matrix short-K kernel takes 609).  Here a workgroup keeps its 64 columns and walks R consecutive
// image rows: the R + 2 input rows are staged once (23 / 46 KB), there is ONE barrier, the weight fragments of row r+1
// are requested under the MFMAs of row r, a row's stores drain under the next row's MFMAs and the BatchNorm statistics
// stay in registers until the last row.
// ---------------------------------------------------------------------------------------------------------------------
template <int IBC, int NT1, int NT2, int NR, int R>
__global__ __launch_bounds__(256, 2) void hcq_first_kernel(const HcqP p) {
    constexpr int KH = 3, KW = 3, TAPS = 9;
    constexpr int NT = NT1 + NT2;
    constexpr int KQ = IBC * TAPS;
    constexpr int NG = (KQ + 3) / 4;
    constexpr int NPAIR = (NG + 1) / 2;
    constexpr int NPC = NR * NPAIR;
    constexpr int XR = R + KH - 1;                    // staged input rows
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int A = p.A;
    constexpr int wext = 72, qw = wext >> 2, DPAD = 4;    // dilation 1: halo of 4 each side (the host checks p.wext)

    const unsigned tiles_per_row = (unsigned)p.W >> 6;
    const unsigned hblocks = (unsigned)p.Himg / R;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give an XCD consecutive tiles, so that the
    // 64-column pieces of one 2 KB image row are written back by ONE L2 (spread over eight, every piece was a separate
    // 256-byte DRAM write)
    unsigned b = blockIdx.x;
    if ((gridDim.x & 7u) == 0) b = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int w0 = (int)(b % tiles_per_row) * 64;
    b /= tiles_per_row;
    const int h0 = (int)(b % hblocks) * R;
    const int n_img = (int)(b / hblocks);

    // ---- stage the R + 2 rows of every (component, block channel) once -------------------------------------------
    const unsigned S = (unsigned)(p.Himg * p.W);
    {
        const unsigned src_bytes = (unsigned)p.N * (unsigned)p.Csrc * S * 4u;
        const unsigned OOB = 0xFFFFFFF0u;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, src_bytes, 0x00020000);
        const int items = A * IBC * XR * qw;
        const unsigned img_base = (unsigned)n_img * (unsigned)p.Csrc * S;
        constexpr float inv_qw = 1.0f / (float)qw;
        // four requests in flight per thread, then their four LDS stores (one load -> wait -> store per iteration exposed a
        // first-touch memory latency six to twelve times at the head of every workgroup)
        for (int f0 = tid; f0 < items; f0 += 4 * 256) {
            uintx4h v[4];
            int dst_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + 256 * u;
                const int row = small_div_h(f, inv_qw);               // (comp * IBC + ibl) * XR + xr
                const int quad = f - row * qw;
                const int ci = row / XR;
                const int xr = row - ci * XR;
                const int comp = ci / IBC;
                const int ibl = ci - comp * IBC;
                const int hh = h0 - (KH - 1) / 2 + xr;
                const int ww = w0 - DPAD + 4 * quad;
                const bool ok = f < items && (unsigned)hh < (unsigned)p.Himg && (unsigned)ww < (unsigned)p.W;
                const unsigned e = img_base + ((unsigned)(comp * p.IB + ibl) * (unsigned)p.Himg + (unsigned)hh) * (unsigned)p.W + (unsigned)ww;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? e * 4u : OOB, 0, 0);
                dst_[u] = f < items ? (row * qw + quad) * 4 : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst_[u] >= 0)
                    *reinterpret_cast<float4*>(lds + dst_[u]) =
                        make_float4(__uint_as_float(v[u][0]), __uint_as_float(v[u][1]), __uint_as_float(v[u][2]), __uint_as_float(v[u][3]));
        }
    }

    int aoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int kq = (4 * g + fk) < KQ ? 4 * g + fk : 0;
        const int ibl = kq / TAPS;
        const int tap = kq - ibl * TAPS;
        const int kh = tap / KW;
        const int kw = tap - kh * KW;
        aoff[g] = (ibl * XR + kh) * wext + DPAD + wave * 16 + fr + (kw - (KW - 1) / 2);
    }
    constexpr int comp_stride = IBC * XR * wext;
    const float* const wbase = p.wpack;                 // one channel tile, one weight set, one chunk

    float2 bfr[8][NT];
    float gm[2][8];
    float raw[4];
    auto load_b1 = [&](const float* blk, int j, int m, auto ntrc) __attribute__((always_inline)) {
        constexpr int NTR = decltype(ntrc)::value;
        const float* q = blk + ((long long)(j * 8 + m) * 64 + lane) * (2 * NTR);
#pragma unroll
        for (int t = 0; t < NTR; ++t) bfr[m][t] = *reinterpret_cast<const float2*>(q + 2 * t);
    };
    auto read_raw = [&](const float* xs, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) raw[q] = xs[aoff[g] + q * comp_stride];
    };
    using INT = std::integral_constant<int, NT>;
    using INT2 = std::integral_constant<int, NT2>;

    float* const dst = p.dst[0];
    const float* const bias = p.bias[0];
    const int epi = p.epilogue[0];
    const int grp = fr >> 3;
    const unsigned img_off = (unsigned)n_img * (unsigned)p.Cdst * S;
    // BatchNorm statistics: one slot per (wave, tile, component, LANE) behind the staged rows, read-modify-written row by
    // row by its owner (24 more registers per lane would spill; LDS float atomics doubled the kernel's time)
    float* const sbuf = lds + A * IBC * XR * wext;
    const bool want_stats = (epi & SELD_EPI_STATS) != 0;
    if (want_stats)
        for (int e = tid; e < 4 * NT * 4 * 64 * 2; e += 256) sbuf[e] = 0.f;
    int chbase[NT];                                     // first-component channel of this lane in tile t, -1: padding
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ob0 = p.tile_ob[t][grp];
        chbase[t] = ob0 >= 0 ? p.tile_half[t][grp] * 4 * p.OB + ob0 + (fr & 7) : -1;
    }

#pragma unroll
    for (int m = 0; m < 8; ++m) load_b1(wbase, 0, m, INT{});
    __syncthreads();

#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        floatx4 acc[NT][8];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[t][m] = (floatx4){0.f, 0.f, 0.f, 0.f};
        const bool more = r + 1 < R;
        // the fragments are the same for every row: without this the compiler hoists all 160 registers of them out of
        // the row loop (414 VGPRs wanted)
        const float* wrow = wbase;
        asm volatile("" : "+s"(wrow));
        const float* xs0 = lds + r * wext + p.half_src[0] * 4 * comp_stride;
        const float* xs1 = lds + r * wext + p.half_src[1] * 4 * comp_stride;
        read_raw(xs0, 0);
        xforms(raw, gm[0]);
#pragma unroll
        for (int s = 0; s < NR * NG; ++s) {
            const int rr = s / NG, g = s - rr * NG;
            const int pc = rr * NPAIR + g / 2;
            const int gst = s & 1;
            const bool last = s + 1 == NR * NG;
            const bool pair_ends = (g & 1) == 1 || g + 1 == NG;
            if (!last) {
                const int rn = (s + 1) / NG, gn = (s + 1) - rn * NG;
                read_raw(rn == 0 ? xs0 : xs1, gn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (rr == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(gm[gst][m], (g & 1) ? bfr[m][t].y : bfr[m][t].x,
                                                                          acc[t][m], 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < NT2; ++t)
                        acc[NT1 + t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(gm[gst][m], (g & 1) ? bfr[m][t].y : bfr[m][t].x,
                                                                                acc[NT1 + t][m], 0, 0, 0);
                }
                if (pair_ends) {
                    const int pn = pc + 1;
                    if (pn < NPC) {
                        const int rn = pn / NPAIR, jn = pn - rn * NPAIR;
                        if (rn == 0) load_b1(wrow, jn, m, INT{});
                        else load_b1(wrow + p.range_stride[0], jn, m, INT2{});
                    } else if (more) {
                        load_b1(wrow, 0, m, INT{});               // the next row starts with the same fragments
                    }
                }
            }
            if (!last) xforms(raw, gm[gst ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- this row's results ------------------------------------------------------------------------------------
        const unsigned pos_off = (unsigned)(h0 + r) * (unsigned)p.W + (unsigned)(w0 + wave * 16 + fk * 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            floatx4 c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h0_ = 0.5f * acc[t][0][e], h1 = 0.5f * acc[t][1][e], h2 = 0.5f * acc[t][2][e], h3 = 0.5f * acc[t][3][e];
                c[0][e] = (h3 - h0_) + (h1 + h2) + acc[t][4][e];
                c[1][e] = (h3 - h0_) - (h1 + h2) + acc[t][5][e];
                c[2][e] = (h3 + h0_) + (h2 - h1) + acc[t][6][e];
                c[3][e] = (h3 + h0_) + (h1 - h2) - acc[t][7][e];
            }
            if (chbase[t] >= 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int chn = chbase[t] + q * p.OB;
                    const float bq = bias ? bias[chn] : 0.f;
                    const float4 o = make_float4(c[q][0] + bq, c[q][1] + bq, c[q][2] + bq, c[q][3] + bq);
                    *reinterpret_cast<float4*>(dst + img_off + (unsigned)chn * S + pos_off) = o;
                    if (want_stats) {
                        const int slot = ((wave * NT + t) * 4 + q) * 64 + lane;
                        sbuf[slot] += (o.x + o.y) + (o.z + o.w);                       // the slot is this lane's own
                        sbuf[4 * NT * 4 * 64 + slot] += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);           // one tile's recombination at a time (registers)
        }
    }

    if (want_stats) {
        __syncthreads();
        float* rep = p.stats[0] + (size_t)(blockIdx.x % SELD_STATS_REPLICAS) * 2 * p.Cdst;
        for (int e = tid; e < NT * 4 * 16; e += 256) {
            const int fr_ = e & 15, q = (e >> 4) & 3, t = e >> 6;
            const int g_ = fr_ >> 3;
            const int ob0 = p.tile_ob[t][g_];
            if (ob0 < 0) continue;
            const int chn = (p.tile_half[t][g_] * 4 + q) * p.OB + ob0 + (fr_ & 7);
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv)
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int slot = ((wv * NT + t) * 4 + q) * 64 + k4 * 16 + fr_;
                    a1 += sbuf[slot];
                    a2 += sbuf[4 * NT * 4 * 64 + slot];
                }
            atomicAdd(rep + chn, a1);
            atomicAdd(rep + p.Cdst + chn, a2);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// First stage with the pooling decision fused in: conv -> [BatchNorm2d -> ReLU ->] MaxPool2d(8, 1).  BatchNorm + ReLU is
// z = relu(a y + b) with a = gamma * invstd, and invstd > 0: the SIGN of a is the sign of gamma, known before the batch
// statistics are.  So the maximum of z over a pooling window sits where y is largest (gamma > 0) or smallest (gamma < 0),
// and the convolution kernel -- whose workgroup walks exactly the 8 rows of a window -- can pick that element itself:
// it writes y (the backward pass needs it), the BatchNorm statistics, and per window the chosen raw value + its row.
// A pooled-size kernel (seld_bn_pool_finish) then applies relu(a v + b) once the statistics are final; the 1.6 GB
// read-back of y by bn_relu_pool_fwd disappears.  (Ties: where the window's maximum of z is 0 every position ties and
// torch takes the first; the element chosen here may differ, but ReLU's derivative is 0 there, so no gradient does.
// gamma == 0 keeps row 0, as torch's first-maximum rule does.)
//
// Unlike hcq_first_kernel a wave finishes one channel tile for all 8 rows before the next (the running window maximum of
// three tiles at once does not fit the register file): 8 accumulators instead of 24, the sums G_m recomputed per tile.
// ---------------------------------------------------------------------------------------------------------------------
struct HcqPoolP {
    const float* gamma;          // (Cdst)
    float* pool_raw;             // (N, Cdst, Himg / 8, W): the conv output at the window's arg-max / arg-min row
    unsigned char* idx;          // same shape: that row (0..7)
    // FIN: BatchNorm with KNOWN statistics (first_stage.hip: from the input's second moments), ReLU and the stage's Dropout
    // applied to the window value before it is stored: out = relu(a raw + b) * mask; pool_raw is then written only for
    // channels with gamma == 0 (the one case in which the backward pass cannot recover xhat from out)
    const float* beta;
    const float* mean;
    const float* invstd;
    float* out;
    DropP drop;
};

// WY = false: y is not written and no statistics are gathered (csrc/first_stage.hip: the statistics come from the input's
// second moments, the backward pass never reads y) -- the kernel's only output is the pooled-size window value + row.
template <int IBC, int NT1, int NT2, int NR, int R, bool WY, bool FIN = false>
__global__ __launch_bounds__(256, 3) void hcq_first_pool_kernel(const HcqP p, const HcqPoolP pp) {
    static_assert(!(WY && FIN), "the finishing variant knows the statistics beforehand: it never writes y");
    constexpr int KH = 3, KW = 3, TAPS = 9;
    constexpr int NT = NT1 + NT2;
    constexpr int KQ = IBC * TAPS;
    constexpr int NG = (KQ + 3) / 4;
    constexpr int NPAIR = (NG + 1) / 2;
    constexpr int XR = R + KH - 1;
    static_assert(R == 8, "one workgroup = one MaxPool2d(8, 1) window");
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const int A = p.A;
    constexpr int wext = 72, qw = wext >> 2, DPAD = 4;

    const unsigned tiles_per_row = (unsigned)p.W >> 6;
    const unsigned hblocks = (unsigned)p.Himg / R;
    unsigned b = blockIdx.x;
    if ((gridDim.x & 7u) == 0) b = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);     // XCD-contiguous ranges
    const int w0 = (int)(b % tiles_per_row) * 64;
    b /= tiles_per_row;
    const int hq = (int)(b % hblocks);
    const int h0 = hq * R;
    const int n_img = (int)(b / hblocks);

    const unsigned S = (unsigned)(p.Himg * p.W);
    {
        const unsigned src_bytes = (unsigned)p.N * (unsigned)p.Csrc * S * 4u;
        const unsigned OOB = 0xFFFFFFF0u;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, src_bytes, 0x00020000);
        const int items = A * IBC * XR * qw;
        const unsigned img_base = (unsigned)n_img * (unsigned)p.Csrc * S;
        constexpr float inv_qw = 1.0f / (float)qw;
        // four requests in flight per thread, then their four LDS stores (one load -> wait -> store per iteration exposed a
        // first-touch memory latency six to twelve times at the head of every workgroup)
        for (int f0 = tid; f0 < items; f0 += 4 * 256) {
            uintx4h v[4];
            int dst_[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = f0 + 256 * u;
                const int row = small_div_h(f, inv_qw);               // (comp * IBC + ibl) * XR + xr
                const int quad = f - row * qw;
                const int ci = row / XR;
                const int xr = row - ci * XR;
                const int comp = ci / IBC;
                const int ibl = ci - comp * IBC;
                const int hh = h0 - (KH - 1) / 2 + xr;
                const int ww = w0 - DPAD + 4 * quad;
                const bool ok = f < items && (unsigned)hh < (unsigned)p.Himg && (unsigned)ww < (unsigned)p.W;
                const unsigned e = img_base + ((unsigned)(comp * p.IB + ibl) * (unsigned)p.Himg + (unsigned)hh) * (unsigned)p.W + (unsigned)ww;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? e * 4u : OOB, 0, 0);
                dst_[u] = f < items ? (row * qw + quad) * 4 : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst_[u] >= 0)
                    *reinterpret_cast<float4*>(lds + dst_[u]) =
                        make_float4(__uint_as_float(v[u][0]), __uint_as_float(v[u][1]), __uint_as_float(v[u][2]), __uint_as_float(v[u][3]));
        }
    }
    int aoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int kq = (4 * g + fk) < KQ ? 4 * g + fk : 0;
        const int ibl = kq / TAPS;
        const int tap = kq - ibl * TAPS;
        const int kh = tap / KW;
        const int kw = tap - kh * KW;
        aoff[g] = (ibl * XR + kh) * wext + DPAD + wave * 16 + fr + (kw - (KW - 1) / 2);
    }
    constexpr int comp_stride = IBC * XR * wext;
    float* const sred = lds + A * IBC * XR * wext;      // statistics: [tile][wave][component][16 channels][2]
    // The current tile's weight fragments live in LDS ([pair][form][lane] float2): with them in global memory every row
    // ended on `s_waitcnt vmcnt` for its fragment loads -- and loads and stores retire in order on one counter, so each
    // row also waited for the previous row's 12 stores to reach memory: store time ADDED to compute time (686 us).  With no
    // global load inside the row loop the stores drain under the next rows' MFMAs.
    float2* const wl = reinterpret_cast<float2*>(sred + NT * 4 * 4 * 16 * 2);
    const float* const wbase = p.wpack;
    float* const dst = p.dst[0];
    const float* const bias = p.bias[0];
    const bool want_stats = WY && (p.epilogue[0] & SELD_EPI_STATS) != 0;
    const int grp = fr >> 3;
    const unsigned img_off = (unsigned)n_img * (unsigned)p.Cdst * S;
    const unsigned PS = (unsigned)hblocks * (unsigned)p.W;                   // pooled plane
    const unsigned pool_off = (unsigned)n_img * (unsigned)p.Cdst * PS + (unsigned)hq * (unsigned)p.W +
                              (unsigned)(w0 + wave * 16 + fk * 4);
    __syncthreads();

    float2 bfr[8];
    float gm[2][8];
    float raw[4];
    auto read_raw = [&](const float* xs, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) raw[q] = xs[aoff[g] + q * comp_stride];
    };

    auto tile_body = [&](auto tc) __attribute__((always_inline)) {
        constexpr int T = decltype(tc)::value;
        constexpr bool BOTH = NR == 2 && T >= NT1;            // this tile's channels take the second K range too
        constexpr int NRT = BOTH ? 2 : 1;
        constexpr int NPCT = NRT * NPAIR;
        const int ob0 = p.tile_ob[T][grp];
        const int chb = ob0 >= 0 ? p.tile_half[T][grp] * 4 * p.OB + ob0 + (fr & 7) : -1;
        // fragments of tile T only (range 0 blocks hold NT tiles per (pair, form, lane), range 1 blocks NT2): into LDS
        __syncthreads();                                  // everybody is done with the previous tile's fragments
        {
            // all of a thread's loads first, then its stores: as one load -> wait -> store per iteration the fill exposed
            // 4 / 8 / 8 cache latencies per tile, a fifth of the workgroup's time
            constexpr int NIT = NPCT * 8 * 64 / 256;
            float2 tmp[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + 256 * it;
                const int ln = e & 63, jm = e >> 6;       // jm = pair * 8 + form over both ranges
                const int rr = jm / (NPAIR * 8), jm0 = jm - rr * (NPAIR * 8);
                const float* blk = rr == 0 ? wbase : wbase + p.range_stride[0];
                const int ntr = rr == 0 ? NT : NT2, tt = rr == 0 ? T : T - NT1;
                tmp[it] = *reinterpret_cast<const float2*>(blk + ((long long)jm0 * 64 + ln) * (2 * ntr) + 2 * tt);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) wl[tid + 256 * it] = tmp[it];
        }
        __syncthreads();
        auto load_b1 = [&](int pc, int m) __attribute__((always_inline)) { bfr[m] = wl[(pc * 8 + m) * 64 + lane]; };
        float sg[4], yb[4][4], sb[4][4], s1[4], s2[4], bq[4];     // sb = sg * yb, the key the window maximum is taken over
        unsigned bi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float gq = chb >= 0 ? pp.gamma[chb + q * p.OB] : 1.f;
            bq[q] = (bias && chb >= 0) ? bias[chb + q * p.OB] : 0.f;
            sg[q] = gq > 0.f ? 1.f : (gq < 0.f ? -1.f : 0.f);
            s1[q] = 0.f; s2[q] = 0.f; bi[q] = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) { yb[q][e] = 0.f; sb[q][e] = -INFINITY; }
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) load_b1(0, m);
#pragma unroll 1
        for (int r = 0; r < R; ++r) {
            floatx4 acc[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = (floatx4){0.f, 0.f, 0.f, 0.f};
            const bool more = r + 1 < R;
            const float* xs0 = lds + r * wext + p.half_src[0] * 4 * comp_stride;
            const float* xs1 = lds + r * wext + p.half_src[1] * 4 * comp_stride;
            read_raw(xs0, 0);
            xforms(raw, gm[0]);
#pragma unroll
            for (int s = 0; s < NRT * NG; ++s) {
                const int rr = s / NG, g = s - rr * NG;
                const int pc = rr * NPAIR + g / 2;
                const int gst = s & 1;
                const bool last = s + 1 == NRT * NG;
                const bool pair_ends = (g & 1) == 1 || g + 1 == NG;
                if (!last) {
                    const int rn = (s + 1) / NG, gn = (s + 1) - rn * NG;
                    read_raw(rn == 0 ? xs0 : xs1, gn);
                }
                __builtin_amdgcn_sched_barrier(0);      // (without the two barriers of a group: 710 us against 670-690)
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(gm[gst][m], (g & 1) ? bfr[m].y : bfr[m].x, acc[m], 0, 0, 0);
                    if (pair_ends) {
                        const int pn = pc + 1;
                        if (pn < NPCT) load_b1(pn, m);
                        else if (more) load_b1(0, m);
                    }
                }
                if (!last) xforms(raw, gm[gst ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- this row of this tile: components, store, statistics, window maximum ------------------------------
            floatx4 c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h0_ = 0.5f * acc[0][e], h1 = 0.5f * acc[1][e], h2 = 0.5f * acc[2][e], h3 = 0.5f * acc[3][e];
                c[0][e] = (h3 - h0_) + (h1 + h2) + acc[4][e];
                c[1][e] = (h3 - h0_) - (h1 + h2) + acc[5][e];
                c[2][e] = (h3 + h0_) + (h2 - h1) + acc[6][e];
                c[3][e] = (h3 + h0_) + (h1 - h2) - acc[7][e];
            }
            if (chb >= 0) {
                const unsigned pos_off = (unsigned)(h0 + r) * (unsigned)p.W + (unsigned)(w0 + wave * 16 + fk * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int chn = chb + q * p.OB;
                    float o[4] = {c[q][0], c[q][1], c[q][2], c[q][3]};
                    if (bias) { o[0] += bq[q]; o[1] += bq[q]; o[2] += bq[q]; o[3] += bq[q]; }
                    if constexpr (WY) {
                        *reinterpret_cast<float4*>(dst + img_off + (unsigned)chn * S + pos_off) = make_float4(o[0], o[1], o[2], o[3]);
                        s1[q] += (o[0] + o[1]) + (o[2] + o[3]);
                        s2[q] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // first maximum wins, NaN propagates: "not (key <= best)" is true for a larger key and for NaN, and
                        // for row 0 (best = -inf); gamma == 0 makes every key 0, so row 0 stays
                        const float so = sg[q] * o[e];
                        const bool take = !(so <= sb[q][e]);
                        yb[q][e] = take ? o[e] : yb[q][e];
                        sb[q][e] = take ? so : sb[q][e];
                        bi[q] = take ? ((bi[q] & ~(0xFFu << (8 * e))) | ((unsigned)r << (8 * e))) : bi[q];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (chb >= 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned po = pool_off + (unsigned)(chb + q * p.OB) * PS;
                *reinterpret_cast<unsigned*>(pp.idx + po) = bi[q];
                if constexpr (FIN) {
                    const int chn = chb + q * p.OB;
                    const float gq = pp.gamma[chn];
                    const float a = gq * pp.invstd[chn], bb = pp.beta[chn] - pp.mean[chn] * a;
                    float z[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = yb[q][e];
                        z[e] = fmaxf(v * a + bb, 0.f);
                        if (v != v) z[e] = v;                                   // NaN propagates (torch's relu / max_pool)
                    }
                    if (pp.drop.p > 0.f) {       // the mask seld_dropout_fwd draws for element group po / 4 of `out`
                        const uint64_t off = pp.drop.offset + (pp.drop.state ? pp.drop.state[0] : 0ull);
                        const float4 mk = dropout_mask4(off + (uint64_t)(po >> 2), pp.drop.seed, pp.drop.p, pp.drop.scale);
                        z[0] = mk.x != 0.f ? z[0] * mk.x : 0.f;
                        z[1] = mk.y != 0.f ? z[1] * mk.y : 0.f;
                        z[2] = mk.z != 0.f ? z[2] * mk.z : 0.f;
                        z[3] = mk.w != 0.f ? z[3] * mk.w : 0.f;
                    }
                    *reinterpret_cast<float4*>(pp.out + po) = make_float4(z[0], z[1], z[2], z[3]);
                    if (gq == 0.f) *reinterpret_cast<float4*>(pp.pool_raw + po) = make_float4(yb[q][0], yb[q][1], yb[q][2], yb[q][3]);
                } else {
                    *reinterpret_cast<float4*>(pp.pool_raw + po) = make_float4(yb[q][0], yb[q][1], yb[q][2], yb[q][3]);
                }
            }
        }
        if (want_stats) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a1 = s1[q], a2 = s2[q];
                a1 += __shfl_xor(a1, 16, 64);
                a1 += __shfl_xor(a1, 32, 64);
                a2 += __shfl_xor(a2, 16, 64);
                a2 += __shfl_xor(a2, 32, 64);
                if (fk == 0) {
                    const int slot = ((T * 4 + wave) * 4 + q) * 16 + fr;
                    sred[slot * 2 + 0] = a1;
                    sred[slot * 2 + 1] = a2;
                }
            }
        }
    };
    tile_body(std::integral_constant<int, 0>{});
    if constexpr (NT > 1) tile_body(std::integral_constant<int, 1>{});
    if constexpr (NT > 2) tile_body(std::integral_constant<int, 2>{});

    if (want_stats) {
        __syncthreads();
        float* rep = p.stats[0] + (size_t)(blockIdx.x % SELD_STATS_REPLICAS) * 2 * p.Cdst;
        for (int e = tid; e < NT * 4 * 16; e += 256) {
            const int fr_ = e & 15, q = (e >> 4) & 3, t = e >> 6;
            const int g_ = fr_ >> 3;
            const int ob0 = p.tile_ob[t][g_];
            if (ob0 < 0) continue;
            const int chn = (p.tile_half[t][g_] * 4 + q) * p.OB + ob0 + (fr_ & 7);
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) {
                const int slot = ((t * 4 + wv) * 4 + q) * 16 + fr_;
                a1 += sred[slot * 2 + 0];
                a2 += sred[slot * 2 + 1];
            }
            atomicAdd(rep + chn, a1);
            atomicAdd(rep + p.Cdst + chn, a2);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight forms in fragment order.  One thread per packed float.
//   mode 0 (forward):        dst block channel = conv output block channel, K runs over (conv input block channel, tap)
//   mode 1 (data gradient):  dst = conv INPUT block channel, K over (conv output block channel, flipped tap), conjugate
// Layout: [weight set (forward pairs)] x { regular channel tiles [ytile][source (gradient pairs)][chunk][range][pair]
// [m][lane][tile][2 groups], the mixed channel tile (if any) last, same layout }.
// ---------------------------------------------------------------------------------------------------------------------
struct HcqPackP {
    WPtrs w[2];                  // component tensors (OA, IA, taps) of the 1 or 2 weight sets
    float* out;
    int A, mode;
    int OA, IA, taps;            // of the CONVOLUTION (not swapped for the data gradient)
    int IBC, nch, NR, NT1, NT2, NG, NPAIR;
    int half_src[2];
    int nreg;                    // regular channel tiles per set
    int has_mix;
    int nsets;                   // forward pairs: separate outputs
    int nsrc;                    // gradient pairs: two (source, weight set) along K
    int ob_step;
    int tile_half[3][2], tile_ob[3][2];
    long long range_stride[2], ytile_stride, set_stride, total;
};

__device__ __forceinline__ float hcq_pack_value(const HcqPackP& p, long long idx) {
    // 32-bit index arithmetic (the host keeps every buffer below 2^31 floats): the 64-bit divisions this decode started
    // with were most of the kernel's 60 us
    const int NT = p.NT1 + p.NT2;
    const unsigned set_stride = (unsigned)p.set_stride, ytile_stride = (unsigned)p.ytile_stride;
    const unsigned set = (unsigned)idx / set_stride;
    unsigned r0 = (unsigned)idx - set * set_stride;
    const int ytile = (int)(r0 / ytile_stride);
    r0 -= (unsigned)ytile * ytile_stride;
    const bool mix = p.has_mix && ytile == p.nreg;          // the mixed channel tile: same layout, other descriptors
    const unsigned rs[2] = {(unsigned)p.range_stride[0], (unsigned)p.range_stride[1]};
    const unsigned chunk_stride = rs[0] + (p.NR > 1 ? rs[1] : 0);
    const int chs = (int)(r0 / chunk_stride);            // chunk over all sources
    r0 -= (unsigned)chs * chunk_stride;
    const int srcsel = chs / p.nch, ch = chs - srcsel * p.nch;
    const int range = (r0 >= rs[0]) ? 1 : 0;
    if (range) r0 -= rs[0];
    const int ntr = range ? p.NT2 : NT;
    const int per_m = 64 * 2 * ntr;
    const int per_pair = 8 * per_m;
    const int j = (int)(r0 / (unsigned)per_pair);
    int r1 = (int)(r0 - (unsigned)j * (unsigned)per_pair);
    const int m = r1 / per_m;
    r1 -= m * per_m;
    const int lane = r1 / (2 * ntr);
    r1 -= lane * 2 * ntr;
    const int t = r1 >> 1, gg = r1 & 1;
    const int k = lane >> 4, n = lane & 15;
    const int g = 2 * j + gg;
    if (g >= p.NG) return 0.f;
    const int kq = 4 * g + k;
    if (kq >= p.IBC * p.taps) return 0.f;                  // padding of the last k-group
    const int tf = range ? p.NT1 + t : t;                  // tile (accumulator slot) of the workgroup
    const int ds = mix ? 2 : tf;                           // descriptor slot: a mixed workgroup's slot NT1 is the mixed
    const int grp = n >> 3;                                // tile, its other slots are padding
    const int ob0 = (mix && tf != p.NT1) ? -1 : p.tile_ob[ds][grp];
    if (ob0 < 0) return 0.f;
    const int dblk = ob0 + (mix ? 0 : ytile * p.ob_step) + (n & 7);     // destination block channel
    const int kbl = kq / p.taps;
    const int tap = kq - kbl * p.taps;
    const int sblk = ch * p.IBC + kbl;                                  // source block channel
    const int hd = p.tile_half[ds][grp], hs = p.half_src[range];
    // which quaternion of the dual quaternion couples (source half hs) -> (destination half hd)
    int qsel = 0;                                                        // 0: Q, 1: Q2, -1: structural zero
    if (p.A == 8) {
        if (p.mode == 0) qsel = (hd == hs) ? 0 : (hd == 1 ? 1 : -1);     // y_p = Q x_p; y_d = Q2 x_p + Q x_d
        else qsel = (hd == hs) ? 0 : (hd == 0 ? 1 : -1);                 // dx_p = Q^ dy_p + Q2^ dy_d; dx_d = Q^ dy_d
    }
    if (qsel < 0) return 0.f;
    const int co = p.mode == 0 ? dblk : sblk;           // conv output / input block channel of the element
    const int ci = p.mode == 0 ? sblk : dblk;
    const int tp = p.mode == 0 ? tap : p.taps - 1 - tap;
    if (co >= p.OA || ci >= p.IA) return 0.f;
    const size_t e = ((size_t)co * p.IA + ci) * p.taps + tp;
    const WPtrs& w = p.w[p.mode == 0 ? set : srcsel];
    float a[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = w.p[qsel * 4 + c][e];
    if (p.mode == 1) { a[1] = -a[1]; a[2] = -a[2]; a[3] = -a[3]; }      // conjugate
    switch (m) {
        case 0: return a[3] + a[1];
        case 1: return a[0] - a[2];
        case 2: return a[0] + a[2];
        case 3: return a[3] - a[1];
        case 4: return a[3] - a[2];
        case 5: return a[1] + a[0];
        case 6: return a[0] - a[1];
        default: return a[3] + a[2];
    }
}

__global__ __launch_bounds__(256) void hcq_pack_kernel(const HcqPackP p) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx < p.total) p.out[idx] = hcq_pack_value(p, idx);
}

// Every registered layer in ONE launch (once per optimiser step), balanced: blockIdx.x runs over the 256-float blocks of
// ALL entries back to back; starts[e] = first block of entry e (starts[nentries] = total).  (A (blocks, entries) grid sized
// for the largest entry launched mostly empty workgroups, a capped one looped: 60 us either way.)
__global__ __launch_bounds__(256) void hcq_pack_flat_kernel(const HcqPackP* __restrict__ table, const int* __restrict__ starts,
                                                            int nentries) {
    const int b = blockIdx.x;
    int lo = 0, hi = nentries;                       // last entry with starts[e] <= b
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (starts[mid] <= b) lo = mid; else hi = mid;
    }
    const HcqPackP& p = table[lo];
    const long long idx = (long long)(b - starts[lo]) * 256 + threadIdx.x;
    if (idx < p.total) p.out[idx] = hcq_pack_value(p, idx);
}

// (the (blocks, entries) form: blockIdx.y = table entry, blockIdx.x = 256-float block of that entry's buffer)
__global__ __launch_bounds__(256) void hcq_pack_table_kernel(const HcqPackP* __restrict__ table, int nentries) {
    const int e = blockIdx.y;
    if (e >= nentries) return;
    const HcqPackP& p = table[e];
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < p.total; idx += (long long)gridDim.x * 256)
        p.out[idx] = hcq_pack_value(p, idx);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
struct HcqPlan {
    int ok;
    int KH, KW, IBC, NT1, NT2, NR, XI, mix;
    int first_rows;              // > 0: hcq_first_kernel walks this many image rows per workgroup
    HcqP kp;
    HcqPackP pp;
    size_t pack_floats;
    dim3 grid;
    size_t smem;
};

// mode 0 forward (npair: 1 or 2 convolutions of the same input -> separate outputs),
// mode 1 data gradient (npair: 1, or 2 = sum of the gradients of two convolutions w.r.t. their common input).
// mode 2 = mode 0 for the pooling first-stage kernel (hcq_first_pool_kernel): same packed forms, and the 8-channel
// dual-quaternion layer is NOT left to the short-K kernel.
static HcqPlan hcq_plan(const seld_conv_desc* d, int mode, int npair) {
    HcqPlan pl{};
    const bool pool = mode == 2;
    if (pool) mode = 0;
    const int A = d->algebra;
    if (A != 4 && A != 8) return pl;
    if (d->stride[0] != 1 || d->stride[1] != 1 || d->dil[0] != 1) return pl;
    int o[2];
    hc_out_shape(d, o);
    if (o[0] != d->in[0] || o[1] != d->in[1]) return pl;                     // 'same' convolutions only
    const int KH = d->k[0], KW = d->k[1];
    if (!((KH == 1 && (KW == 1 || KW == 3)) || (KH == 3 && KW == 3))) return pl;
    if (2 * d->pad[1] != d->dil[1] * (KW - 1) || 2 * d->pad[0] != (KH - 1)) return pl;
    if (KW == 1 && d->dil[1] != 1) return pl;
    const int W = d->in[1], Himg = d->in[0];
    if (W % 64) return pl;
    const int Csrc = mode == 0 ? d->Cin : d->Cout, Cdst = mode == 0 ? d->Cout : d->Cin;
    const int IB = Csrc / A, OB = Cdst / A;
    if ((long long)d->N * Csrc * Himg * W * 4 >= 0xFFFFFFF0ll || (long long)d->N * Cdst * Himg * W * 4 >= 0xFFFFFFF0ll) return pl;
    const int taps = KH * KW;
    const int nsets = mode == 0 ? npair : 1, nsrc = mode == 1 ? npair : 1;
    const int dil = KW == 3 ? d->dil[1] : 0;
    const int dpad = KW == 3 ? (dil + 3) / 4 * 4 : 0;
    const int wext = 64 + 2 * dpad;
    // K chunk: candidates in order of preference; two workgroups per CU must fit (78 KB each)
    static const int cand11[] = {16, 24, 8, 0}, cand13[] = {8, 4, 0}, cand33[] = {4, 2, 1, 0};
    const int* cand = taps == 1 ? cand11 : (taps == 3 ? cand13 : cand33);
    int IBC = 0;
    size_t smem = 0;
    for (int i = 0; cand[i]; ++i) {
        if (IB % cand[i]) continue;
        const int nbuf = (IB / cand[i]) * nsrc > 1 ? 2 : 1;
        const size_t need = (size_t)nbuf * A * cand[i] * KH * wext * sizeof(float);
        if (need <= 78 * 1024) { IBC = cand[i]; smem = need; break; }
    }
    if (!IBC) return pl;
    // The 8-channel dual-quaternion first layer (ONE block channel: 9 of 12 k-slots used) stays on the persistent short-K
    // kernel hc_conv_smallk_kernel<12,3,3,4,18,9>: 610 us (530 inside the step) against 629 (571) for hcq_first_kernel
    // and 688 for hcq_conv_kernel at batch 32; 280 / 287 / 326 at 16.  With two block channels the fast product wins:
    // config 4's 16-channel layer (which does not fit that kernel's LDS) 1303 -> 405 us, config 2's quaternion layer
    // 342 -> 225.
    if (!pool && mode == 0 && taps == 9 && A == 8 && IB == 1 && Cdst == 192 && (long long)d->N * Himg * W >= 256 * 128 &&
        !env().conv_no_smallk) return pl;
    const int rows = A * IBC * KH;
    const int items = rows * (wext / 4);
    const int XI = (items + 255) / 256;
    if (smem < 8 * 1024) smem = 8 * 1024;                                     // statistics scratch of the epilogue
    // channel tiles
    HcqP& k = pl.kp;
    int NT1, NT2, NR, nreg, ob_step, mix = 0;
    for (int t = 0; t < 3; ++t)
        for (int g = 0; g < 2; ++g) { k.tile_half[t][g] = 0; k.tile_ob[t][g] = -1; }
    if (A == 8) {
        NR = 2;
        // forward: range 0 = primal source (both destination halves), range 1 = dual source (dual destination only)
        // gradient: range 0 = dual source (both halves), range 1 = primal source (primal destination only)
        const int both = mode == 0 ? 1 : 0;           // destination half active in both ranges
        const int once = 1 - both;
        k.half_src[0] = mode == 0 ? 0 : 1;
        k.half_src[1] = 1 - k.half_src[0];
        if (OB % 16 != 0 && OB % 16 != 8) return pl;
        NT1 = 1; NT2 = 1; nreg = OB / 16; ob_step = 16;
        k.tile_half[0][0] = k.tile_half[0][1] = once; k.tile_ob[0][0] = 0; k.tile_ob[0][1] = 8;
        k.tile_half[1][0] = k.tile_half[1][1] = both; k.tile_ob[1][0] = 0; k.tile_ob[1][1] = 8;
        if (OB % 16 == 8) {                           // the last 8 block channels of both halves share one tile
            k.tile_half[2][0] = once; k.tile_ob[2][0] = OB - 8;
            k.tile_half[2][1] = both; k.tile_ob[2][1] = OB - 8;
            // 24 block channels: either ONE workgroup per position tile with three tiles (24 accumulators, the input
            // staged once: cnn.1 949 us against 1139), or the mixed tile in workgroups of its own (twice the
            // workgroups for layers with few position tiles: TCN data gradient 47.5 us against 68.6).  A forward PAIR
            // brings its own factor of two (two weight sets): skip || residual at batch 32 (256 position tiles) 39.8 ->
            // 32.3 us as 512 three-tile workgroups instead of 1024 two-tile ones, a third of them half padding
            const long long ptiles = (long long)d->N * Himg * W / 64;
            if (OB == 24 && ptiles * nsets >= 512) { NT2 = 2; ob_step = 0; }
            else mix = 1;
        }
        if (nreg == 0) return pl;
    } else {
        NR = 1; NT2 = 0;
        k.half_src[0] = k.half_src[1] = 0;
        if (OB % 32 == 0) { NT1 = 2; nreg = OB / 32; ob_step = 32; }
        else if (OB % 16 == 0) { NT1 = 1; nreg = OB / 16; ob_step = 16; }
        else return pl;
        for (int t = 0; t < NT1; ++t) { k.tile_ob[t][0] = 16 * t; k.tile_ob[t][1] = 16 * t + 8; }
    }
    const int NG = (IBC * taps + 3) / 4, NPAIR = (NG + 1) / 2;
    const int NT = NT1 + NT2;
    k.A = A; k.N = d->N; k.Csrc = Csrc; k.Cdst = Cdst; k.IB = IB; k.OB = OB;
    k.W = W; k.Himg = Himg; k.dil = dil; k.dpad = dpad; k.wext = wext; k.nch = IB / IBC;
    k.nsrc = nsrc;
    k.ytiles = nreg + mix; k.mix_ytile = mix ? nreg : -1; k.ob_step = ob_step;
    k.range_stride[0] = (long long)NPAIR * 8 * 64 * 2 * NT;
    k.range_stride[1] = (long long)NPAIR * 8 * 64 * 2 * NT2;
    k.ytile_stride = (long long)nsrc * k.nch * (k.range_stride[0] + (NR > 1 ? k.range_stride[1] : 0));
    k.set_stride = (long long)(nreg + mix) * k.ytile_stride;
    pl.pack_floats = (size_t)k.set_stride * nsets;
    HcqPackP& q = pl.pp;
    q.A = A; q.mode = mode; q.OA = d->Cout / A; q.IA = d->Cin / A; q.taps = taps;
    q.IBC = IBC; q.nch = k.nch; q.NR = NR; q.NT1 = NT1; q.NT2 = NT2; q.NG = NG; q.NPAIR = NPAIR;
    q.half_src[0] = k.half_src[0]; q.half_src[1] = k.half_src[1];
    q.nreg = nreg; q.has_mix = mix; q.nsets = nsets; q.nsrc = nsrc; q.ob_step = ob_step;
    for (int t = 0; t < 3; ++t)
        for (int g = 0; g < 2; ++g) { q.tile_half[t][g] = k.tile_half[t][g]; q.tile_ob[t][g] = k.tile_ob[t][g]; }
    q.range_stride[0] = k.range_stride[0]; q.range_stride[1] = k.range_stride[1];
    q.ytile_stride = k.ytile_stride; q.set_stride = k.set_stride; q.total = (long long)pl.pack_floats;
    pl.KH = KH; pl.KW = KW; pl.IBC = IBC; pl.NT1 = NT1; pl.NT2 = NT2; pl.NR = NR; pl.XI = XI; pl.mix = mix;
    const long long ptot = (long long)d->N * Himg * W;
    pl.grid = dim3((unsigned)(ptot / 64), (unsigned)(k.ytiles * nsets), 1);
    pl.smem = smem;
    // first layers: one chunk, one channel tile, one weight set -> the row-walking kernel
    constexpr int FIRST_R = 8;
    if (mode == 0 && taps == 9 && IBC == IB && IB <= 2 && k.ytiles == 1 && nsets == 1 && !mix && Himg % FIRST_R == 0 && wext == 72 &&
        !env().hcq_no_first) {
        pl.first_rows = FIRST_R;
        pl.grid = dim3((unsigned)(ptot / 64 / FIRST_R), 1, 1);
        pl.smem = ((size_t)A * IBC * (FIRST_R + 2) * wext + 4 * NT * 4 * 64 * 2) * sizeof(float);   // rows + statistics slots
    }
    if (pool && (!pl.first_rows || env().hcq_no_pool)) return HcqPlan{};
    pl.ok = 1;
    return pl;
}

// The instantiation a plan runs: staging items per thread are fixed by (taps, IBC, algebra) except for the dilated 1x3
// layers (halo width).  Returns 0 if there is none.
struct HcqKern { int KH, KW, IBC, NT1, NT2, NR, XI; };
static int hcq_pick(const HcqPlan& pl, HcqKern* k) {
    int xi8 = 0, xi4 = 0;
    if (pl.KH == 1 && pl.KW == 3 && pl.IBC == 8) {
        xi8 = pl.XI <= 5 ? 5 : (pl.XI <= 7 ? 7 : 9);
        xi4 = pl.XI <= 3 ? 3 : 5;
    } else if (pl.KH == 1 && pl.KW == 3 && pl.IBC == 4) { xi8 = 6; xi4 = 3; }
    else if (pl.KH == 1 && pl.KW == 1 && pl.IBC == 16) { xi8 = 8; xi4 = 4; }
    else if (pl.KH == 1 && pl.KW == 1 && pl.IBC == 24) { xi8 = 12; xi4 = 6; }
    else if (pl.KH == 1 && pl.KW == 1 && pl.IBC == 8) { xi8 = 4; xi4 = 2; }
    else if (pl.KH == 3 && pl.KW == 3 && pl.IBC == 4) { xi8 = 7; xi4 = 4; }
    else if (pl.KH == 3 && pl.KW == 3 && pl.IBC == 2) { xi8 = 4; xi4 = 2; }
    else if (pl.KH == 3 && pl.KW == 3 && pl.IBC == 1) { xi8 = 2; xi4 = 1; }
    else return 0;
    const int xi = pl.NR == 2 ? xi8 : xi4;
    if (pl.XI > xi) return 0;
    *k = HcqKern{pl.KH, pl.KW, pl.IBC, pl.NT1, pl.NT2, pl.NR, xi};
    return 1;
}

template <int KH, int KW, int IBC, int NT1, int NT2, int NR, int XI, bool MX = false>
static int hcq_launch_one(const HcqPlan& pl, hipStream_t st) {
    auto kern = hcq_conv_kernel<KH, KW, IBC, NT1, NT2, NR, XI, MX>;
    if (pl.smem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem) != hipSuccess)
        return SELD_ELAUNCH;
    hipLaunchKernelGGL(kern, pl.grid, dim3(256), pl.smem, st, pl.kp);
    return check_launch();
}

template <int KH, int KW, int IBC, int XI8, int XI4>
static int hcq_launch_cfg(const HcqPlan& pl, const HcqKern& k, hipStream_t st) {
    if (k.NR == 2) {
        if (k.XI != XI8) return SELD_EUNSUPPORTED;
        if (k.NT2 == 2) return hcq_launch_one<KH, KW, IBC, 1, 2, 2, XI8>(pl, st);
        if (pl.mix) return hcq_launch_one<KH, KW, IBC, 1, 1, 2, XI8, true>(pl, st);
        return hcq_launch_one<KH, KW, IBC, 1, 1, 2, XI8>(pl, st);
    }
    if (k.XI != XI4) return SELD_EUNSUPPORTED;
    return k.NT1 == 2 ? hcq_launch_one<KH, KW, IBC, 2, 0, 1, XI4>(pl, st)
                      : hcq_launch_one<KH, KW, IBC, 1, 0, 1, XI4>(pl, st);
}

template <int IBC, int NT1, int NT2, int NR>
static int hcq_launch_first(const HcqPlan& pl, hipStream_t st) {
    auto kern = hcq_first_kernel<IBC, NT1, NT2, NR, 8>;
    if (pl.smem > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.smem) != hipSuccess)
        return SELD_ELAUNCH;
    hipLaunchKernelGGL(kern, pl.grid, dim3(256), pl.smem, st, pl.kp);
    return check_launch();
}

// The row-walking first-layer kernel takes no addend / accumulate epilogue (the first layer has none).
static bool hcq_first_takes(const HcqPlan& pl) {
    if (!pl.first_rows || pl.first_rows != 8) return false;
    if (pl.kp.epilogue[0] & ~SELD_EPI_STATS) return false;
    if (pl.NR == 2) return pl.NT1 == 1 && (pl.NT2 == 2 || pl.NT2 == 1);
    return pl.NT2 == 0 && (pl.NT1 == 1 || pl.NT1 == 2);
}

static int hcq_launch(const HcqPlan& pl, hipStream_t st) {
    HcqKern k;
    if (!hcq_pick(pl, &k)) return SELD_EUNSUPPORTED;
    if (hcq_first_takes(pl)) {
        if (k.IBC == 1) {
            if (k.NR == 2) return k.NT2 == 2 ? hcq_launch_first<1, 1, 2, 2>(pl, st) : hcq_launch_first<1, 1, 1, 2>(pl, st);
            return k.NT1 == 2 ? hcq_launch_first<1, 2, 0, 1>(pl, st) : hcq_launch_first<1, 1, 0, 1>(pl, st);
        }
        if (k.NR == 2) return k.NT2 == 2 ? hcq_launch_first<2, 1, 2, 2>(pl, st) : hcq_launch_first<2, 1, 1, 2>(pl, st);
        return k.NT1 == 2 ? hcq_launch_first<2, 2, 0, 1>(pl, st) : hcq_launch_first<2, 1, 0, 1>(pl, st);
    }
    if (k.KW == 3 && k.KH == 1 && k.IBC == 8) {
        if (k.NR == 2) {
            if (k.XI == 5) return hcq_launch_cfg<1, 3, 8, 5, 3>(pl, k, st);
            if (k.XI == 7) return hcq_launch_cfg<1, 3, 8, 7, 3>(pl, k, st);
            return hcq_launch_cfg<1, 3, 8, 9, 3>(pl, k, st);
        }
        if (k.XI == 3) return hcq_launch_cfg<1, 3, 8, 5, 3>(pl, k, st);
        return hcq_launch_cfg<1, 3, 8, 7, 5>(pl, k, st);
    }
    if (k.KH == 1 && k.KW == 3) return hcq_launch_cfg<1, 3, 4, 6, 3>(pl, k, st);
    if (k.KH == 1 && k.IBC == 16) return hcq_launch_cfg<1, 1, 16, 8, 4>(pl, k, st);
    if (k.KH == 1 && k.IBC == 24) return hcq_launch_cfg<1, 1, 24, 12, 6>(pl, k, st);
    if (k.KH == 1 && k.IBC == 8) return hcq_launch_cfg<1, 1, 8, 4, 2>(pl, k, st);
    if (k.IBC == 2) return hcq_launch_cfg<3, 3, 2, 4, 2>(pl, k, st);
    if (k.IBC == 1) return hcq_launch_cfg<3, 3, 1, 2, 1>(pl, k, st);
    return hcq_launch_cfg<3, 3, 4, 7, 4>(pl, k, st);
}

}  // namespace seld

using namespace seld;

extern "C" size_t seld_hcq_pack_floats(const seld_conv_desc* d, int32_t mode, int32_t npair) {
    if (hc_validate(d) != SELD_OK || mode < 0 || mode > 2 || npair < 1 || npair > 2) return 0;
    if (env().conv_no_hcq) return 0;
    const HcqPlan pl = hcq_plan(d, mode, npair);
    HcqKern k;
    return (pl.ok && hcq_pick(pl, &k)) ? pl.pack_floats : 0;
}

extern "C" size_t seld_hcq_pack_entry_bytes(void) { return sizeof(HcqPackP); }

/* Kernel symbol (as rocprofv3 prints it) that seld_hcq_conv launches for (desc, mode, npair). */
extern "C" int seld_hcq_kernel_label(const seld_conv_desc* d, int32_t mode, int32_t npair, char* buf, int32_t buflen) {
    if (hc_validate(d) != SELD_OK || !buf || buflen < 64) return SELD_EINVAL;
    const HcqPlan pl = hcq_plan(d, mode, npair);
    HcqKern k;
    if (!pl.ok || !hcq_pick(pl, &k)) return SELD_EUNSUPPORTED;
    if (mode == 2)
        snprintf(buf, buflen, "hcq_first_pool_kernel<%d, %d, %d, %d, 8>", k.IBC, k.NT1, k.NT2, k.NR);
    else if (pl.first_rows && hcq_first_takes(pl))   // (label of the plain / statistics epilogue: what the first layer runs)
        snprintf(buf, buflen, "hcq_first_kernel<%d, %d, %d, %d, %d>", k.IBC, k.NT1, k.NT2, k.NR, pl.first_rows);
    else
        snprintf(buf, buflen, "hcq_conv_kernel<%d, %d, %d, %d, %d, %d, %d, %s>", k.KH, k.KW, k.IBC, k.NT1, k.NT2, k.NR, k.XI,
                 (pl.mix && k.NR == 2 && k.NT2 == 1) ? "true" : "false");
    return SELD_OK;
}

/* Fill one table entry (host memory, seld_hcq_pack_entry_bytes() bytes) for seld_hcq_pack_table. */
extern "C" int seld_hcq_pack_entry(const seld_conv_desc* d, int32_t mode, int32_t npair, const float* const wA[8],
                                   const float* const wB[8], float* wpack, void* entry) {
    if (hc_validate(d) != SELD_OK || !wA || !wpack || !entry) return SELD_EINVAL;
    if (mode < 0 || mode > 2 || npair < 1 || npair > 2 || (npair == 2 && !wB)) return SELD_EINVAL;
    HcqPlan pl = hcq_plan(d, mode, npair);
    if (!pl.ok) return SELD_EUNSUPPORTED;
    for (int i = 0; i < 8; ++i) {
        pl.pp.w[0].p[i] = i < d->algebra ? wA[i] : nullptr;
        pl.pp.w[1].p[i] = (npair == 2 && i < d->algebra) ? wB[i] : nullptr;
    }
    pl.pp.out = wpack;
    memcpy(entry, &pl.pp, sizeof(HcqPackP));
    return SELD_OK;
}

extern "C" int seld_hcq_pack(const seld_conv_desc* d, int32_t mode, int32_t npair, const float* const wA[8],
                             const float* const wB[8], float* wpack, void* stream) {
    HcqPackP e;
    const int rc = seld_hcq_pack_entry(d, mode, npair, wA, wB, wpack, &e);
    if (rc != SELD_OK) return rc;
    const unsigned blocks = (unsigned)((e.total + 255) / 256);
    hipLaunchKernelGGL(hcq_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, e);
    return check_launch();
}

/* All entries of a DEVICE table in one launch; max_floats = the largest entry's float count. */
extern "C" int seld_hcq_pack_table(const void* table_dev, int32_t nentries, int64_t max_floats, void* stream) {
    if (!table_dev || nentries <= 0 || max_floats <= 0) return SELD_EINVAL;
    long long bx = (max_floats + 255) / 256;
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(hcq_pack_table_kernel, dim3((unsigned)bx, (unsigned)nentries), dim3(256), 0, (hipStream_t)stream,
                       (const HcqPackP*)table_dev, nentries);
    return check_launch();
}

/* The balanced form: starts_dev[e] = first 256-float block of entry e, starts_dev[nentries] = total_blocks. */
extern "C" int seld_hcq_pack_flat(const void* table_dev, const int32_t* starts_dev, int32_t nentries, int32_t total_blocks,
                                  void* stream) {
    if (!table_dev || !starts_dev || nentries <= 0 || total_blocks <= 0) return SELD_EINVAL;
    hipLaunchKernelGGL(hcq_pack_flat_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const HcqPackP*)table_dev, starts_dev, nentries);
    return check_launch();
}

extern "C" int seld_hcq_conv(const seld_conv_desc* d, int32_t mode, int32_t npair, const float* x, const float* x2,
                             const float* wpack, float* const y[2], const float* const bias[2],
                             const int32_t epilogue[2], const float* const addend[2], float* const stats[2],
                             void* stream) {
    if (hc_validate(d) != SELD_OK || !x || !wpack || !y || !y[0]) return SELD_EINVAL;
    if ((mode != 0 && mode != 1) || npair < 1 || npair > 2) return SELD_EINVAL;
    if (mode == 1 && npair == 2 && !x2) return SELD_EINVAL;
    HcqPlan pl = hcq_plan(d, mode, npair);
    if (!pl.ok) return SELD_EUNSUPPORTED;
    const int nsets = mode == 0 ? npair : 1;
    for (int s = 0; s < 2; ++s) {
        const bool on = s < nsets;
        pl.kp.dst[s] = on ? y[s] : nullptr;
        pl.kp.bias[s] = (on && bias) ? bias[s] : nullptr;
        pl.kp.epilogue[s] = (on && epilogue) ? epilogue[s] : 0;
        pl.kp.addend[s] = (on && addend) ? addend[s] : nullptr;
        pl.kp.stats[s] = (on && stats) ? stats[s] : nullptr;
        if (on && !pl.kp.dst[s]) return SELD_EINVAL;
        if ((pl.kp.epilogue[s] & SELD_EPI_ADD) && !pl.kp.addend[s]) return SELD_EINVAL;
        if ((pl.kp.epilogue[s] & SELD_EPI_STATS) && !pl.kp.stats[s]) return SELD_EINVAL;
    }
    pl.kp.src = x;
    pl.kp.src2 = x2;
    pl.kp.wpack = wpack;
    return hcq_launch(pl, (hipStream_t)stream);
}


/* conv -> [BatchNorm2d -> ReLU ->] MaxPool2d(8, 1) for the network's first layer (model.py:273-281), pooling decision
 * inside the convolution (hcq_first_pool_kernel): writes y (nullable: not written, and then no statistics either -- the
 * seld_first_stage_* path), the BatchNorm statistics (want_stats), and per pooling window
 * the raw value at the row that will be the maximum after BatchNorm + ReLU (by the sign of gamma) and that row.
 * wpack: seld_hcq_pack(desc, mode 2).  Follow with seld_bn_finalize_ex and seld_bn_pool_finish. */
static int hcq_first_pool_impl(const seld_conv_desc* d, const float* x, const float* wpack, const float* bias,
                               int32_t want_stats, float* y, float* stats, const HcqPoolP& pp, bool fin, void* stream) {
    HcqPlan pl = hcq_plan(d, 2, 1);
    HcqKern k;
    if (!pl.ok || !hcq_pick(pl, &k) || pl.first_rows != 8) return SELD_EUNSUPPORTED;
    pl.kp.src = x; pl.kp.src2 = nullptr; pl.kp.wpack = wpack;
    pl.kp.dst[0] = y; pl.kp.bias[0] = bias; pl.kp.epilogue[0] = want_stats ? SELD_EPI_STATS : 0; pl.kp.stats[0] = stats;
    const int NT = pl.NT1 + pl.NT2;
    const int npair = ((pl.IBC * 9 + 3) / 4 + 1) / 2;
    const size_t smem = ((size_t)pl.kp.A * pl.IBC * 10 * 72 + (size_t)NT * 4 * 4 * 16 * 2 +
                         (size_t)pl.NR * npair * 8 * 64 * 2) * sizeof(float);    // rows + statistics + one tile's fragments
    hipStream_t st = (hipStream_t)stream;
#define SELD_FP(IBC_, NT1_, NT2_, NR_)                                                                                  \
    do {                                                                                                                \
        auto kern = fin ? hcq_first_pool_kernel<IBC_, NT1_, NT2_, NR_, 8, false, true>                                  \
                        : (y ? hcq_first_pool_kernel<IBC_, NT1_, NT2_, NR_, 8, true> : hcq_first_pool_kernel<IBC_, NT1_, NT2_, NR_, 8, false>); \
        if (smem > 64 * 1024 &&                                                                                         \
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) \
            return SELD_ELAUNCH;                                                                                        \
        hipLaunchKernelGGL(kern, pl.grid, dim3(256), smem, st, pl.kp, pp);                                              \
        return check_launch();                                                                                          \
    } while (0)
    if (k.IBC == 1) {
        if (k.NR == 2) { if (k.NT2 == 2) SELD_FP(1, 1, 2, 2); else SELD_FP(1, 1, 1, 2); }
        if (k.NT1 == 2) SELD_FP(1, 2, 0, 1); else SELD_FP(1, 1, 0, 1);
    }
    if (k.NR == 2) { if (k.NT2 == 2) SELD_FP(2, 1, 2, 2); else SELD_FP(2, 1, 1, 2); }
    if (k.NT1 == 2) SELD_FP(2, 2, 0, 1); else SELD_FP(2, 1, 0, 1);
#undef SELD_FP
}

extern "C" int seld_hcq_first_pool(const seld_conv_desc* d, const float* x, const float* wpack, const float* bias,
                                   const float* gamma, int32_t want_stats, float* y, float* stats, float* pool_raw,
                                   uint8_t* idx, void* stream) {
    if (hc_validate(d) != SELD_OK || !x || !wpack || !gamma || !pool_raw || !idx || (want_stats && (!stats || !y))) return SELD_EINVAL;
    const HcqPoolP pp{gamma, pool_raw, idx, nullptr, nullptr, nullptr, nullptr, DropP{0.f, 1.f, 0, 0, nullptr}};
    return hcq_first_pool_impl(d, x, wpack, bias, want_stats, y, stats, pp, false, stream);
}

extern "C" int seld_hcq_first_pool_bn(const seld_conv_desc* d, const float* x, const float* wpack, const float* bias,
                                      const float* gamma, const float* beta, const float* mean, const float* invstd,
                                      float drop_p, uint64_t seed, uint64_t offset, const uint64_t* state, float* pool_raw,
                                      uint8_t* idx, float* out, void* stream) {
    if (hc_validate(d) != SELD_OK || !x || !wpack || !gamma || !beta || !mean || !invstd || !pool_raw || !idx || !out ||
        drop_p < 0.f || drop_p >= 1.f)
        return SELD_EINVAL;
    const HcqPoolP pp{gamma, pool_raw, idx, beta, mean, invstd, out, DropP{drop_p, 1.0f / (1.0f - drop_p), seed, offset, state}};
    return hcq_first_pool_impl(d, x, wpack, bias, 0, nullptr, nullptr, pp, true, stream);
}
