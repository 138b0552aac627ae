// Hypercomplex convolution, forward and data gradient: the loop-invariant-staging variant (gfx950).
//
// Same implicit GEMM as hc_conv_fwd.hip (D[pos][ch] = sum_kk X(pos,kk) * Wfull(ch,kk) on v_mfma_f32_16x16x4_f32,
// Hamilton component / sign applied while the component tensors are staged, dual-quaternion zero quadrant skipped),
// but built around what the PMC counters of that kernel showed: at 46 % MFMA busy it issued ~6.7 other
// instructions per MFMA -- the fp32 MFMA is only 32 cycles long, so every address computation in the K loop
// competes with it.  Here the K chunk is a whole number of input channels (KC = 36 for 3x3 taps, 24 for 1x3 and
// 1x1), which makes EVERYTHING a thread needs for its staging loads loop-invariant:
//
//   * X (im2col) item = (k in chunk, 4 consecutive output positions).  k fixes the tap, so the halo masks and the
//     offset inside an image never change; per chunk an item costs one v_add (advance by KC/KK channels), one
//     16-byte buffer load (positions are contiguous along W because the W stride is 1) and four v_and that zero
//     the halo elements on the way to LDS.  The old kernel spent ~30 instructions on four 4-byte gathers.
//   * W item = (row, 16-byte piece of the row's KC floats).  Pointer and Hamilton sign change only when the chunk
//     enters the next component block (a wave-uniform branch every CK/KC chunks); per chunk: one 64-bit add, one
//     global_load_dwordx4, two v_pk_mul, one ds_write_b128.
//   * 108 (KC = 36) or 72 MFMAs per wave between barriers instead of 48.
//
// LDS images: X as [k][pos] (+4 floats of padding per row), W as [k/4][row][4].  MFMA step s of a 16-k super group
// uses k = 16r + 4*(lane/16) + s, so one ds_read_b128 of W still feeds four MFMAs; the 4 or 8 k's left over after
// the super groups are read with b32 / b64.
//
// A 16-byte load whose first or last elements lie in the halo touches the neighbouring row.  That is harmless
// (masked) except at the very start / end of the tensor, where part of the access would fall outside the buffer
// descriptor; workgroups that can reach those rows ("edge" workgroups, wave-uniform) gather per element instead.
//
// Eligibility is checked by hc_conv_vec_try(); everything else runs on hc_conv_kernel.
#include <type_traits>
#include "hc_common.h"

namespace seld {

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

// n / d for 0 <= n < 2^20, d >= 1 with a precomputed 1.0f / d: exact (the +0.5 keeps the float quotient at least
// 0.5/d away from an integer, far more than the 2^-23 relative rounding error); ~5 instructions instead of ~30.
__device__ __forceinline__ int small_div(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }

template <int CT, int PT, int KH_T, int KW_T, int MODE, int KC, int PAIRS>
__global__ __launch_bounds__(256) void hc_conv_vec_kernel(const ConvP p) {
    constexpr int KK = KH_T * KW_T;                // KC: K chunk, whole channels, multiple of 4 (36: 3x3 and 1x3; 24: 1x3 and 1x1)
    constexpr int NG = KC / 4;                     // k-groups of 4 (16-byte pieces of a weight row)
    constexpr int NS = NG / 4;                     // super groups of 16 k
    constexpr int NL = NG % 4;                     // left-over k-groups: 1 (b32 reads) or 2 (b64 reads)
    constexpr int CADV = KC / KK;                  // source channels per chunk
    constexpr int BC = CT * 16;
    constexpr int BP = PT * 64;
    constexpr int QP = BP / 4;                     // position quads per k row
    constexpr int XROW = BP + 4;                   // padded row: the 4 k rows of one A read land on different banks
    constexpr int KSTEP = 256 / QP;                // k distance between the items of one thread
    constexpr int KWAVE = 64 / QP;                 // k rows one wave covers per item
    constexpr int XI = (KC + KSTEP - 1) / KSTEP;
    constexpr int WTOT = BC * NG;
    constexpr int WI = (WTOT + 255) / 256;
    // Row pitch of the W image in 16-byte pieces: the k-pieces of a row are contiguous (conflict-free 16-byte stores of
    // consecutive pieces); 6 and 9 both leave the ds_read_b128 fragment reads where they are (SQ_LDS_BANK_CONFLICT is
    // the same for pitches 9 and 10, and for 7 and 6 -- the remaining conflicts are the 4-byte left-over reads).
    constexpr int RS = NG;
    static_assert(NL == 1 || NL == 2, "KC is 36 or 24");
    static_assert(KC % KK == 0 && KC % KWAVE == 0, "chunk = whole channels");

    __shared__ __attribute__((aligned(16))) float Xs[2][KC][XROW];
    __shared__ __attribute__((aligned(16))) float Ws[2][BC][RS][4];
    __shared__ unsigned wdelta_s[16];
    // statistics scratch of a forward pair: the staging buffers hold slot 1's first chunk while slot 0 is written out
    __shared__ float pair_red_s[(PAIRS && MODE == MODE_FWD) ? 4 * CT * 16 * 2 : 1];              // [slot][component] byte offsets from p.wmin

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = blockIdx.y * BC;
    const long long p0 = (long long)blockIdx.x * BP;
    const int A = p.algebra;
    const int CK = (MODE == MODE_FWD ? p.IA : p.OA) * KK;   // K extent of one component block (multiple of KC)

    if (tid < 16) {
        const float* wp_ = tid < 8 ? p.w.p[tid] : p.w2.p[tid - 8];
        wdelta_s[tid] = wp_ ? (unsigned)((const char*)wp_ - (const char*)p.wmin) : 0u;
    }
    // PAIRS: two convolutions in one launch -- the K loop runs over slot 0's source and weights, then slot 1's.
    //   data gradient: dst = dgrad(src, w) + dgrad(src2, w2), one result;
    //   forward (1x1 layers, same src): slot 0's result is written after the first pass (its stores drain under the
    //   second pass), slot 1's at the end.  On the 1x1 layers a launch's fixed cost is a third of its time
    //   (52 us for the pair against 2 x 33); on the 1x3 layers it did not pay (141 against 2 x 68).
    // A separate instantiation: the extra state costs the single-convolution kernels 10 %.
    static_assert(!PAIRS || MODE == MODE_DGRAD || (KH_T == 1 && KW_T == 1), "forward pairs: 1x1 layers only");
    constexpr bool FWD_PAIR = PAIRS && MODE == MODE_FWD;
    const int zslot = 0;
    const int kslots = PAIRS ? 2 : 1;
    int wslot = zslot;                             // weight set the next chunk to load belongs to

    // ---- buffer descriptor of the streamed operand: based one image before the tile's first image so that the
    // halo of the tile's first row reads (and masks) the end of the previous image instead of wrapping -----------
    const long long img0 = p0 / p.dstS;
    const long long imgb = img0 > 0 ? img0 - 1 : 0;
    const long long img_elems = (long long)p.Csrc * p.srcS;
    const float* sbase = p.src + imgb * img_elems;
    const long long remain = (p.src_elems - imgb * img_elems) * 4;
    const unsigned OOB = 0xFFFFFFF0u;
    const unsigned nrec = remain > (long long)OOB ? OOB : (unsigned)remain;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, nrec, 0x00020000);

    const int rem0 = (int)(p0 - img0 * p.dstS);
    auto decode = [&](int local, int* dimg, int* rem) __attribute__((always_inline)) {
        int r = rem0 + local;
        int di = 0;
        if (p.dstS >= BP) {
            if (r >= p.dstS) { r -= p.dstS; di = 1; }
        } else {
            di = r / p.dstS;
            r -= di * p.dstS;
        }
        *dimg = di;
        *rem = r;
    };

    // Workgroups that can touch the first row of the tensor with a left halo, or the last row with a right halo.
    const int hs = max(abs(p.OFFh), (KH_T - 1) * abs(p.KDh)) + 1;
    const long long edge_span = (long long)hs * p.dstW;
    const bool edge_wg = (p0 < edge_span) || (p0 + BP + edge_span > p.Ptot);

    // ---- K range of this workgroup (dual-quaternion zero quadrant), in whole chunks -------------------------
    const int half_c = p.Cdst >> 1;
    const int half_k = p.Ktot >> 1;
    const bool halves_aligned = (p.skip_mode != 0) && (half_k % KC == 0) && (half_c % 16 == 0);
    // (Pairing primal and dual tiles in one workgroup for balance was measured 0-7 % SLOWER on the TCN layers: the
    // half-K chunks then carry half the MFMAs for the same staging and barrier.  Tiles stay contiguous.)
    // (Balanced tiles -- every workgroup half primal, half dual channels, so that all run 12 full + 12 half chunks
    // instead of 12 or 24 full ones -- were measured in round 2: 1x3 forward 67.4 vs 66.7 us, pair data gradient 77.6 vs
    // 69.4: a chunk costs the same with 36 MFMAs as with 72, the loop is bound by the staging round trip, not by the
    // matrix pipe.  Tiles stay contiguous.)
    auto chan_of = [&](int t) __attribute__((always_inline)) { return c0 + t; };
    int kbeg = 0, kend = p.Ktot;
    if (halves_aligned) {
        if (p.skip_mode == 1 && c0 + BC <= half_c) kend = half_k;      // all channels primal
        if (p.skip_mode == 2 && c0 >= half_c) kbeg = half_k;           // all channels dual
    }
    // p.pairing carries SELD_VEC_DBG (timing experiments, wrong results): 4 = one chunk only, 8 = no epilogue stores,
    // 16 = return after the setup, 32 = return after staging the first chunk.  On the 1x3 TCN layer (74 us): launch +
    // setup 2.6 us, first load round trip 2.7, first iteration 4.7 (cold), output write-back 4.5, 23 more iterations 2.4 each
    const int nchunks = (p.pairing & 4) ? 1 : (kend - kbeg) / KC;
    const bool mixed_wg = !(p.pairing & 4) && halves_aligned && (CT % 2 == 0) && (c0 + BC / 2 == half_c);

    // ---- X items: everything but the channel advance is loop-invariant ----------------------------------------
    const int quad = tid % QP;
    const int kx0 = tid / QP;
    unsigned xoff[XI], xmask[XI];
    int xlds[XI];
    const unsigned xadv = (unsigned)(CADV * p.srcS * 4);   // invalid items drift too: they are masked, and a
                                                           // range-checked load cannot fault
    {
        const long long pg = p0 + 4 * quad;
        const bool pvalid = pg < p.Ptot;               // Ptot is a multiple of 4: the quad is all-in or all-out
        int dimg = 0, rem = 0;
        if (pvalid) decode(4 * quad, &dimg, &rem);
        const int oh = (p.dstS < (1 << 20)) ? small_div(rem, 1.0f / (float)p.dstW) : rem / p.dstW;
        const int ow = rem - oh * p.dstW;              // multiple of 4, the quad stays inside the row
        const int base_h = oh * p.SMh + p.OFFh;
        const int base_w = ow + p.OFFw;                // W stride is 1
        const unsigned img_b = (unsigned)(img0 - imgb + dimg) * (unsigned)img_elems * 4u;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            // A thread whose last item falls outside the chunk (wave-uniform) repeats its item 0 instead: the same
            // load and the same LDS store twice is harmless and keeps the K loop free of branches.
            const int k = (kx0 + j * KSTEP < KC) ? kx0 + j * KSTEP : kx0;
            const int cl = k / KK;
            const int tap = k - cl * KK;
            const int kh = tap / KW_T;
            const int kw = tap - kh * KW_T;
            const int ih = base_h + kh * p.KDh;
            const int iw0 = base_w + kw * p.KDw;
            unsigned m = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) m |= ((unsigned)(iw0 + e) < (unsigned)p.srcW) ? (1u << e) : 0u;
            if (!pvalid || (unsigned)ih >= (unsigned)p.srcH) m = 0;
            const unsigned off = img_b + (unsigned)(((kbeg / KK + cl) * p.srcS + ih * p.srcW + iw0) * 4);
            xmask[j] = m;
            xoff[j] = m ? off : OOB;
            xlds[j] = k * XROW + 4 * quad;
        }
    }

    // ---- W items: (row, 16-byte piece g) = (f / NG, f % NG) for f = tid + 256*i; the LDS image is [row][g] with an
    // odd row pitch (conflict-free b128 fragment reads), so for NG odd the write address is just f ------------
    // The component tensors are read through one range-checked descriptor [wmin, wmin + wspan): the prefetch of the
    // chunk after the last one needs no guard.
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wmin, 0, p.wspan, 0x00020000);
    int kq = kbeg / CK;                  // component block the next chunk to load lies in
    int kl = kbeg - kq * CK;             // its offset inside the block (multiple of KC)
    int lc = 0;                          // chunks of the current slot already requested
    unsigned woff[WI];                   // byte offset from wmin of this item's piece in the chunk to load next
    float wmul[WI];                      // Hamilton sign, or 0 (structural zero / row outside the tensor)
    unsigned wrow[WI];                   // loop-invariant: byte offset of the piece inside a component tensor (a
                                         // multiple of 16) | 8 if the row exists | the row's component index a
    int wlds[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int f = (tid + 256 * i < WTOT) ? tid + 256 * i : tid;     // a missing last item repeats item 0
        const int row = f / NG;
        const int g = f - row * NG;
        wlds[i] = (row * RS + g) * 4;
        const int chg = chan_of(row);
        const bool ok = chg < p.Cdst;
        const int cc = ok ? chg : 0;
        const int per = (MODE == MODE_FWD) ? p.OA : p.IA;
        const int a = small_div(cc, 1.0f / (float)per);
        wrow[i] = (unsigned)(((cc - a * per) * CK + 4 * g) * 4) | (ok ? 8u : 0u) | (unsigned)a;
    }
    // A new component block: ~14 instructions per item (this runs every CK/KC chunks, i.e. every 3rd..6th).
    auto setup_comp = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            unsigned v = wrow[i];
            asm volatile("" : "+v"(v));      // nothing derived from v may be hoisted into the K loop's live registers
            const int a = v & 7;
            bool zero, neg;
            const int comp = (MODE == MODE_FWD) ? hc_comp(A, a, kq, &zero, &neg) : hc_comp(A, kq, a, &zero, &neg);
            woff[i] = wdelta_s[wslot * 8 + (comp & 7)] + (v & ~15u) + (unsigned)(kl * 4);
            wmul[i] = ((v & 8u) && !zero) ? (neg ? -1.f : 1.f) : 0.f;
        }
    };

    floatx4 xr[XI];
    floatx4 wr[WI];

    // The component block changes BETWEEN two chunks: the multipliers of the chunk that is still in registers are
    // needed until its store_chunk, so the switch is carried out at the start of the following load.
    bool comp_switch = true;
    auto load_chunk = [&](auto edgec) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edgec)::value;
        if (PAIRS && lc == nchunks && wslot == 0) {
            // data gradient of a pair: the first slot's K range is exhausted, go on with the second source / weights
            wslot = 1;
            lc = 0;
            rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src2 + imgb * img_elems), 0, nrec, 0x00020000);
#pragma unroll
            for (int j = 0; j < XI; ++j) xoff[j] -= (unsigned)nchunks * xadv;
            kq = kbeg / CK;
            kl = kbeg - kq * CK;
            setup_comp();
        }
        ++lc;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            if (!EDGE) {
                const uintx4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, xoff[j], 0, 0);
                xr[j][0] = __uint_as_float(v[0]); xr[j][1] = __uint_as_float(v[1]);
                xr[j][2] = __uint_as_float(v[2]); xr[j][3] = __uint_as_float(v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned o = ((xmask[j] >> e) & 1u) ? xoff[j] + 4u * e : OOB;
                    xr[j][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, o, 0, 0));
                }
            }
            xoff[j] += xadv;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const uintx4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff[i], 0, 0);
            wr[i][0] = __uint_as_float(v[0]); wr[i][1] = __uint_as_float(v[1]);
            wr[i][2] = __uint_as_float(v[2]); wr[i][3] = __uint_as_float(v[3]);
            woff[i] += KC * 4;
        }
        kl += KC;
        comp_switch = kl >= CK;
        if (comp_switch) {
            kl = 0;
            ++kq;
        }
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
        float* xs = &Xs[buf][0][0];
        float* ws = &Ws[buf][0][0][0];
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            float4 o;
            o.x = __uint_as_float(__float_as_uint(xr[j][0]) & (unsigned)__builtin_amdgcn_sbfe(xmask[j], 0, 1));
            o.y = __uint_as_float(__float_as_uint(xr[j][1]) & (unsigned)__builtin_amdgcn_sbfe(xmask[j], 1, 1));
            o.z = __uint_as_float(__float_as_uint(xr[j][2]) & (unsigned)__builtin_amdgcn_sbfe(xmask[j], 2, 1));
            o.w = __uint_as_float(__float_as_uint(xr[j][3]) & (unsigned)__builtin_amdgcn_sbfe(xmask[j], 3, 1));
            *reinterpret_cast<float4*>(xs + xlds[j]) = o;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i)
            *reinterpret_cast<float4*>(ws + wlds[i]) =
                make_float4(wr[i][0] * wmul[i], wr[i][1] * wmul[i], wr[i][2] * wmul[i], wr[i][3] * wmul[i]);
    };

    floatx4 acc[PT][CT];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15;      // row / col inside a 16x16 tile
    const int fk = lane >> 4;      // k group of this lane

    __syncthreads();               // wdelta_s visible
    if (p.pairing & 16) return;    // SELD_VEC_DBG timing experiment: setup only
    using ETrue = std::integral_constant<bool, true>;
    using EFalse = std::integral_constant<bool, false>;
    if (nchunks > 0) {
        setup_comp();
        load_chunk(ETrue{});           // per-element gather: right for every workgroup
        store_chunk(0);
    }
    __syncthreads();
    if (p.pairing & 32) return;    // SELD_VEC_DBG timing experiment: setup + first chunk staged

    // One chunk: [component switch, rare] then ONE basic block -- prefetch of the next chunk into registers, the
    // MFMAs of this one, the registers to the other LDS buffer, barrier.  The prefetch after the last chunk reads
    // through range-checked descriptors and lands in an LDS buffer nobody reads.
    int gchunk = 0;                    // chunks computed so far (both slots): LDS buffer parity
    auto run_chunks = [&](int cbeg, int cend, auto j0c, auto j1c, auto edgec) __attribute__((always_inline)) {
        constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
        for (int chunk = cbeg; chunk < cend; ++chunk) {
            const int buf = gchunk & 1;
            ++gchunk;
            if (comp_switch) setup_comp();
            load_chunk(edgec);
            // The MFMA stream is software-pipelined by hand, one scheduling region per stage (sched_barrier): the
            // fragments of stage r+1 are read from LDS while the MFMAs of stage r run.  Left to itself the scheduler
            // reads both super groups up front (a long wait before the first MFMA), has no registers left to
            // prefetch the left-over group (one exposed LDS round trip per two MFMAs at the end), sinks the global
            // loads into the middle of the stream and hoists the LDS stores -- measured 10-15 %.
            const float* xb = &Xs[buf][0][wave * (PT * 16) + fr];
            float av[2][PT][4];
            floatx4 bv[2][CT];
            float lav[PT][2];            // left-over stage: NL == 1 uses [0], NL == 2 both
            float2 lbv[CT];
            auto read_super = [&](int r, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < PT; ++i) av[slot][i][s] = xb[(16 * r + 4 * fk + s) * XROW + i * 16];
#pragma unroll
                for (int j = J0; j < J1; ++j) bv[slot][j] = *reinterpret_cast<const floatx4*>(&Ws[buf][j * 16 + fr][4 * r + fk][0]);
            };
            auto read_left = [&]() __attribute__((always_inline)) {
                if (NL == 1) {
#pragma unroll
                    for (int i = 0; i < PT; ++i) lav[i][0] = xb[(16 * NS + fk) * XROW + i * 16];
#pragma unroll
                    for (int j = J0; j < J1; ++j) lbv[j].x = Ws[buf][j * 16 + fr][4 * NS][fk];
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int i = 0; i < PT; ++i) lav[i][t] = xb[(16 * NS + 2 * fk + t) * XROW + i * 16];
#pragma unroll
                    for (int j = J0; j < J1; ++j)
                        lbv[j] = *reinterpret_cast<const float2*>(&Ws[buf][j * 16 + fr][4 * NS + (fk >> 1)][(fk & 1) * 2]);
                }
            };
            read_super(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < NS; ++r) {
                if (r + 1 < NS) read_super(r + 1, (r + 1) & 1);
                else read_left();
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = J0; j < J1; ++j)
#pragma unroll
                        for (int i = 0; i < PT; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r & 1][i][s], bv[r & 1][j][s], acc[i][j], 0, 0, 0);
                // inside the region: one LDS read of the next stage after every second MFMA, from the start
#pragma unroll
                for (int q = 0; q < (J1 - J0) + 4 * PT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // last region: the left-over MFMAs mix with the register -> LDS stores of the next chunk
#pragma unroll
            for (int j = J0; j < J1; ++j)
#pragma unroll
                for (int i = 0; i < PT; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(lav[i][0], lbv[j].x, acc[i][j], 0, 0, 0);
            if (NL == 2) {
#pragma unroll
                for (int j = J0; j < J1; ++j)
#pragma unroll
                    for (int i = 0; i < PT; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(lav[i][1], lbv[j].y, acc[i][j], 0, 0, 0);
            }
            store_chunk(buf ^ 1);
            __syncthreads();
        }
    };
    using IC0 = std::integral_constant<int, 0>;
    using ICH = std::integral_constant<int, CT / 2>;
    using ICT = std::integral_constant<int, CT>;
    // ---- epilogue: lane holds 4 consecutive positions (regs) of channel c0 + j*16 + fr ----------------------
    auto epilogue = [&](int zs) __attribute__((always_inline)) {
        float* const dstz = zs ? p.dst2 : p.dst;
        const float* const addz = zs ? p.addend2 : p.addend;
        const float* const biasz = zs ? p.bias2 : p.bias;
        float* const statsz = zs ? p.stats2 : p.stats;
        const int epi = zs ? p.epilogue2 : p.epilogue;
        float* const dst0 = dstz + (size_t)img0 * p.Cdst * p.dstS;
        const float* const add0 = addz ? addz + (size_t)img0 * p.Cdst * p.dstS : nullptr;
        int poff[PT];
    #pragma unroll
        for (int i = 0; i < PT; ++i) {
            int dimg, rem;
            decode(wave * (PT * 16) + i * 16 + fk * 4, &dimg, &rem);
            poff[i] = dimg * p.Cdst * p.dstS + rem;
        }
    #pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int ch = chan_of(j * 16 + fr);
            const bool chok = ch < p.Cdst;
            const float bvv = (chok && biasz) ? biasz[ch] : 0.0f;
            float s1 = 0.f, s2 = 0.f;
    #pragma unroll
            for (int i = 0; i < PT; ++i) {
                const long long pos = p0 + wave * (PT * 16) + i * 16 + fk * 4;
                const floatx4 v = acc[i][j];
                if (chok && pos < p.Ptot && !(p.pairing & 8)) {
                    const size_t off = (size_t)(poff[i] + ch * p.dstS);
                    float4 o = make_float4(v[0] + bvv, v[1] + bvv, v[2] + bvv, v[3] + bvv);
                    if (epi & SELD_EPI_ADD) {
                        const float4 ad = *reinterpret_cast<const float4*>(add0 + off);
                        o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                    }
                    if (epi & SELD_EPI_ACCUMULATE) {
                        const float4 old = *reinterpret_cast<const float4*>(dst0 + off);
                        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                    }
                    *reinterpret_cast<float4*>(dst0 + off) = o;
                    s1 += o.x + o.y + o.z + o.w;
                    s2 += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
                }
            }
            if (epi & SELD_EPI_STATS) {
                s1 += __shfl_xor(s1, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 16, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (fk == 0) {
                    float* redbuf = FWD_PAIR ? pair_red_s : &Xs[0][0][0];   // the K loop is over: the staging buffers are free
                    redbuf[(wave * BC + j * 16 + fr) * 2 + 0] = s1;
                    redbuf[(wave * BC + j * 16 + fr) * 2 + 1] = s2;
                }
            }
        }
        if (epi & SELD_EPI_STATS) {
            static_assert(sizeof(Xs) >= 4 * BC * 2 * sizeof(float), "statistics scratch fits the X staging buffers");
            __syncthreads();
            const float* redbuf = &Xs[0][0][0];
            float* rep = statsz + (size_t)(blockIdx.x % SELD_STATS_REPLICAS) * 2 * p.Cdst;
            for (int t = tid; t < BC; t += 256) {
                const int ch = chan_of(t);
                if (ch < p.Cdst) {
                    float a1 = 0.f, a2 = 0.f;
    #pragma unroll
                    for (int wv = 0; wv < 4; ++wv) {
                        a1 += redbuf[(wv * BC + t) * 2 + 0];
                        a2 += redbuf[(wv * BC + t) * 2 + 1];
                    }
                    atomicAdd(rep + ch, a1);
                    atomicAdd(rep + p.Cdst + ch, a2);
                }
            }
        }
    };

    // (data gradient of a pair: the chunk ranges are simply run again for the second source; load_chunk's state
    // machine switches descriptor and weights by itself.  Written as straight-line repeats, not as a loop over the
    // slots: the nested loop cost the 12-tile kernels 100+ VGPRs.)
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
    };
    auto run_all = [&](auto edgec) __attribute__((always_inline)) {
        if (mixed_wg) {
            const int csplit = (half_k - kbeg) / KC;          // first chunk of the upper K half
            if (p.skip_mode == 1) {                           // forward: primal tiles (lower half) see zeros there
                run_chunks(0, csplit, IC0{}, ICT{}, edgec);
                run_chunks(csplit, nchunks, ICH{}, ICT{}, edgec);
                if (FWD_PAIR) {
                    epilogue(0);
                    zero_acc();
                    run_chunks(0, csplit, IC0{}, ICT{}, edgec);
                    run_chunks(csplit, nchunks, ICH{}, ICT{}, edgec);
                }
            } else {                                          // dgrad: dual tiles (upper half) see zeros in the lower K half
                run_chunks(0, csplit, IC0{}, ICH{}, edgec);
                run_chunks(csplit, nchunks, IC0{}, ICT{}, edgec);
                if (PAIRS) {
                    run_chunks(0, csplit, IC0{}, ICH{}, edgec);
                    run_chunks(csplit, nchunks, IC0{}, ICT{}, edgec);
                }
            }
        } else if (FWD_PAIR) {
            run_chunks(0, nchunks, IC0{}, ICT{}, edgec);
            epilogue(0);
            zero_acc();
            run_chunks(0, nchunks, IC0{}, ICT{}, edgec);
        } else {
            run_chunks(0, kslots * nchunks, IC0{}, ICT{}, edgec);
        }
    };
    if (edge_wg) run_all(ETrue{});
    else run_all(EFalse{});
    epilogue(FWD_PAIR ? 1 : 0);

}

// Can this problem run on the vector-staging kernel with tile (ct, pt)?  Returns the K chunk (36 / 24) or 0.
int hc_conv_vec_chunk(const ConvP& p, int mode, int ct, int pt) {
    if (env().conv_novec) return 0;
    if (!(ct == 12 || ct == 6) || pt != 1) return 0;
    if (p.nslots > 1 && p.KH != 1) return 0;                            // pairs: 1-D layers
    if (p.nslots > 1 && mode == MODE_FWD && p.KW != 1) return 0;       // forward pairs: 1x1 layers
    if (!(mode == MODE_FWD || p.wt)) return 0;
    if (p.SDh != 1 || p.SDw != 1 || p.SMh != 1 || p.SMw != 1) return 0;
    if (p.dstW % 4 != 0) return 0;
    const bool t11 = p.KH == 1 && p.KW == 1, t13 = p.KH == 1 && p.KW == 3, t33 = p.KH == 3 && p.KW == 3;
    if (!(t11 || t13 || t33)) return 0;
    const int ck = (mode == MODE_FWD ? p.IA : p.OA) * p.KH * p.KW;
    if (t33) return ck % 36 == 0 ? 36 : 0;          // (36-deep chunks on the 1x3 layers: measured 15 % slower forward)
    return ck % 24 == 0 ? 24 : 0;
    return 0;
}

template <int CT, int MODE>
static void launch_vec(const ConvP& p, hipStream_t st) {
    constexpr int BC = CT * 16, BP = 64;
    dim3 grid((unsigned)((p.Ptot + BP - 1) / BP), (unsigned)((p.Cdst + BC - 1) / BC), 1);
    constexpr int DG = (MODE == MODE_DGRAD) ? 1 : 0;
    if (p.KH == 3) hipLaunchKernelGGL((hc_conv_vec_kernel<CT, 1, 3, 3, MODE, 36, 0>), grid, dim3(256), 0, st, p);
    else if (DG && p.nslots > 1 && p.KW == 3) hipLaunchKernelGGL((hc_conv_vec_kernel<CT, 1, 1, 3, MODE, 24, DG>), grid, dim3(256), 0, st, p);
    else if (p.nslots > 1) hipLaunchKernelGGL((hc_conv_vec_kernel<CT, 1, 1, 1, MODE, 24, 1>), grid, dim3(256), 0, st, p);
    else if (p.KW == 3) hipLaunchKernelGGL((hc_conv_vec_kernel<CT, 1, 1, 3, MODE, 24, 0>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((hc_conv_vec_kernel<CT, 1, 1, 1, MODE, 24, 0>), grid, dim3(256), 0, st, p);
}

// Launches when eligible (returns 1), otherwise returns 0 and the caller falls back to hc_conv_kernel.
int hc_conv_vec_try(const ConvP& p_in, int mode, int ct, int pt, hipStream_t st) {
    const int kc = hc_conv_vec_chunk(p_in, mode, ct, pt);
    if (!kc) return 0;
    ConvP p = p_in;
    p.pairing = env().vec_dbg;          // non-zero only in -DSELD_TUNING builds (timing experiments, wrong results)
    // the component tensors are addressed as wmin + 32-bit byte offset: they must lie within 4 GB of each other
    const size_t comp_bytes = (size_t)p.OA * p.IA * p.KH * p.KW * sizeof(float);
    uintptr_t lo = UINTPTR_MAX, hi = 0;
    if (p.nslots < 1) p.nslots = 1;
    for (int sl = 0; sl < p.nslots; ++sl)
        for (int i = 0; i < p.algebra; ++i) {
            const uintptr_t a = (uintptr_t)(sl ? p.w2.p[i] : p.w.p[i]);
            lo = a < lo ? a : lo;
            hi = a > hi ? a : hi;
        }
    if (hi - lo + comp_bytes >= 0xFFFFFFF0ull) return 0;
    p.wmin = (const float*)lo;
    p.wspan = (unsigned)(hi - lo + comp_bytes);

    if (mode == MODE_FWD) {
        if (ct == 12) launch_vec<12, MODE_FWD>(p, st);
        else launch_vec<6, MODE_FWD>(p, st);
    } else {
        if (ct == 12) launch_vec<12, MODE_DGRAD>(p, st);
        else launch_vec<6, MODE_DGRAD>(p, st);
    }
    return 1;
}

}  // namespace seld
