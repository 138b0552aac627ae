// Hypercomplex (real / quaternion / dual-quaternion) convolution, forward and data gradient, gfx950.
//
// ONE implicit-GEMM kernel template on the fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32):
//
//     D[pos][ch] = sum_kk  X(pos, kk) * Wfull(ch, kk)
//
//   forward : ch = co, kk = ci*KK + kidx, X = im2col(x)            (quaternion_ops.py:147)
//   dgrad   : ch = ci, kk = co*KK + kidx, X = im2col of dy with the mirrored index map
//
// Wfull is the real block matrix of the Hamilton product (quaternion_ops.py:131-135,
// dual_quaternion_ops.py:122-140).  It is never materialised: the weight stager reads the 1/4/8
// COMPONENT tensors and applies comp(p,q) / sign(p,q) on the way into LDS, and the MFMA loop skips the
// structurally-zero quadrant of the dual-quaternion matrix (25 % of the flops).
//
// Tiling (wave64): a workgroup of 4 waves owns BC = 16*CT channels x BP = 64*PT positions; the waves
// split the positions.  Positions sit on the MFMA row index so that each lane ends up with 4 consecutive
// positions of one channel -> 16-byte coalesced stores along T / W.  LDS images are [k/4][row][4] so one
// ds_read_b128 feeds four MFMAs (bank-conflict free: measured SQ_LDS_BANK_CONFLICT = 0); staging is
// register double-buffered with one barrier per 16-deep K step.
//
// The f32 MFMA runs at the VALU rate, so staging instructions are what such a kernel must economise:
//   * everything that depends only on the K index is wave-uniform (SALU): wave w stages k-group w of the
//     weight rows, the (q, local-k) / (co, kidx) split of the K index is tracked incrementally in SGPRs;
//   * the im2col gather uses raw buffer loads: 4 VALU per element (tap offset add, unsigned range
//     compare, 3-input address add, select of an out-of-range offset that the hardware zero-fills);
//   * per-row decodes are hoisted out of the K loop.
// (CT, PT) is picked per problem so that small layers (B*T = 16k positions) still fill 256 CUs.
#include <type_traits>
#include "hc_common.h"

namespace seld {

// FAST = 1 fixes at compile time what the launcher has checked: weight rows contiguous along K (forward, or data
// gradient with transposed weights), component extent CK a multiple of 4 and >= 16, no strided data gradient.
// The staging code then has NO control flow (hipcc's waitcnt insertion is conservative at branch merges: with
// the run-time branches it waited for the im2col loads before issuing the weight loads).
template <int CT, int PT, int KH_T, int KW_T, int MODE, int FAST>
__global__ __launch_bounds__(256) void hc_conv_kernel(const ConvP p) {
    constexpr int BC = CT * 16;
    constexpr int BP = PT * 64;
    constexpr int XG = PT;                         // float4 k-groups of the X tile staged per thread
    constexpr int XSTEP = 256 / BP;                // distance between the k-groups one thread stages
    constexpr int WR = (BC + 63) / 64;             // weight rows staged per lane
    static_assert(PT == 1 || PT == 2 || PT == 4, "BP must divide 256");

    __shared__ __attribute__((aligned(16))) float Xs[2][4][BP][4];
    __shared__ __attribute__((aligned(16))) float Ws[2][4][BC][4];
    __shared__ const float* wptr_s[8];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xpos = tid & (BP - 1);
    const int xg0 = __builtin_amdgcn_readfirstlane(tid / BP);
    const int c0 = blockIdx.y * BC;
    const long long p0 = (long long)blockIdx.x * BP;

    const int KH = KH_T ? KH_T : p.KH;
    const int KW = KW_T ? KW_T : p.KW;
    const int KK = KH * KW;
    const int A = p.algebra;
    // `lin`: the weight rows are contiguous along the K axis inside one component block (forward always; data
    // gradient when the caller supplied transposed component tensors Wt[c][o][k], p.wt)
    const bool lin = FAST ? true : ((MODE == MODE_FWD) || (p.wt != 0));
    const int CK = (MODE == MODE_FWD ? p.IA : p.OA) * KK;   // K extent of one component block

    if (tid < 8) wptr_s[tid] = p.w.p[tid];

    // ---- buffer descriptor of the streamed operand, based at the first image this tile touches -----
    const long long img0 = p0 / p.dstS;
    const long long img_elems = (long long)p.Csrc * p.srcS;
    const float* sbase = p.src + img0 * img_elems;
    const long long remain = (p.src_elems - img0 * img_elems) * 4;
    const unsigned nrec = remain > 0xFFFFFFFFLL ? 0xFFFFFFFFu : (unsigned)remain;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, nrec, 0x00020000);

    // Position -> (image relative to img0, offset in image) without per-lane 64-bit division: the tile
    // starts at (img0, rem0) (scalar), a lane adds its local index and wraps (32-bit divide only when an
    // image is shorter than the tile).
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

    // ---- per-thread position decode for the X stage (one position per thread) ----------------------
    const long long pg = p0 + xpos;
    const bool pvalid = pg < p.Ptot;
    int base_h = 0, base_w = 0;
    int img_b = 0;                                 // byte offset of this position's image from sbase
    if (pvalid) {
        int dimg, rem;
        decode(xpos, &dimg, &rem);
        const int oh = rem / p.dstW;
        const int ow = rem - oh * p.dstW;
        base_h = oh * p.SMh + p.OFFh;
        base_w = ow * p.SMw + p.OFFw;
        img_b = dimg * (int)img_elems * 4;
    }
    const bool strided = FAST ? false : ((p.SDh > 1) || (p.SDw > 1));
    const int srcWb = p.srcW * 4;
    const int base_wb = base_w * 4;
    // row term for tap row 0 (all of it when KH == 1)
    const bool hok0 = pvalid && ((unsigned)base_h < (unsigned)p.srcH);
    const int rowb0 = img_b + base_h * srcWb;

    // ---- channel tile of this workgroup -----------------------------------------------------------
    // Dual quaternion: the first half of the channels (primal part) sees a structurally-zero K half.  So
    // that every workgroup carries the same work, a workgroup MAY take BC/2 primal channels and the BC/2 dual
    // channels at the same offset in the upper half (local tile j < CT/2 primal, j >= CT/2 dual) whenever
    // the shape allows; otherwise tiles are contiguous.
    const int half_c = p.Cdst >> 1;
    const int half_k = p.Ktot >> 1;
    const bool halves_aligned = (p.skip_mode != 0) && (half_k % 16 == 0) && (half_c % 16 == 0);
    const bool paired = p.pairing && halves_aligned && (CT % 2 == 0) && (half_c % (BC / 2) == 0);
    auto chan_of = [&](int t) __attribute__((always_inline)) {
        if (!paired) return c0 + t;
        const int base = blockIdx.y * (BC / 2);
        return t < BC / 2 ? base + t : half_c + base + (t - BC / 2);
    };

    // ---- per-lane decode of the weight rows this lane stages ---------------------------------------
    int w_a[WR], w_off[WR];        // fwd: a = output component p, off = o*CK ; dgrad: a = input component q, off = c*KK
    bool w_ok[WR];
#pragma unroll
    for (int j = 0; j < WR; ++j) {
        const int ch = lane + 64 * j;
        const int chg = chan_of(ch);
        w_ok[j] = (ch < BC) && (chg < p.Cdst);
        const int cc = w_ok[j] ? chg : 0;
        if (MODE == MODE_FWD) {
            w_a[j] = cc / p.OA;
            w_off[j] = (cc - w_a[j] * p.OA) * CK;
        } else {
            w_a[j] = cc / p.IA;
            w_off[j] = (cc - w_a[j] * p.IA) * (p.wt ? CK : KK);
        }
    }

    // ---- K range of this workgroup (dual-quaternion zero quadrant) ----------------------------------
    int kbeg = 0, kend = p.Ktot;
    if (halves_aligned && !paired) {
        if (p.skip_mode == 1 && c0 + BC <= half_c) kend = half_k;      // all channels primal
        if (p.skip_mode == 2 && c0 >= half_c) kbeg = half_k;           // all channels dual
    }
    const int nchunks = (kend - kbeg + 15) >> 4;
    // workgroup whose lower tiles are primal and upper tiles dual: half of its tiles skip the zero quadrant's
    // chunks (any other straddle just multiplies by the staged zeros)
    const bool mixed_wg = paired || (halves_aligned && (CT % 2 == 0) && (c0 + BC / 2 == half_c));

    // ---- scalar trackers of this wave's weight k-group start kw = kbeg + 16*chunk + 4*wave ----------
    // fwd  : kw = kq*CK + kl           (input component, local index)
    // dgrad: kw = (kp*OA + ko)*KK + kx (output component, output channel in component, tap)
    int kq = 0, kl = 0, kp = 0, ko = 0, kx = 0;
    {
        const int kw0 = kbeg + wave * 4;
        if (lin) {
            kq = kw0 / CK;
            kl = kw0 - kq * CK;
        } else {
            const int co = kw0 / KK;
            kx = kw0 - co * KK;
            kp = co / p.OA;
            ko = co - kp * p.OA;
        }
    }
    const bool w_vec = FAST ? true : (lin && ((CK & 3) == 0));

    float xr[XG][4];
    float wr[WR][4], wm[WR][4];      // staged weights and their sign / zero multipliers

    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
        const int kk0 = kbeg + chunk * 16;
        // X operand: im2col gather, coalesced along positions; kk is wave-uniform
#pragma unroll
        for (int j = 0; j < XG; ++j) {
            const int g = xg0 + j * XSTEP;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kk = kk0 + g * 4 + s;
                const int kkc = kk < kend ? kk : 0;
                const int chan = kkc / KK;
                const int kidx = kkc - chan * KK;
                const int kh = kidx / KW;
                const int kw = kidx - kh * KW;
                unsigned off;
                if (!strided) {
                    const int iwb = base_wb + kw * p.KDw * 4;
                    bool ok = (unsigned)iwb < (unsigned)srcWb;
                    int rowb = rowb0;
                    if (KH_T == 1) {
                        ok = ok && hok0;
                    } else {
                        const int khd = kh * p.KDh;
                        ok = ok && pvalid && ((unsigned)(base_h + khd) < (unsigned)p.srcH);
                        rowb = rowb0 + khd * srcWb;
                    }
                    off = (unsigned)(rowb + iwb + chan * p.srcS * 4);
                    if (!ok || kk >= kend) off = 0xFFFFFFFFu;
                } else {
                    int ih = base_h + kh * p.KDh;
                    int iw = base_w + kw * p.KDw;
                    bool ok = pvalid && (kk < kend) && (ih % p.SDh == 0) && (iw % p.SDw == 0);
                    ih /= p.SDh;
                    iw /= p.SDw;
                    ok = ok && ih >= 0 && ih < p.srcH && iw >= 0 && iw < p.srcW;
                    off = ok ? (unsigned)(img_b + (chan * p.srcS + ih * p.srcW + iw) * 4) : 0xFFFFFFFFu;
                }
                xr[j][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
            }
        }
        // W operand: gather from the component tensors, k-group = wave.  Loads are UNCONDITIONAL (an unused
        // element reads the component's first word) and the Hamilton sign / zero is kept as a multiplier that is
        // applied when the registers go to LDS: nothing between the load and the MFMAs waits on memory.
        typedef const __attribute__((address_space(1))) float* gptr;
        const int kw = kk0 + wave * 4;
        if (lin) {
            if (w_vec) {
                const bool kin = kw < kend;
#pragma unroll
                for (int j = 0; j < WR; ++j) {
                    bool zero, neg;
                    const int comp = (MODE == MODE_FWD) ? hc_comp(A, w_a[j], kq, &zero, &neg) : hc_comp(A, kq, w_a[j], &zero, &neg);
                    const bool use = kin && w_ok[j] && !zero;
                    gptr base = (gptr)wptr_s[comp];
                    const floatx4 v = *reinterpret_cast<const __attribute__((address_space(1))) floatx4*>(base + (use ? w_off[j] + kl : 0));
                    wr[j][0] = v[0]; wr[j][1] = v[1]; wr[j][2] = v[2]; wr[j][3] = v[3];
                    const float m = use ? (neg ? -1.f : 1.f) : 0.f;
                    wm[j][0] = m; wm[j][1] = m; wm[j][2] = m; wm[j][3] = m;
                }
            } else {
                int q = kq, l = kl;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bool kin = (kw + s) < kend;
#pragma unroll
                    for (int j = 0; j < WR; ++j) {
                        bool zero, neg;
                        const int comp = (MODE == MODE_FWD) ? hc_comp(A, w_a[j], q, &zero, &neg) : hc_comp(A, q, w_a[j], &zero, &neg);
                        const bool use = kin && w_ok[j] && !zero;
                        gptr base = (gptr)wptr_s[comp];
                        wr[j][s] = base[use ? w_off[j] + l : 0];
                        wm[j][s] = use ? (neg ? -1.f : 1.f) : 0.f;
                    }
                    if (++l >= CK) { l = 0; ++q; }
                }
            }
            kl += 16;
            if (FAST) {
                if (kl >= CK) { kl -= CK; ++kq; }          // CK >= 16: at most one wrap, a scalar select
            } else {
                while (kl >= CK) { kl -= CK; ++kq; }
            }
        } else {
            int pp = kp, o = ko, x = kx;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool kin = (kw + s) < kend;
                const int soff = o * p.IA * KK + x;
#pragma unroll
                for (int j = 0; j < WR; ++j) {
                    bool zero, neg;
                    const int comp = hc_comp(A, pp, w_a[j], &zero, &neg);
                    const bool use = kin && w_ok[j] && !zero;
                    gptr base = (gptr)wptr_s[comp];
                    wr[j][s] = base[use ? soff + w_off[j] : 0];
                    wm[j][s] = use ? (neg ? -1.f : 1.f) : 0.f;
                }
                if (++x >= KK) { x = 0; if (++o >= p.OA) { o = 0; ++pp; } }
            }
            // advance the tracker by 16 taps
            kx += 16;
            const int dco = kx / KK;
            kx -= dco * KK;
            ko += dco;
            while (ko >= p.OA) { ko -= p.OA; ++kp; }
        }
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < XG; ++j)
            *reinterpret_cast<float4*>(&Xs[buf][xg0 + j * XSTEP][xpos][0]) = make_float4(xr[j][0], xr[j][1], xr[j][2], xr[j][3]);
#pragma unroll
        for (int j = 0; j < WR; ++j) {
            const int ch = lane + 64 * j;
            if (ch < BC)
                *reinterpret_cast<float4*>(&Ws[buf][wave][ch][0]) =
                    make_float4(wr[j][0] * wm[j][0], wr[j][1] * wm[j][1], wr[j][2] * wm[j][2], wr[j][3] * wm[j][3]);
        }
    };

    floatx4 acc[PT][CT];
#pragma unroll
    for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15;      // row / col inside a 16x16 tile
    const int fk = lane >> 4;      // k group of this lane

    __syncthreads();               // wptr_s visible
    if (nchunks > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();

    // The K loop is run as up to two consecutive loops, each with a compile-time set of channel tiles
    // [J0, J1): a workgroup that straddles the primal/dual boundary computes all of its tiles over one K half
    // and only half of them over the other (the zero quadrant).  Each loop body is straight-line code (no
    // per-tile branches, accumulators stay in place).  s is outermost so that consecutive MFMAs hit different
    // accumulators (16x16x4 f32: 40-cycle dependent latency against a 32-cycle issue interval).
    auto run_chunks = [&](int cbeg, int cend, auto j0c, auto j1c) __attribute__((always_inline)) {
        constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
        for (int chunk = cbeg; chunk < cend; ++chunk) {
            const int buf = chunk & 1;
            if (chunk + 1 < nchunks) load_chunk(chunk + 1);
            float av[PT][4], bv[CT][4];
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(&Xs[buf][fk][wave * (PT * 16) + i * 16 + fr][0]);
                av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
            }
#pragma unroll
            for (int j = J0; j < J1; ++j) {
                const float4 t = *reinterpret_cast<const float4*>(&Ws[buf][fk][j * 16 + fr][0]);
                bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = J0; j < J1; ++j)
#pragma unroll
                    for (int i = 0; i < PT; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
            if (chunk + 1 < nchunks) store_chunk(buf ^ 1);
            __syncthreads();
        }
    };
    using IC0 = std::integral_constant<int, 0>;
    using ICH = std::integral_constant<int, CT / 2>;
    using ICT = std::integral_constant<int, CT>;
    if (mixed_wg) {
        const int csplit = (half_k - kbeg) >> 4;          // first chunk of the upper K half
        if (p.skip_mode == 1) {                           // forward: primal tiles (lower half) see zeros there
            run_chunks(0, csplit, IC0{}, ICT{});
            run_chunks(csplit, nchunks, ICH{}, ICT{});
        } else {                                          // dgrad: dual tiles (upper half) see zeros in the lower K half
            run_chunks(0, csplit, IC0{}, ICH{});
            run_chunks(csplit, nchunks, IC0{}, ICT{});
        }
    } else {
        run_chunks(0, nchunks, IC0{}, ICT{});
    }


    // ---- epilogue: lane holds positions prow..prow+3 (regs) of channel ch -----------------------------
    const bool vec_ok = (p.dstS % 4 == 0);
    // element offset of (first image of the tile, channel 0, position 0) and, per position tile i, of this
    // lane's 4 positions relative to it
    float* const dst0 = p.dst + (size_t)img0 * p.Cdst * p.dstS;
    const float* const add0 = p.addend ? p.addend + (size_t)img0 * p.Cdst * p.dstS : nullptr;
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
        const float bvv = (chok && p.bias) ? p.bias[ch] : 0.0f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const long long pos = p0 + wave * (PT * 16) + i * 16 + fk * 4;
            floatx4 v = acc[i][j];
            if (chok && pos < p.Ptot) {
                if (vec_ok) {
                    const size_t off = (size_t)(poff[i] + ch * p.dstS);
                    float4 o = make_float4(v[0] + bvv, v[1] + bvv, v[2] + bvv, v[3] + bvv);
                    if (p.epilogue & SELD_EPI_ADD) {
                        float4 ad = *reinterpret_cast<const float4*>(add0 + off);
                        o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
                    }
                    if (p.epilogue & SELD_EPI_ACCUMULATE) {
                        float4 old = *reinterpret_cast<const float4*>(dst0 + off);
                        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                    }
                    *reinterpret_cast<float4*>(dst0 + off) = o;
                    s1 += o.x + o.y + o.z + o.w;
                    s2 += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        long long pr = pos + r;
                        if (pr < p.Ptot) {
                            long long img = pr / p.dstS;
                            int rem = (int)(pr - img * p.dstS);
                            size_t off = ((size_t)img * p.Cdst + ch) * p.dstS + rem;
                            float o = v[r] + bvv;
                            if (p.epilogue & SELD_EPI_ADD) o += p.addend[off];
                            if (p.epilogue & SELD_EPI_ACCUMULATE) o += p.dst[off];
                            p.dst[off] = o;
                            s1 += o;
                            s2 += o * o;
                        }
                    }
                }
            }
        }
        if (p.epilogue & SELD_EPI_STATS) {
            // lanes fr, fr+16, fr+32, fr+48 hold the same channel; the 4 waves hold different positions
            s1 += __shfl_xor(s1, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (fk == 0) {
                float* redbuf = &Xs[0][0][0][0];       // the K loop is over: the staging buffers are free
                redbuf[(wave * BC + j * 16 + fr) * 2 + 0] = s1;
                redbuf[(wave * BC + j * 16 + fr) * 2 + 1] = s2;
            }
        }
    }
    if (p.epilogue & SELD_EPI_STATS) {
        // one atomic pair per channel per workgroup, spread over SELD_STATS_REPLICAS rows so that the
        // thousands of workgroups of a layer do not serialise on 2*C addresses
        __syncthreads();
        const float* redbuf = &Xs[0][0][0][0];
        float* rep = p.stats + (size_t)(blockIdx.x % SELD_STATS_REPLICAS) * 2 * p.Cdst;
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
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct TileCfg { int ct, pt; };

// Tile choice: fill >= 2 workgroups per CU, avoid padded channels, prefer tiles that restage the im2col
// operand least.  SELD_CONV_CFG="ct,pt" overrides (tuning aid).
static TileCfg pick_cfg(int C, long long P) {
    static const TileCfg cand[] = {{12, 1}, {12, 2}, {6, 1}, {4, 4}, {2, 4}, {1, 4}};
    static const double pref[] = {1.00, 0.95, 0.80, 0.75, 0.45, 0.30};   // measured ranking on the config-3 layers
    if (env().conv_cfg_ct)
        for (const TileCfg& c : cand)
            if (c.ct == env().conv_cfg_ct && c.pt == env().conv_cfg_pt) return c;
    double best = -1.0;
    TileCfg pick = cand[3];
    for (int i = 0; i < 6; ++i) {
        const int bc = cand[i].ct * 16, bp = cand[i].pt * 64;
        const long long cb = (C + bc - 1) / bc;
        const long long wgs = ((P + bp - 1) / bp) * cb;
        const double fill = wgs >= 512 ? 1.0 : (double)wgs / 512.0;
        const double use = (double)C / (double)(cb * bc);
        const double score = fill * use * pref[i];
        if (score > best) { best = score; pick = cand[i]; }
    }
    return pick;
}

// Wt[comp][c][o][k] = W[comp][o][c][k]: makes the data gradient's weight rows contiguous along its K axis
__global__ void hc_transpose_w_kernel(WPtrs w, int A, int OA, int IA, int KK, float* __restrict__ out) {
    const int per = OA * IA * KK;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per * A) return;
    const int comp = idx / per;
    int r = idx - comp * per;
    const int c = r / (OA * KK);
    r -= c * OA * KK;
    const int o = r / KK;
    const int k = r - o * KK;
    out[idx] = w.p[comp][((size_t)o * IA + c) * KK + k];
}

int hc_conv_vec_try(const ConvP& p, int mode, int ct, int pt, hipStream_t st);
int hc_conv_vec_chunk(const ConvP& p, int mode, int ct, int pt);

template <int CT, int PT, int MODE>
static void launch_conv(const ConvP& p, hipStream_t st) {
    constexpr int BC = CT * 16, BP = PT * 64;
    if (hc_conv_vec_try(p, MODE, CT, PT, st)) return;     // loop-invariant staging variant (hc_conv_vec.hip)
    dim3 grid((unsigned)((p.Ptot + BP - 1) / BP), (unsigned)((p.Cdst + BC - 1) / BC), 1);
    const int KK = p.KH * p.KW;
    const int CK = (MODE == MODE_FWD ? p.IA : p.OA) * KK;
    const bool fast = (MODE == MODE_FWD || p.wt) && (CK % 4 == 0) && CK >= 16 && p.SDh == 1 && p.SDw == 1 &&
                      !env().conv_nofast;
    if (fast && p.KH == 1 && p.KW == 1) hipLaunchKernelGGL((hc_conv_kernel<CT, PT, 1, 1, MODE, 1>), grid, dim3(256), 0, st, p);
    else if (fast && p.KH == 1 && p.KW == 3) hipLaunchKernelGGL((hc_conv_kernel<CT, PT, 1, 3, MODE, 1>), grid, dim3(256), 0, st, p);
    else if (fast && p.KH == 3 && p.KW == 3) hipLaunchKernelGGL((hc_conv_kernel<CT, PT, 3, 3, MODE, 1>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((hc_conv_kernel<CT, PT, 0, 0, MODE, 0>), grid, dim3(256), 0, st, p);
}

template <int MODE>
static int run_conv(ConvP& p, hipStream_t st) {
    const TileCfg c = pick_cfg(p.Cdst, p.Ptot);
    if (c.ct == 12 && c.pt == 2) launch_conv<12, 2, MODE>(p, st);
    else if (c.ct == 12 && c.pt == 1) launch_conv<12, 1, MODE>(p, st);
    else if (c.ct == 6 && c.pt == 1) launch_conv<6, 1, MODE>(p, st);
    else if (c.ct == 2) launch_conv<2, 4, MODE>(p, st);
    else if (c.ct == 1) launch_conv<1, 4, MODE>(p, st);
    else launch_conv<4, 4, MODE>(p, st);
    return check_launch();
}

int hc_conv_smallk_try(const ConvP& p, hipStream_t st, int* rc, int dry_run);

static void fill_common(ConvP& p, const seld_conv_desc* d, const float* const w[8]) {
    p.algebra = d->algebra;
    p.pairing = env().conv_pair ? 1 : 0;   // measured 5-14 % slower on the TCN layers: off by default
    p.KH = d->k[0]; p.KW = d->k[1];
    p.OA = d->Cout / d->algebra; p.IA = d->Cin / d->algebra;
    for (int i = 0; i < 8; ++i) p.w.p[i] = (w && i < d->algebra) ? w[i] : nullptr;
}

static void fill_fwd(ConvP& p, const seld_conv_desc* d, const float* const w[8], const int o[2]) {
    fill_common(p, d, w);
    p.mode = MODE_FWD;
    p.Csrc = d->Cin; p.Cdst = d->Cout;
    p.srcH = d->in[0]; p.srcW = d->in[1]; p.dstH = o[0]; p.dstW = o[1];
    p.SMh = d->stride[0]; p.OFFh = -d->pad[0]; p.KDh = d->dil[0]; p.SDh = 1;
    p.SMw = d->stride[1]; p.OFFw = -d->pad[1]; p.KDw = d->dil[1]; p.SDw = 1;
    p.Ktot = d->Cin * p.KH * p.KW;
    p.srcS = p.srcH * p.srcW; p.dstS = p.dstH * p.dstW;
    p.Ptot = (long long)d->N * p.dstS;
    p.src_elems = (long long)d->N * p.Csrc * p.srcS;
    p.skip_mode = (d->algebra == 8) ? 1 : 0;
}

static void fill_dgrad(ConvP& p, const seld_conv_desc* d, const float* const w[8], const int o[2]) {
    fill_common(p, d, w);
    p.mode = MODE_DGRAD;
    p.Csrc = d->Cout; p.Cdst = d->Cin;
    p.srcH = o[0]; p.srcW = o[1]; p.dstH = d->in[0]; p.dstW = d->in[1];
    // oh = (ih + pad - kh*dil) / stride
    p.SMh = 1; p.OFFh = d->pad[0]; p.KDh = -d->dil[0]; p.SDh = d->stride[0];
    p.SMw = 1; p.OFFw = d->pad[1]; p.KDw = -d->dil[1]; p.SDw = d->stride[1];
    p.Ktot = d->Cout * p.KH * p.KW;
    p.srcS = p.srcH * p.srcW; p.dstS = p.dstH * p.dstW;
    p.Ptot = (long long)d->N * p.dstS;
    p.src_elems = (long long)d->N * p.Csrc * p.srcS;
    p.skip_mode = (d->algebra == 8) ? 2 : 0;
    p.epilogue = 0;
}

}  // namespace seld

using namespace seld;

extern "C" int seld_hc_conv_out_shape(const seld_conv_desc* d, int32_t out[2]) {
    int rc = hc_validate(d);
    if (rc) return rc;
    hc_out_shape(d, out);
    return (out[0] > 0 && out[1] > 0) ? SELD_OK : SELD_EINVAL;
}

extern "C" int seld_hc_conv_fwd_ex(const seld_conv_desc* d, const float* x, const float* const w[8],
                                   const float* bias, float* y, int32_t epilogue, const float* addend,
                                   float* stats, void* stream) {
    int rc = hc_validate(d);
    if (rc) return rc;
    int o[2];
    hc_out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0 || !x || !w || !y) return SELD_EINVAL;
    if ((epilogue & SELD_EPI_ADD) && !addend) return SELD_EINVAL;
    if ((epilogue & SELD_EPI_STATS) && !stats) return SELD_EINVAL;
    ConvP p{};
    fill_fwd(p, d, w, o);
    p.epilogue = epilogue;
    p.src = x; p.bias = bias; p.dst = y; p.addend = addend; p.stats = stats;
    int rc2 = SELD_OK;
    p.wt = env().smallk_dbg;          // non-zero only in -DSELD_TUNING builds (timing experiments)
    if (hc_conv_smallk_try(p, (hipStream_t)stream, &rc2, 0)) return rc2;       // short reductions: persistent kernel
    return run_conv<MODE_FWD>(p, (hipStream_t)stream);
}

extern "C" int seld_hc_conv_fwd(const seld_conv_desc* d, const float* x, const float* const w[8],
                                const float* bias, float* y, void* stream) {
    return seld_hc_conv_fwd_ex(d, x, w, bias, y, SELD_EPI_NONE, nullptr, nullptr, stream);
}

extern "C" size_t seld_hc_conv_bwd_data_workspace(const seld_conv_desc* d) {
    if (hc_validate(d)) return 0;
    return (size_t)d->Cout * (d->Cin / d->algebra) * d->k[0] * d->k[1] * sizeof(float);
}

extern "C" int seld_hc_conv_bwd_data(const seld_conv_desc* d, const float* dy, const float* const w[8],
                                     float* dx, void* stream) {
    return seld_hc_conv_bwd_data_ex(d, dy, w, dx, nullptr, 0, stream);
}

extern "C" int seld_hc_conv_bwd_data_ex(const seld_conv_desc* d, const float* dy, const float* const w[8],
                                        float* dx, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = hc_validate(d);
    if (rc) return rc;
    int o[2];
    hc_out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0 || !dy || !w || !dx) return SELD_EINVAL;
    if ((long long)d->Cout * o[0] * o[1] >= (1LL << 28)) return SELD_EUNSUPPORTED;
    ConvP p{};
    fill_dgrad(p, d, w, o);
    p.src = dy; p.bias = nullptr; p.dst = dx;
    const size_t need = seld_hc_conv_bwd_data_workspace(d);
    if (workspace && workspace_bytes >= need) {
        // fast path: transposed component tensors in the workspace, staged with 16-byte loads like the forward
        const int per = p.OA * p.IA * p.KH * p.KW;
        float* wt = (float*)workspace;
        hipLaunchKernelGGL(hc_transpose_w_kernel, dim3((per * d->algebra + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           p.w, d->algebra, p.OA, p.IA, p.KH * p.KW, wt);
        rc = check_launch();
        if (rc) return rc;
        for (int i = 0; i < d->algebra; ++i) p.w.p[i] = wt + (size_t)i * per;
        p.wt = 1;
    }
    return run_conv<MODE_DGRAD>(p, (hipStream_t)stream);
}

// The data gradient in two steps, so that the weight re-layout can be done ahead of time (the host mirror issues it
// on a side stream during the forward pass): workspace <- Wt[comp][c][o][k] = W[comp][o][c][k], then the data gradient
// from that workspace.  seld_hc_conv_bwd_data_ex does both.
extern "C" int seld_hc_conv_transpose_weights(const seld_conv_desc* d, const float* const w[8], void* workspace,
                                              size_t workspace_bytes, void* stream) {
    int rc = hc_validate(d);
    if (rc) return rc;
    if (!w || !workspace || workspace_bytes < seld_hc_conv_bwd_data_workspace(d)) return SELD_EINVAL;
    WPtrs src{};
    for (int i = 0; i < 8; ++i) src.p[i] = (i < d->algebra) ? w[i] : nullptr;
    const int OA = d->Cout / d->algebra, IA = d->Cin / d->algebra, KK = d->k[0] * d->k[1];
    const int per = OA * IA * KK;
    hipLaunchKernelGGL(hc_transpose_w_kernel, dim3((per * d->algebra + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       src, d->algebra, OA, IA, KK, (float*)workspace);
    return check_launch();
}

extern "C" int seld_hc_conv_bwd_data_wt(const seld_conv_desc* d, const float* dy, const void* wt_workspace, float* dx,
                                        void* stream) {
    int rc = hc_validate(d);
    if (rc) return rc;
    int o[2];
    hc_out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0 || !dy || !wt_workspace || !dx) return SELD_EINVAL;
    if ((long long)d->Cout * o[0] * o[1] >= (1LL << 28)) return SELD_EUNSUPPORTED;
    ConvP p{};
    fill_dgrad(p, d, nullptr, o);
    p.src = dy; p.bias = nullptr; p.dst = dx;
    const int per = p.OA * p.IA * p.KH * p.KW;
    for (int i = 0; i < d->algebra; ++i) p.w.p[i] = (const float*)wt_workspace + (size_t)i * per;
    p.wt = 1;
    return run_conv<MODE_DGRAD>(p, (hipStream_t)stream);
}

namespace seld {
int hc_wgrad_label(const seld_conv_desc* d, char* buf, int buflen);
int hc_wgrad_pair_ok(const seld_conv_desc* d);
}

// ---- two convolutions of one geometry in one launch (filter | gate, skip | residual of a residual block) ---------
// which: 0 forward, 1 data gradient, 2 weight gradient.  1 if the pair entry point below will run, else 0 (the caller
// then issues the two single calls).
extern "C" int seld_hc_conv_pair_supported(const seld_conv_desc* d, int32_t which) {
    if (hc_validate(d)) return 0;
    int o[2];
    hc_out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0) return 0;
    if (which == 2) return hc_wgrad_pair_ok(d);
    if (which == 0) return 1;                      // forward pairs are two launches of the single entry point
    ConvP p{};
    if (which == 0) fill_fwd(p, d, nullptr, o);
    else { fill_dgrad(p, d, nullptr, o); p.wt = 1; }
    p.nslots = 2;
    if (which == 0) { int dummy; if (hc_conv_smallk_try(p, nullptr, &dummy, 1)) return 0; }
    const TileCfg c = pick_cfg(p.Cdst, p.Ptot);
    return hc_conv_vec_chunk(p, which, c.ct, c.pt) ? 1 : 0;
}

extern "C" int seld_hc_conv_pair_fwd(const seld_conv_desc* d, const float* x, const float* const wA[8],
                                     const float* const wB[8], const float* biasA, const float* biasB, float* yA,
                                     float* yB, int32_t epilogueA, int32_t epilogueB, const float* addendA,
                                     const float* addendB, float* statsA, float* statsB, void* stream) {
    int rc = hc_validate(d);
    if (rc) return rc;
    int o[2];
    hc_out_shape(d, o);
    if (o[0] <= 0 || o[1] <= 0 || !x || !wA || !wB || !yA || !yB) return SELD_EINVAL;
    if (((epilogueA & SELD_EPI_ADD) && !addendA) || ((epilogueB & SELD_EPI_ADD) && !addendB)) return SELD_EINVAL;
    if (((epilogueA & SELD_EPI_STATS) && !statsA) || ((epilogueB & SELD_EPI_STATS) && !statsB)) return SELD_EINVAL;
    // One launch (two K passes per workgroup) on the 1x1 layers, where a launch's fixed cost is a third of its time;
    // two launches of the single kernel everywhere else (on the 1x3 layers the pair measured 141 us against 2 x 68).
    ConvP p{};
    fill_fwd(p, d, wA, o);
    p.epilogue = epilogueA; p.src = x; p.bias = biasA; p.dst = yA; p.addend = addendA; p.stats = statsA;
    p.nslots = 2;
    for (int i = 0; i < 8; ++i) p.w2.p[i] = (i < d->algebra) ? wB[i] : nullptr;
    p.epilogue2 = epilogueB; p.src2 = x; p.bias2 = biasB; p.dst2 = yB; p.addend2 = addendB; p.stats2 = statsB;
    const TileCfg c = pick_cfg(p.Cdst, p.Ptot);
    if (!env().no_fwd_pair && hc_conv_vec_try(p, MODE_FWD, c.ct, c.pt, (hipStream_t)stream)) return check_launch();
    rc = seld_hc_conv_fwd_ex(d, x, wA, biasA, yA, epilogueA, addendA, statsA, stream);
    if (rc) return rc;
    return seld_hc_conv_fwd_ex(d, x, wB, biasB, yB, epilogueB, addendB, statsB, stream);
}

// dx = dgrad(dyA, wA) + dgrad(dyB, wB) from two workspaces filled by seld_hc_conv_transpose_weights
extern "C" int seld_hc_conv_pair_bwd_data_wt(const seld_conv_desc* d, const float* dyA, const float* dyB,
                                             const void* wtA, const void* wtB, float* dx, void* stream) {
    if (!seld_hc_conv_pair_supported(d, 1)) return SELD_EUNSUPPORTED;
    if (!dyA || !dyB || !wtA || !wtB || !dx) return SELD_EINVAL;
    int o[2];
    hc_out_shape(d, o);
    if ((long long)d->Cout * o[0] * o[1] >= (1LL << 28)) return SELD_EUNSUPPORTED;
    ConvP p{};
    fill_dgrad(p, d, nullptr, o);
    p.src = dyA; p.bias = nullptr; p.dst = dx;
    p.nslots = 2; p.src2 = dyB;
    const int per = p.OA * p.IA * p.KH * p.KW;
    for (int i = 0; i < d->algebra; ++i) {
        p.w.p[i] = (const float*)wtA + (size_t)i * per;
        p.w2.p[i] = (const float*)wtB + (size_t)i * per;
    }
    p.wt = 1;
    const TileCfg c = pick_cfg(p.Cdst, p.Ptot);
    if (!hc_conv_vec_try(p, MODE_DGRAD, c.ct, c.pt, (hipStream_t)stream)) return SELD_EUNSUPPORTED;
    return check_launch();
}

// dx = dgrad(dyA, wA) + dgrad(dyB, wB).  workspace: 2 * seld_hc_conv_bwd_data_workspace(d) bytes (required).
extern "C" int seld_hc_conv_pair_bwd_data(const seld_conv_desc* d, const float* dyA, const float* dyB,
                                          const float* const wA[8], const float* const wB[8], float* dx,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!seld_hc_conv_pair_supported(d, 1)) return SELD_EUNSUPPORTED;
    if (!dyA || !dyB || !wA || !wB || !dx) return SELD_EINVAL;
    const size_t need = seld_hc_conv_bwd_data_workspace(d);
    if (!workspace || workspace_bytes < 2 * need) return SELD_EINVAL;
    int o[2];
    hc_out_shape(d, o);
    if ((long long)d->Cout * o[0] * o[1] >= (1LL << 28)) return SELD_EUNSUPPORTED;
    ConvP p{};
    fill_dgrad(p, d, wA, o);
    p.src = dyA; p.bias = nullptr; p.dst = dx;
    p.nslots = 2; p.src2 = dyB;
    const int per = p.OA * p.IA * p.KH * p.KW;
    float* wt = (float*)workspace;
    for (int sl = 0; sl < 2; ++sl) {
        WPtrs src{};
        for (int i = 0; i < 8; ++i) src.p[i] = (i < d->algebra) ? (sl ? wB[i] : wA[i]) : nullptr;
        float* out = wt + (size_t)sl * d->algebra * per;
        hipLaunchKernelGGL(hc_transpose_w_kernel, dim3((per * d->algebra + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           src, d->algebra, p.OA, p.IA, p.KH * p.KW, out);
        int rc = check_launch();
        if (rc) return rc;
        for (int i = 0; i < d->algebra; ++i) (sl ? p.w2 : p.w).p[i] = out + (size_t)i * per;
    }
    p.wt = 1;
    const TileCfg c = pick_cfg(p.Cdst, p.Ptot);
    if (!hc_conv_vec_try(p, MODE_DGRAD, c.ct, c.pt, (hipStream_t)stream)) return SELD_EUNSUPPORTED;
    return check_launch();
}

// Label of the kernel symbol a call would launch (as rocprofv3 prints the template arguments);
// which = 0 forward, 1 data gradient, 2 weight gradient.
extern "C" int seld_hc_conv_kernel_label(const seld_conv_desc* d, int32_t which, char* buf, int32_t buflen) {
    int rc = hc_validate(d);
    if (rc || !buf || buflen < 48) return SELD_EINVAL;
    if (which == 2) return hc_wgrad_label(d, buf, buflen);
    int kh = d->k[0], kw = d->k[1];
    if (!((kh == 1 && kw == 1) || (kh == 1 && kw == 3) || (kh == 3 && kw == 3))) kh = kw = 0;
    int o[2];
    hc_out_shape(d, o);
    ConvP p{};
    if (which == 0) fill_fwd(p, d, nullptr, o);
    else { fill_dgrad(p, d, nullptr, o); p.wt = 1; }          // the host mirror always supplies the transposed-weight workspace
    const long long P = p.Ptot;
    if (which == 0 && d->stride[0] == 1 && d->stride[1] == 1) {
        int dummy;
        const int ct = hc_conv_smallk_try(p, nullptr, &dummy, 1);
        if (ct) {
            const bool t33 = d->k[0] == 3 && d->k[1] == 3, t13 = d->k[0] == 1 && d->k[1] == 3;
            snprintf(buf, buflen, "hc_conv_smallk_kernel<%d, %d, %d>", ct, t33 ? 3 : (t13 ? 1 : 0), (t33 || t13) ? 3 : 0);
            return SELD_OK;
        }
    }
    const TileCfg c = pick_cfg(which == 0 ? d->Cout : d->Cin, P);
    if (const int kc = hc_conv_vec_chunk(p, which, c.ct, c.pt)) {
        snprintf(buf, buflen, "hc_conv_vec_kernel<%d, %d, %d, %d, %d, %d, %d>", c.ct, c.pt, kh, kw, which, kc, 0);
        return SELD_OK;
    }
    const int CKl = ((which == 0 ? d->Cin : d->Cout) / d->algebra) * d->k[0] * d->k[1];
    const bool fast = (CKl % 4 == 0) && CKl >= 16 && (which == 0 || (d->stride[0] == 1 && d->stride[1] == 1)) && kh != 0 &&
                      !env().conv_nofast;
    snprintf(buf, buflen, "hc_conv_kernel<%d, %d, %d, %d, %d, %d>", c.ct, c.pt, fast ? kh : 0, fast ? kw : 0, which, fast ? 1 : 0);
    return SELD_OK;
}
