// Post-processing + test metrics on the device.
//
// Replaces, for a batch of recordings whose network outputs are already resident:
//   gen_submission_list_task2      utility_functions.py:184-210   (threshold + decode of the (T, 42) / (T, 126) outputs)
//   location_sensitive_detection   metrics.py:123-182             (frame-wise TP / FP / FN, L3DAS21)
//   segment_labels                 Dcase21_metrics.py:239-278     (1-second blocks, class-wise)
//   SELDMetrics.update_seld_scores Dcase21_metrics.py:51-154      (track association, DCASE21 counters)
// The reference walks Python dictionaries frame by frame on the host (tens of ms per recording); here one wave owns one
// (recording, block of `frames_per_block` frames) at a time: rows staged in LDS with coalesced loads, the activity of
// a frame as a 64-bit ballot mask, counters kept in registers and added to 13 int64 totals + 1 double at the end.
// HBM-bound by construction: 32 n bytes per frame (n = classes * overlaps), read once.
//
// Semantics kept on purpose (they are what the reference computes, see oracle/seld_oracle.py lsd_counts):
//   * an activity is "on" when np.round(value) != 0 (half to even: 0.5 is off) and the frame's rounded activities do
//     not sum to zero;
//   * location_sensitive_detection counts a frame's predictions TWICE as false positives when the frame has no
//     reference event, and a frame's references TWICE as false negatives when it has no prediction;
//   * the Hungarian association of <= 3 reference and <= 3 predicted DOAs of one class in one frame is solved by
//     enumeration (first minimum in lexicographic order; scipy may pick another optimum only on exact cost ties).
// Coordinates are float32(doa * max_loc_value) widened to double, distances in double as in the reference.
#include "common.h"

namespace seld {

constexpr int NUM_COUNTERS = 13;      // TP FP FN | dcase: TP FP FN S D I Nref DE_TP DE_FP DE_FN

struct MetricsP {
    const float* sed;       // (clips, frames, n)
    const float* doa;       // (clips, frames, 3n)
    const float* target;    // (clips, frames, 4n) = [activity | location]
    int clips, frames, classes, overlaps, fpb, blocks;
    float max_loc;
    double spatial_threshold, doa_threshold;
    long long* counters;
    double* total_de;
};

struct __attribute__((packed, aligned(4))) Xyz {
    float v[3];                                         // one 12-byte load (global_load_dwordx3), dword aligned
};

__device__ __forceinline__ void load_xyz(const float* loc, int slot, float max_loc, double v[3]) {
    const Xyz t = *reinterpret_cast<const Xyz*>(loc + slot * 3);
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = (double)(t.v[k] * max_loc);
}

// Dcase21_metrics.py:171-188
__device__ __forceinline__ double angular_distance_deg(const double a[3], const double b[3]) {
#pragma clang fp contract(off)
    const double n1 = sqrt(((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]) + 1e-10);
    const double n2 = sqrt(((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]) + 1e-10);
    double d = ((a[0] / n1) * (b[0] / n2) + (a[1] / n1) * (b[1] / n2)) + (a[2] / n1) * (b[2] / n2);
    d = fmin(fmax(d, -1.0), 1.0);
    return acos(d) * 180.0 / 3.141592653589793;
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int src_lane) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, src_lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), src_lane);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One WAVE per unit = (recording, block of fpb frames); a wave walks units u, u + total_waves, ... and keeps its
// counters in registers until the end (13 + 1 atomics per wave in all).  Per unit:
//   1. the block's rows (fpb * 8n floats, three contiguous runs in memory) are copied to the wave's LDS slice with
//      coalesced loads;
//   2. activity masks: lane j < n rounds element j of a row, a ballot makes the frame's 64-bit mask, a wave sum of the
//      rounded values applies the "sum == 0 -> no event" rule; every lane ends up with all masks;
//   3. location_sensitive_detection: lane j = reference slot j looks for a prediction of its class within the threshold;
//   4. DCASE21 block metrics: lane c = class c walks the block's frames (<= 3 x 3 association per frame);
//      loc_FP / loc_FN are summed over the classes with a wave reduction before S / D / I.
__global__ __launch_bounds__(256) void metrics_kernel(const MetricsP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = p.classes * p.overlaps;
    // masks (u64) | costs (f64 [class][frame][3][3]) | pair list (u32), later matches (f64 [class][frame][3])
    const int slice_floats = 4 * p.fpb + 27 * p.classes * p.fpb;
    unsigned long long* mp_s = reinterpret_cast<unsigned long long*>(smem + (size_t)wave * slice_floats);
    unsigned long long* mt_s = mp_s + p.fpb;
    double* cost_s = reinterpret_cast<double*>(mt_s + p.fpb);
    double* match_s = cost_s + 9 * p.classes * p.fpb;
    unsigned* list_s = reinterpret_cast<unsigned*>(match_s);     // dead before match_s is written
    const long long total = (long long)p.clips * p.blocks;
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    long long cnt[NUM_COUNTERS];
#pragma unroll
    for (int i = 0; i < NUM_COUNTERS; ++i) cnt[i] = 0;      // lane-local partial sums, reduced over the wave at the end
    double total_de = 0.0;
    const unsigned long long cls_mask = (1ull << p.overlaps) - 1ull;

    for (long long u = (long long)blockIdx.x * (blockDim.x >> 6) + wave; u < total; u += n_waves) {
        const int clip = (int)(u / p.blocks), blk = (int)(u % p.blocks);
        const int f0 = blk * p.fpb;
        const int nf = min(p.fpb, p.frames - f0);
        const float* sed = p.sed + ((size_t)clip * p.frames + f0) * n;
        const float* doa = p.doa + ((size_t)clip * p.frames + f0) * 3 * n;
        const float* tgt = p.target + ((size_t)clip * p.frames + f0) * 4 * n;
        // the frames' activity masks, kept in LDS.  Activities are read straight from memory (one coalesced row per
        // frame, 8 frames in flight); coordinates are fetched only for the few active slots further down - the rows
        // are not staged in LDS, which would cap the CU at 5 waves.
        for (int fb = 0; fb < nf; fb += 8) {
            float vp[8], vt[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int f = min(fb + k, nf - 1);
                vp[k] = lane < n ? sed[(size_t)f * n + lane] : 0.f;
                vt[k] = lane < n ? tgt[(size_t)f * 4 * n + lane] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int f = fb + k;
                if (f >= nf) break;                             // uniform
                const float rp = rintf(vp[k]), rt = rintf(vt[k]);       // round half to even, as np.round
                const unsigned long long bp = __ballot(rp != 0.f), bt = __ballot(rt != 0.f);
                const bool zp = wave_sum_f32(rp) == 0.f, zt = wave_sum_f32(rt) == 0.f;
                if (lane == 0) {
                    mp_s[f] = zp ? 0ull : bp;
                    mt_s[f] = zt ? 0ull : bt;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the wave's own LDS stores have landed

        // ---- location_sensitive_detection ----
        // per frame (lane = frame): the counts that do not depend on the matching
        for (int f = lane; f < nf; f += 64) {
            const int n_p = __popcll(mp_s[f]), n_t = __popcll(mt_s[f]);
            if (n_t == 0) {
                cnt[1] += 2 * n_p;
            } else if (n_p == 0) {
                cnt[2] += 2 * n_t;
            } else {
                cnt[2] += n_t;          // FN += n_t - matched, FP += n_p - matched: the matches are subtracted below
                cnt[1] += n_p;
            }
        }
        // per (frame, reference slot): is there a prediction of its class within the threshold?  All twelve coordinate
        // loads of an item are issued together (clamped addresses, results masked): one memory latency per item.
        for (int base = 0; base < nf * n; base += 64) {
            const int item = min(base + lane, nf * n - 1);
            const int f = item / n, j = item - f * n;
            const unsigned long long mp_f = mp_s[f];
            const int c = j / p.overlaps;
            const bool act = base + lane < nf * n && ((mt_s[f] >> j) & 1ull) && mp_f != 0ull;
            const float* lt = tgt + (size_t)f * 4 * n + n + (size_t)j * 3;
            const float* lp = doa + (size_t)f * 3 * n + (size_t)c * p.overlaps * 3;
            Xyz tq = {{0.f, 0.f, 0.f}}, qq[3] = {{{0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f}}};
            if (act) {
                tq = *reinterpret_cast<const Xyz*>(lt);
#pragma unroll
                for (int e = 0; e < 3; ++e)
                    if (e < p.overlaps) qq[e] = *reinterpret_cast<const Xyz*>(lp + e * 3);
            }
            const float* tv = tq.v;
            bool match = false;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const bool on = act && e < p.overlaps && ((mp_f >> (c * p.overlaps + e)) & 1ull);
                const double dx = (double)(tv[0] * p.max_loc) - (double)(qq[e].v[0] * p.max_loc);
                const double dy = (double)(tv[1] * p.max_loc) - (double)(qq[e].v[1] * p.max_loc);
                const double dz = (double)(tv[2] * p.max_loc) - (double)(qq[e].v[2] * p.max_loc);
                if (on && sqrt(dx * dx + dy * dy + dz * dz) < p.spatial_threshold) match = true;
            }
            if (match) {
                cnt[0] += 1;
                cnt[2] -= 1;
                cnt[1] -= 1;
            }
        }
        // ---- DCASE21 segment metrics ----
        // step 1a (lane = (class, frame)): list the (reference slot, predicted slot) pairs that need a distance.  Only
        // ~1 % of the 9 * classes * frames candidates exist; computing them where they fall would make every lane
        // walk the fp64 acos / sqrt / divide code of all nine positions (that was 90 % of the kernel's time).
        int n_pairs = 0;
        for (int base = 0; base < p.classes * nf; base += 64) {
            const int item = base + lane;
            unsigned g_bits = 0, p_bits = 0;
            int c = 0, f = 0;
            if (item < p.classes * nf) {
                c = item / nf;
                f = item - c * nf;
                g_bits = (unsigned)((mt_s[f] >> (c * p.overlaps)) & cls_mask);
                p_bits = (unsigned)((mp_s[f] >> (c * p.overlaps)) & cls_mask);
            }
            const int mine = __popc(g_bits) * __popc(p_bits);
            int incl = mine;                                    // inclusive prefix sum over the lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o, 64);
                if (lane >= o) incl += v;
            }
            int pos = n_pairs + incl - mine;
#pragma unroll
            for (int e = 0; e < 3; ++e)
#pragma unroll
                for (int e2 = 0; e2 < 3; ++e2)
                    if (((g_bits >> e) & 1u) && ((p_bits >> e2) & 1u)) list_s[pos++] = (unsigned)(c << 16 | f << 8 | e << 2 | e2);
            n_pairs += __shfl(incl, 63, 64);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // step 1b (lane = pair): one angular distance per lane
        for (int i = lane; i < n_pairs; i += 64) {
            const unsigned d = list_s[i];
            const int c = d >> 16, f = (d >> 8) & 255, e = (d >> 2) & 3, e2 = d & 3;
            double a[3], b[3];
            load_xyz(tgt + (size_t)f * 4 * n + n, c * p.overlaps + e, p.max_loc, a);
            load_xyz(doa + (size_t)f * 3 * n, c * p.overlaps + e2, p.max_loc, b);
            cost_s[((size_t)c * p.fpb + f) * 9 + e * 3 + e2] = angular_distance_deg(a, b);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // step 1c (lane = (class, frame)): the <= 3 x 3 association of that frame
        for (int item = lane; item < p.classes * nf; item += 64) {
            const int c = item / nf, f = item - c * nf;
            const int sh = c * p.overlaps;
            const unsigned g_bits = (unsigned)((mt_s[f] >> sh) & cls_mask), p_bits = (unsigned)((mp_s[f] >> sh) & cls_mask);
            const int g = __popc(g_bits), q = __popc(p_bits);
            // Everything below is indexed with compile-time constants (registers, no scratch): the cost matrix lives on
            // the 3 x 3 event SLOTS, absent slots are masked out of the enumeration instead of being compacted away.
            double o0 = -1.0, o1 = -1.0, o2 = -1.0;              // matched distance of reference track 0 / 1 / 2, -1 = none
            if (g && q) {
                const double* cs = cost_s + ((size_t)c * p.fpb + f) * 9;
                double cost[3][3];
#pragma unroll
                for (int e = 0; e < 3; ++e)
#pragma unroll
                    for (int e2 = 0; e2 < 3; ++e2)
                        cost[e][e2] = (((g_bits >> e) & 1u) && ((p_bits >> e2) & 1u)) ? cs[e * 3 + e2] : 0.0;
                // all 6 row -> column maps of the slots; a map counts when it pairs min(g, q) present rows with present
                // columns (a maximum matching); the cheapest one wins, the first on ties
                const int need = min(g, q);
                int best = -1;
                double best_cost = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    constexpr int P[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
                    int pairs = 0;
                    double tot = 0.0;
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const bool on = ((g_bits >> e) & 1u) && ((p_bits >> P[k][e]) & 1u);
                        pairs += on ? 1 : 0;
                        tot += on ? cost[e][P[k][e]] : 0.0;
                    }
                    if (pairs == need && (best < 0 || tot < best_cost)) {
                        best = k;
                        best_cost = tot;
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    constexpr int P[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
                    if (k != best) continue;
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        if (!(((g_bits >> e) & 1u) && ((p_bits >> P[k][e]) & 1u))) continue;
                        const int rank = __popc(g_bits & ((1u << e) - 1u));      // index of slot e among the present references
                        const double d = cost[e][P[k][e]];
                        o0 = rank == 0 ? d : o0;
                        o1 = rank == 1 ? d : o1;
                        o2 = rank == 2 ? d : o2;
                    }
                }
            }
            const double out[3] = {o0, o1, o2};
            double* dst = match_s + ((size_t)c * p.fpb + f) * 3;
            dst[0] = out[0];
            dst[1] = out[1];
            dst[2] = out[2];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);

        // ---- step 2: lane = class, tracks accumulated over the block's frames in frame order ----
        int loc_fn = 0, loc_fp = 0;
        if (lane < p.classes) {
            const int sh = lane * p.overlaps;
            int nb_gt = 0, nb_pred = 0;
#pragma unroll 5
            for (int f = 0; f < nf; ++f) {
                nb_gt = max(nb_gt, __popcll((mt_s[f] >> sh) & cls_mask));
                nb_pred = max(nb_pred, __popcll((mp_s[f] >> sh) & cls_mask));
            }
            cnt[9] += nb_gt;
            if (nb_gt && nb_pred) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
                int n0 = 0, n1 = 0, n2 = 0;
#pragma unroll 5
                for (int f = 0; f < nf; ++f) {
                    const double* src = match_s + ((size_t)lane * p.fpb + f) * 3;
                    const double d0 = src[0], d1 = src[1], d2 = src[2];
                    if (d0 >= 0.0) { s0 += d0; ++n0; }
                    if (d1 >= 0.0) { s1 += d1; ++n1; }
                    if (d2 >= 0.0) { s2 += d2; ++n2; }
                }
                if (n0 + n1 + n2 == 0) {
                    loc_fn += nb_pred;
                    cnt[5] += nb_pred;
                    cnt[12] += nb_pred;
                } else {
                    // (the reference adds the tracks' averages in order of first appearance; the order only moves the last
                    //  bit of _total_DE, which the cross-block atomics reorder anyway)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int tn = r == 0 ? n0 : r == 1 ? n1 : n2;
                        if (tn == 0) continue;
                        const double avg = (r == 0 ? s0 : r == 1 ? s1 : s2) / (double)tn;
                        total_de += avg;
                        cnt[10] += 1;
                        if (avg <= p.doa_threshold) {
                            cnt[3] += 1;
                        } else {
                            loc_fp += 1;
                            cnt[4] += 1;
                        }
                    }
                    if (nb_pred > nb_gt) {
                        loc_fp += nb_pred - nb_gt;
                        cnt[4] += nb_pred - nb_gt;
                        cnt[11] += nb_pred - nb_gt;
                    } else if (nb_pred < nb_gt) {
                        loc_fn += nb_gt - nb_pred;
                        cnt[5] += nb_gt - nb_pred;
                        cnt[12] += nb_gt - nb_pred;
                    }
                }
            } else if (nb_gt) {
                loc_fn += nb_gt;
                cnt[5] += nb_gt;
                cnt[12] += nb_gt;
            } else if (nb_pred) {
                loc_fp += nb_pred;
                cnt[4] += nb_pred;
                cnt[11] += nb_pred;
            }
        }
        const int blk_fn = wave_sum_i32(loc_fn), blk_fp = wave_sum_i32(loc_fp);
        if (lane == 0) {
            cnt[6] += min(blk_fp, blk_fn);
            cnt[7] += max(0, blk_fn - blk_fp);
            cnt[8] += max(0, blk_fp - blk_fn);
        }
        __builtin_amdgcn_wave_barrier();                // the slice is rewritten by the next unit
    }

#pragma unroll
    for (int i = 0; i < NUM_COUNTERS; ++i) {
        const long long s = wave_sum_i64(cnt[i]);
        if (lane == 0 && s != 0) atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + i), (unsigned long long)s);
    }
    const double de = wave_sum_d(total_de);
    if (lane == 0 && de != 0.0) atomicAdd(p.total_de, de);
}

}  // namespace seld

using namespace seld;

extern "C" int seld_metrics_accumulate(const float* sed, const float* doa, const float* target, int32_t clips, int32_t frames,
                                       int32_t num_frames, int32_t classes, int32_t overlaps, float max_loc_value,
                                       double spatial_threshold, double doa_threshold, int32_t frames_per_block,
                                       int64_t* counters, double* total_de, void* stream) {
    if (clips < 0 || frames < 0 || classes <= 0 || overlaps <= 0 || frames_per_block <= 0 || !counters || !total_de)
        return SELD_EINVAL;
    if (frames > num_frames) return SELD_EINVAL;       // the reference indexes frames[i[0]] with i[0] < n_frames only
    if (overlaps > 3 || classes * overlaps > 64 || frames_per_block > 64) return SELD_EUNSUPPORTED;   // masks, pair descriptors
    if (clips == 0 || frames == 0) return SELD_OK;
    if (!sed || !doa || !target) return SELD_EINVAL;
    MetricsP p;
    p.sed = sed;
    p.doa = doa;
    p.target = target;
    p.clips = clips;
    p.frames = frames;
    p.classes = classes;
    p.overlaps = overlaps;
    p.fpb = frames_per_block;
    p.blocks = (frames + frames_per_block - 1) / frames_per_block;
    p.max_loc = max_loc_value;
    p.spatial_threshold = spatial_threshold;
    p.doa_threshold = doa_threshold;
    p.counters = reinterpret_cast<long long*>(counters);
    p.total_de = total_de;
    const long long total = (long long)clips * p.blocks;
    // per wave: 2 fpb masks (u64), 9 classes fpb costs (f64), 9 classes fpb pair descriptors / 3 classes fpb matched
    // distances (f64)
    const size_t slice = (4 * (size_t)frames_per_block + 27 * (size_t)classes * frames_per_block) * sizeof(float);
    const int wpw = 1;                                   // waves per workgroup (15 KB of LDS at 14 x 3 x 10: 10 workgroups per CU)
    const size_t smem = wpw * slice;
    if (smem > 64 * 1024) return SELD_EUNSUPPORTED;
    long long wgs = (total + wpw - 1) / wpw;
    if (wgs > 4096) wgs = 4096;                          // 2 generations of resident waves; waves loop over their units
    hipLaunchKernelGGL(metrics_kernel, dim3((unsigned)wgs), dim3(64 * wpw), smem, (hipStream_t)stream, p);
    return check_launch();
}
