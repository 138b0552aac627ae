// Post-processing + test metrics on the device.
//
// Replaces, for a batch of recordings whose network outputs are already resident:
//   gen_submission_list_task2      utility_functions.py:184-210   (threshold + decode of the (T, 42) / (T, 126) outputs)
//   location_sensitive_detection   metrics.py:123-182             (frame-wise TP / FP / FN, L3DAS21)
//   segment_labels                 Dcase21_metrics.py:239-278     (1-second blocks, class-wise)
//   SELDMetrics.update_seld_scores Dcase21_metrics.py:51-154      (track association, DCASE21 counters)
// The reference walks Python dictionaries frame by frame on the host (seconds per recording); here one thread owns one
// (recording, block of `frames_per_block` frames), keeps the activity of its frames as 64-bit masks and adds its
// counters to 13 int64 totals + 1 double (wave reduction first, then one atomic per wave and counter).
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

constexpr int MAX_BLOCK_FRAMES = 16;
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

__device__ __forceinline__ unsigned long long activity_mask(const float* __restrict__ row, int n) {
    unsigned long long m = 0;
    float sum = 0.f;
    for (int j = 0; j < n; ++j) {
        const float r = rintf(row[j]);          // round half to even, as np.round
        sum += r;
        if (r != 0.f) m |= 1ull << j;
    }
    return sum == 0.f ? 0ull : m;
}

__device__ __forceinline__ void load_xyz(const float* __restrict__ loc, int slot, float max_loc, double v[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = (double)(loc[slot * 3 + k] * max_loc);
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

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(64) void metrics_kernel(const MetricsP p) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)p.clips * p.blocks;
    long long cnt[NUM_COUNTERS];
#pragma unroll
    for (int i = 0; i < NUM_COUNTERS; ++i) cnt[i] = 0;
    double total_de = 0.0;

    if (gid < total) {
        const int clip = (int)(gid / p.blocks), blk = (int)(gid % p.blocks);
        const int n = p.classes * p.overlaps;
        const int f0 = blk * p.fpb;
        const int nf = min(p.fpb, p.frames - f0);
        const float* sed = p.sed + ((size_t)clip * p.frames + f0) * n;
        const float* doa = p.doa + ((size_t)clip * p.frames + f0) * 3 * n;
        const float* tgt = p.target + ((size_t)clip * p.frames + f0) * 4 * n;
        unsigned long long mp[MAX_BLOCK_FRAMES], mt[MAX_BLOCK_FRAMES];
        for (int f = 0; f < MAX_BLOCK_FRAMES; ++f) {
            mp[f] = f < nf ? activity_mask(sed + (size_t)f * n, n) : 0ull;
            mt[f] = f < nf ? activity_mask(tgt + (size_t)f * 4 * n, n) : 0ull;
        }
        const unsigned long long cls_mask = (1ull << p.overlaps) - 1ull;

        // ---- location_sensitive_detection, frame by frame ----
        for (int f = 0; f < nf; ++f) {
            const int n_p = __popcll(mp[f]), n_t = __popcll(mt[f]);
            if (n_t == 0) {
                cnt[1] += 2 * n_p;
            } else if (n_p == 0) {
                cnt[2] += 2 * n_t;
            } else {
                const float* lp = doa + (size_t)f * 3 * n;
                const float* lt = tgt + (size_t)f * 4 * n + n;
                int matched = 0;
                for (int j = 0; j < n; ++j) {
                    if (!((mt[f] >> j) & 1ull)) continue;
                    const int c = j / p.overlaps;
                    double t[3];
                    load_xyz(lt, j, p.max_loc, t);
                    bool match = false;
                    for (int e = 0; e < p.overlaps; ++e) {
                        const int k = c * p.overlaps + e;
                        if (!((mp[f] >> k) & 1ull)) continue;
                        double q[3];
                        load_xyz(lp, k, p.max_loc, q);
                        const double dx = t[0] - q[0], dy = t[1] - q[1], dz = t[2] - q[2];
                        if (sqrt(dx * dx + dy * dy + dz * dz) < p.spatial_threshold) match = true;
                    }
                    matched += match ? 1 : 0;
                }
                cnt[0] += matched;
                cnt[2] += n_t - matched;
                cnt[1] += n_p - matched;
            }
        }

        // ---- DCASE21 segment metrics for this block ----
        int loc_fn = 0, loc_fp = 0;
        for (int c = 0; c < p.classes; ++c) {
            const int sh = c * p.overlaps;
            int nb_gt = 0, nb_pred = 0;
            for (int f = 0; f < nf; ++f) {
                nb_gt = max(nb_gt, __popcll((mt[f] >> sh) & cls_mask));
                nb_pred = max(nb_pred, __popcll((mp[f] >> sh) & cls_mask));
            }
            cnt[9] += nb_gt;
            if (nb_gt && nb_pred) {
                double tsum[3] = {0.0, 0.0, 0.0};
                int tn[3] = {0, 0, 0};
                int order[3] = {-1, -1, -1}, n_tracks = 0;      // tracks in order of first appearance (the dict's order)
                for (int f = 0; f < nf; ++f) {
                    const unsigned g_bits = (unsigned)((mt[f] >> sh) & cls_mask), p_bits = (unsigned)((mp[f] >> sh) & cls_mask);
                    const int g = __popc(g_bits), q = __popc(p_bits);
                    if (!g || !q) continue;
                    double cost[3][3];
                    {
                        const float* lt = tgt + (size_t)f * 4 * n + n;
                        const float* lp = doa + (size_t)f * 3 * n;
                        int r = 0;
                        for (int e = 0; e < p.overlaps; ++e) {
                            if (!((g_bits >> e) & 1u)) continue;
                            double a[3];
                            load_xyz(lt, sh + e, p.max_loc, a);
                            int col = 0;
                            for (int e2 = 0; e2 < p.overlaps; ++e2) {
                                if (!((p_bits >> e2) & 1u)) continue;
                                double b[3];
                                load_xyz(lp, sh + e2, p.max_loc, b);
                                cost[r][col++] = angular_distance_deg(a, b);
                            }
                            ++r;
                        }
                    }
                    // assignment by enumeration over the permutations of max(g, q) elements
                    const int m = max(g, q);
                    const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
                    int best = -1;
                    double best_cost = 0.0;
                    for (int k = 0; k < 6; ++k) {
                        bool ok = true;
                        for (int i = m; i < 3; ++i) ok = ok && perms[k][i] == i;
                        if (!ok) continue;
                        double tot = 0.0;
                        for (int r = 0; r < g; ++r)
                            if (perms[k][r] < q) tot += cost[r][perms[k][r]];
                        if (best < 0 || tot < best_cost) {
                            best = k;
                            best_cost = tot;
                        }
                    }
                    for (int r = 0; r < g; ++r) {
                        const int col = perms[best][r];
                        if (col >= q) continue;
                        if (tn[r] == 0) order[n_tracks++] = r;
                        tsum[r] += cost[r][col];
                        tn[r] += 1;
                    }
                }
                if (n_tracks == 0) {
                    loc_fn += nb_pred;
                    cnt[5] += nb_pred;
                    cnt[12] += nb_pred;
                } else {
                    for (int i = 0; i < n_tracks; ++i) {
                        const int r = order[i];
                        const double avg = tsum[r] / (double)tn[r];
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
        cnt[6] += min(loc_fp, loc_fn);
        cnt[7] += max(0, loc_fn - loc_fp);
        cnt[8] += max(0, loc_fp - loc_fn);
    }

#pragma unroll
    for (int i = 0; i < NUM_COUNTERS; ++i) {
        const long long s = wave_sum_i64(cnt[i]);
        if ((threadIdx.x & 63) == 0 && s != 0) atomicAdd(reinterpret_cast<unsigned long long*>(p.counters + i), (unsigned long long)s);
    }
    const double de = wave_sum_d(total_de);
    if ((threadIdx.x & 63) == 0 && de != 0.0) atomicAdd(p.total_de, de);
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
    if (overlaps > 3 || classes * overlaps > 64 || frames_per_block > MAX_BLOCK_FRAMES) return SELD_EUNSUPPORTED;
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
    hipLaunchKernelGGL(metrics_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream, p);
    return check_launch();
}
