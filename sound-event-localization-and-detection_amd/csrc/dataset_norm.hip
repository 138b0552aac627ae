// Dataset normalisation on the device (reference train.py:242-408).
//
// The reference normalises the three predictor arrays (N, C, F, T) on the host before training:
//   * dual-quaternion unit norm (train.py:257-275, repeated for validation / test at 277-308):
//       per (n, f, t) position, channels 0..7 = (q0..q3, p0..p3):
//         den0 = q0^2 + q1^2 + q2^2 + q3^2 ; den1 = sqrt(den0) ; cross = q0 p0 + q1 p1 + q2 p2 + q3 p3
//         p_i <- p_i - cross / den0 * q_i ;  q_i <- q_i / den1
//   * mean / std per channel group (train.py:341-405): g = x[:, c0:c1]; g <- (g - mean(g)) / std(g),
//     one scalar mean and one population std over the whole group.
//
// Both are HBM-bound streaming passes: 64 B per position for the unit norm (8 planes read + written),
// 4 + 8 B per element for the standardisation (one moments pass, one apply pass).  Every access
// is a 16-byte load/store along the contiguous (f, t) axis; one item's group is one contiguous segment.
//
// The unit norm evaluates the reference's expression tree operation by operation (no FMA contraction,
// correctly rounded divide and square root): bit for bit the IEEE 754 value of that tree, NaN for a zero q
// included.  (torch's CPU sqrt is 1 ulp off for ~0.6 % of inputs, so the reference's own q channels are
// reproduced to 1 ulp, its p channels exactly.)  The moments are accumulated in double (numpy: float32 pairwise sums) and rounded to float32
// once, then applied with the reference's two float32 operations (subtract, divide).
#include "common.h"

namespace seld {

template <int V>
struct Vec;
template <>
struct Vec<4> {
    typedef float4 T;
};
template <>
struct Vec<1> {
    typedef float T;
};

// Correctly rounded float32 square root: v_sqrt_f32 is good to 1 ulp, the two residual tests pick the neighbour when
// it is the nearer one (inputs below 2^-96 are scaled out of the denormal range first).
__device__ __forceinline__ float sqrt_rn(float x) {
    const bool scale = x < 0x1p-96f;
    const float xs = scale ? x * 0x1p+32f : x;
    float s = __builtin_amdgcn_sqrtf(xs);
    const float sd = __int_as_float(__float_as_int(s) - 1);
    const float su = __int_as_float(__float_as_int(s) + 1);
    const float vp = __builtin_fmaf(-sd, s, xs);
    const float vs = __builtin_fmaf(-su, s, xs);
    s = vp <= 0.f ? sd : s;
    s = vs > 0.f ? su : s;
    s = scale ? s * 0x1p-16f : s;
    return (xs == 0.f || __builtin_isinf(xs)) ? xs : s;
}

__device__ __forceinline__ void dq_unit_norm_1(float q[4], float p[4]) {
#pragma clang fp contract(off)
    const float den0 = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    const float den1 = sqrt_rn(den0);
    const float cross = ((q[0] * p[0] + q[1] * p[1]) + q[2] * p[2]) + q[3] * p[3];
    const float ratio = __fdiv_rn(cross, den0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[i] = p[i] - ratio * q[i];
        q[i] = __fdiv_rn(q[i], den1);
    }
}

// grid.x walks the positions of one item (V floats per thread), grid.y walks items
template <int V>
__global__ __launch_bounds__(256) void dq_unit_norm_kernel(float* __restrict__ x, long long items, long long item_stride,
                                                           long long hw) {
    typedef typename Vec<V>::T VT;
    const long long nv = hw / V;
    for (long long n = blockIdx.y; n < items; n += gridDim.y) {
        float* base = x + n * item_stride;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
            VT v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = reinterpret_cast<const VT*>(base + c * hw)[i];
            float* f = reinterpret_cast<float*>(v);      // f[c * V + e]
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float q[4], p[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    q[c] = f[c * V + e];
                    p[c] = f[(c + 4) * V + e];
                }
                dq_unit_norm_1(q, p);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f[c * V + e] = q[c];
                    f[(c + 4) * V + e] = p[c];
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) reinterpret_cast<VT*>(base + c * hw)[i] = v[c];
        }
    }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One read of the group gives both moments: acc[0] += sum(x - K), acc[1] += sum((x - K)^2) in double, K = the group's
// first element (a pilot shift, so the variance below does not cancel when |mean| >> std).
template <int V>
__global__ __launch_bounds__(256) void group_moment_kernel(const float* __restrict__ x, long long items, long long item_stride,
                                                           long long seg, double* __restrict__ acc) {
    typedef typename Vec<V>::T VT;
    __shared__ double part[2][4];
    const long long nv = seg / V;
    const double pilot = (double)x[0];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) acc[2] = pilot;     // x[0] is overwritten by the apply pass
    double s1 = 0.0, s2 = 0.0;
    for (long long n = blockIdx.y; n < items; n += gridDim.y) {
        const VT* base = reinterpret_cast<const VT*>(x + n * item_stride);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
            VT v = base[i];
            const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const double d = (double)f[e] - pilot;
                s1 += d;
                s2 = fma(d, d, s2);
            }
        }
    }
    s1 = wave_sum_f64(s1);
    s2 = wave_sum_f64(s2);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[0][wave] = s1;
        part[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 2) atomicAdd(acc + threadIdx.x, (part[threadIdx.x][0] + part[threadIdx.x][1]) +
                                                         (part[threadIdx.x][2] + part[threadIdx.x][3]));
}

// mean = K + S1/n, var = S2/n - (S1/n)^2, both rounded to float32 once (numpy returns float32 scalars), then the
// reference's two float32 operations: subtract, divide
template <int V>
__global__ __launch_bounds__(256) void group_apply_kernel(float* __restrict__ x, long long items, long long item_stride, long long seg,
                                                          double count, const double* __restrict__ acc,
                                                          float* __restrict__ mean_std) {
    typedef typename Vec<V>::T VT;
    const long long nv = seg / V;
    const double m1 = acc[0] / count;
    const float mean = (float)(acc[2] + m1);
    const float sd = (float)sqrt(fmax(acc[1] / count - m1 * m1, 0.0));
    if (mean_std && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        mean_std[0] = mean;
        mean_std[1] = sd;
    }
    for (long long n = blockIdx.y; n < items; n += gridDim.y) {
        VT* base = reinterpret_cast<VT*>(x + n * item_stride);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
            VT v = base[i];
            float* f = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int e = 0; e < V; ++e) f[e] = __fdiv_rn(f[e] - mean, sd);
            base[i] = v;
        }
    }
}

// about 2048 workgroups in all: x over one item's vectors (`per_thread` each), y over items
static dim3 stream_grid(long long nv, long long items, int per_thread) {
    long long gx = (nv + 256LL * per_thread - 1) / (256LL * per_thread);
    if (gx < 1) gx = 1;
    if (gx > 2048) gx = 2048;
    long long gy = 2048 / gx;
    if (gy < 1) gy = 1;
    if (gy > items) gy = items;
    return dim3((unsigned)gx, (unsigned)gy);
}

static bool vec4_ok(const void* x, long long a, long long b) {
    return (reinterpret_cast<uintptr_t>(x) & 15) == 0 && a % 4 == 0 && b % 4 == 0;
}

}  // namespace seld

using namespace seld;

extern "C" int seld_dq_unit_norm(float* x, int64_t items, int32_t channels, int64_t hw, void* stream) {
    if (items < 0 || channels < 8 || hw < 0) return SELD_EINVAL;
    if (items == 0 || hw == 0) return SELD_OK;          // an empty array has no storage: x may be NULL
    if (!x) return SELD_EINVAL;
    const long long stride = (long long)channels * hw;
    hipStream_t st = (hipStream_t)stream;
    if (vec4_ok(x, hw, hw)) {
        hipLaunchKernelGGL(dq_unit_norm_kernel<4>, stream_grid(hw / 4, items, 1), dim3(256), 0, st, x, (long long)items, stride,
                           (long long)hw);
    } else {
        hipLaunchKernelGGL(dq_unit_norm_kernel<1>, stream_grid(hw, items, 1), dim3(256), 0, st, x, (long long)items, stride,
                           (long long)hw);
    }
    return check_launch();
}

extern "C" int seld_group_standardize(float* x, int64_t items, int32_t channels, int32_t c0, int32_t c1, int64_t hw, double* work,
                                      float* mean_std, void* stream) {
    if (!x || !work || items < 0 || channels <= 0 || c0 < 0 || c1 > channels || c0 > c1 || hw < 0) return SELD_EINVAL;
    if (items == 0 || hw == 0 || c0 == c1) return SELD_EINVAL;      // numpy: mean of an empty slice is nan + a warning
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(work, 0, 3 * sizeof(double), st) != hipSuccess) return SELD_ELAUNCH;
    const long long stride = (long long)channels * hw;
    const long long seg = (long long)(c1 - c0) * hw;
    const double count = (double)items * (double)seg;
    float* g = x + (long long)c0 * hw;
    if (vec4_ok(g, hw, stride)) {
        dim3 grid = stream_grid(seg / 4, items, 4);
        hipLaunchKernelGGL(group_moment_kernel<4>, grid, dim3(256), 0, st, g, (long long)items, stride, seg, work);
        hipLaunchKernelGGL(group_apply_kernel<4>, grid, dim3(256), 0, st, g, (long long)items, stride, seg, count, work, mean_std);
    } else {
        dim3 grid = stream_grid(seg, items, 4);
        hipLaunchKernelGGL(group_moment_kernel<1>, grid, dim3(256), 0, st, g, (long long)items, stride, seg, work);
        hipLaunchKernelGGL(group_apply_kernel<1>, grid, dim3(256), 0, st, g, (long long)items, stride, seg, count, work, mean_std);
    }
    return check_launch();
}
