// STFT magnitude / phase feature extractor for gfx950 (utility_functions.py:129-155).
//
// Restates scipy.signal.stft(x, window='hamming', nperseg, noverlap) with its defaults
// (boundary='zeros', padded=True, detrend=False, onesided, scaling='spectrum'):
//   periodic Hamming window w[n] = 0.54 - 0.46 cos(2 pi n / N); the signal is extended by N/2
//   zeros on both sides and zero-padded at the end to a whole number of hops; frame m starts at
//   m*hop - N/2; Z = rfft(frame * w) / sum(w).  The reference then keeps |Z| (and angle(Z) stacked
//   on the channel axis), drops the DC bin and drops the last frame.
//
// One workgroup transforms FT consecutive frames of one channel with a radix-2 Stockham FFT in
// LDS (N/2 butterflies per pass, twiddles from an LDS table built with sincospif), collects
// |Z| / angle(Z) in an LDS tile [bin][frame] and writes it with the frame index fastest, which is
// the output's contiguous axis.
#include "common.h"
#include "env.h"

namespace seld {

constexpr int FT = 16;

// bin0: first bin kept (1 = DC dropped, the reference's cut_dc=True; 0 = all N/2 + 1 bins), nbins = N/2 + 1 - bin0.
// window_g (nullable): N window values already divided by their sum; null = periodic Hamming.
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ x, int C, int L, int N, int logN, int hop,
                                                   int frames_out, int output_phase, int bin0,
                                                   const float* __restrict__ window_g, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int half = N >> 1;
    float2* buf0 = reinterpret_cast<float2*>(smem);                 // N
    float2* buf1 = buf0 + N;                                        // N
    float2* tw = buf1 + N;                                          // N/2
    float* win = reinterpret_cast<float*>(tw + half);               // N
    const int nbins = half + 1 - bin0;
    float* mag = win + N;                                           // (half + 1) * (FT + 1)
    float* pha = mag + (half + 1) * (FT + 1);                       // (half + 1) * (FT + 1)

    const int tid = threadIdx.x;
    const int c = blockIdx.y;
    const int m0 = blockIdx.x * FT;
    const float inv_wsum = 1.0f / (0.54f * (float)N);

    for (int j = tid; j < half; j += blockDim.x) {
        float sn, cs;
        sincospif(-2.0f * (float)j / (float)N, &sn, &cs);
        tw[j] = make_float2(cs, sn);
    }
    for (int n = tid; n < N; n += blockDim.x)
        win[n] = window_g ? window_g[n] : (0.54f - 0.46f * cospif(2.0f * (float)n / (float)N)) * inv_wsum;
    __syncthreads();

    const float* xc = x + (size_t)c * L;
    for (int f = 0; f < FT; ++f) {
        const int m = m0 + f;
        if (m >= frames_out) break;          // uniform across the workgroup
        const long long start = (long long)m * hop - half;
        for (int n = tid; n < N; n += blockDim.x) {
            const long long s = start + n;
            const float v = (s >= 0 && s < L) ? xc[s] * win[n] : 0.f;
            buf0[n] = make_float2(v, 0.f);
        }
        __syncthreads();
        float2* src = buf0;
        float2* dst = buf1;
        for (int pass = 0; pass < logN; ++pass) {
            const int p = 1 << pass;
            for (int i = tid; i < half; i += blockDim.x) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 1) + k;
                const float2 w = tw[k * (half >> pass)];
                const float2 u0 = src[i];
                const float2 a = src[i + half];
                const float2 u1 = make_float2(w.x * a.x - w.y * a.y, w.x * a.y + w.y * a.x);
                dst[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
                dst[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
            }
            __syncthreads();
            float2* t = src; src = dst; dst = t;
        }
        // bins bin0 .. N/2
        for (int b = tid; b < nbins; b += blockDim.x) {
            const float2 z = src[b + bin0];
            mag[b * (FT + 1) + f] = sqrtf(z.x * z.x + z.y * z.y);
            if (output_phase) pha[b * (FT + 1) + f] = atan2f(z.y, z.x);
        }
        __syncthreads();
    }
    const int nf = (frames_out - m0) < FT ? (frames_out - m0) : FT;
    for (int e = tid; e < nbins * FT; e += blockDim.x) {
        const int b = e / FT, f = e - b * FT;
        if (f < nf) {
            out[((size_t)c * nbins + b) * frames_out + m0 + f] = mag[b * (FT + 1) + f];
            if (output_phase) out[((size_t)(C + c) * nbins + b) * frames_out + m0 + f] = pha[b * (FT + 1) + f];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// N = 512 (the reference's segment length, utility_functions.py:129): one WAVE transforms TWO consecutive frames as the
// real and imaginary part of one complex signal with three radix-8 Stockham passes -- 8 points per lane in registers,
// wave-private LDS exchanges (skewed by one float per 8: conflict-free), no workgroup barrier inside the transform --
// and separates them afterwards (A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / 2i).  The radix-2 kernel
// above spends 11 workgroup barriers per frame on a complex transform of real input: 248 us for a 60 s x 8 channel clip
// (7 % of HBM); this one is bound by its sqrt / atan2 and the tile transpose.
// ---------------------------------------------------------------------------------------------------------------------
struct Cx { float re, im; };
__device__ __forceinline__ Cx cadd(Cx a, Cx b) { return Cx{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cx csub(Cx a, Cx b) { return Cx{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Cx cmul(Cx a, Cx b) { return Cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cx cmul_mi(Cx a) { return Cx{a.im, -a.re}; }          // a * (-i)

// forward 8-point DFT in place, outputs in natural order
__device__ __forceinline__ void dft8(Cx (&v)[8]) {
    constexpr float H = 0.70710678118654752f;
    Cx a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    Cx a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    Cx a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    Cx a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    a5 = Cx{H * (a5.re + a5.im), H * (a5.im - a5.re)};              // * (1 - i) / sqrt 2
    a6 = cmul_mi(a6);
    a7 = Cx{H * (a7.im - a7.re), -H * (a7.re + a7.im)};             // * (-1 - i) / sqrt 2
    // two 4-point transforms: evens from a0..a3, odds from a4..a7
    {
        const Cx c0 = cadd(a0, a2), c2 = csub(a0, a2), c1 = cadd(a1, a3), c3 = cmul_mi(csub(a1, a3));
        v[0] = cadd(c0, c1); v[4] = csub(c0, c1); v[2] = cadd(c2, c3); v[6] = csub(c2, c3);
    }
    {
        const Cx c0 = cadd(a4, a6), c2 = csub(a4, a6), c1 = cadd(a5, a7), c3 = cmul_mi(csub(a5, a7));
        v[1] = cadd(c0, c1); v[5] = csub(c0, c1); v[3] = cadd(c2, c3); v[7] = csub(c2, c3);
    }
}

constexpr int FT512 = 16;                 // frames per workgroup
constexpr int SK512 = 512 + 64;           // one skewed 512-float array: element i lives at i + (i >> 3)
__device__ __forceinline__ int sk(int i) { return i + (i >> 3); }

__global__ __launch_bounds__(256) void stft512_kernel(const float* __restrict__ x, int C, int L, int hop, int frames_out,
                                                      int output_phase, int bin0, const float* __restrict__ window_g,
                                                      float* __restrict__ out) {
    constexpr int N = 512, HALF = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* win = smem;                                   // N
    float* twr = win + N;                                // N: cos(2 pi k / N)
    float* twi = twr + N;                                // N: -sin(2 pi k / N)
    float* scr = twi + N;                                // 4 waves x (re[SK512] | im[SK512])
    float* mag = scr + 4 * 2 * SK512;                    // (HALF + 1) x (FT512 + 1)
    float* pha = mag + (HALF + 1) * (FT512 + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.y;
    const int m0 = blockIdx.x * FT512;
    const int nbins = HALF + 1 - bin0;
    const float inv_wsum = 1.0f / (0.54f * (float)N);
    for (int k = tid; k < N; k += 256) {
        float sn, cs;
        sincospif(-2.0f * (float)k / (float)N, &sn, &cs);
        twr[k] = cs;
        twi[k] = sn;
        win[k] = window_g ? window_g[k] : (0.54f - 0.46f * cospif(2.0f * (float)k / (float)N)) * inv_wsum;
    }
    __syncthreads();
    const float* xc = x + (size_t)c * L;
    float* re = scr + wave * 2 * SK512;
    float* im = re + SK512;
    for (int pr = wave; pr < FT512 / 2; pr += 4) {
        const int m = m0 + 2 * pr;
        if (m >= frames_out) break;                      // wave-uniform
        const bool has_b = m + 1 < frames_out;
        const long long sa = (long long)m * hop - HALF, sb = sa + hop;
        Cx v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = lane + 64 * r;
            const long long ia = sa + n, ib = sb + n;
            const float w = win[n];
            v[r].re = (ia >= 0 && ia < L) ? xc[ia] * w : 0.f;
            v[r].im = (has_b && ib >= 0 && ib < L) ? xc[ib] * w : 0.f;
        }
        // pass 1 (sub-transform length 1: no twiddles): out[8 j + r]
        dft8(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) { re[sk(8 * lane + r)] = v[r].re; im[sk(8 * lane + r)] = v[r].im; }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // pass 2 (length 8): in[j + 64 r] * W_64^(r k), k = j % 8; out[(j / 8) * 64 + k + 8 r]
#pragma unroll
        for (int r = 0; r < 8; ++r) { v[r].re = re[sk(lane + 64 * r)]; v[r].im = im[sk(lane + 64 * r)]; }
        {
            const int k = lane & 7;
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], Cx{twr[r * k * 8], twi[r * k * 8]});
        }
        dft8(v);
        __builtin_amdgcn_wave_barrier();
        {
            const int j0 = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
            for (int r = 0; r < 8; ++r) { re[sk(j0 + 8 * r)] = v[r].re; im[sk(j0 + 8 * r)] = v[r].im; }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // pass 3 (length 64): in[j + 64 r] * W_512^(r j); out[j + 64 r] = Z in natural order
#pragma unroll
        for (int r = 0; r < 8; ++r) { v[r].re = re[sk(lane + 64 * r)]; v[r].im = im[sk(lane + 64 * r)]; }
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], Cx{twr[r * lane], twi[r * lane]});
        dft8(v);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) { re[sk(lane + 64 * r)] = v[r].re; im[sk(lane + 64 * r)] = v[r].im; }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // the two real transforms, bins bin0 .. N/2, into the [bin][frame] tile
        for (int b = lane; b < nbins; b += 64) {
            const int k = b + bin0, km = (N - k) & (N - 1);
            const float zr = re[sk(k)], zi = im[sk(k)], yr = re[sk(km)], yi = im[sk(km)];
            const float ar = 0.5f * (zr + yr), ai = 0.5f * (zi - yi);
            const float br = 0.5f * (zi + yi), bi = -0.5f * (zr - yr);
            const int f = 2 * pr;
            mag[b * (FT512 + 1) + f] = sqrtf(ar * ar + ai * ai);
            mag[b * (FT512 + 1) + f + 1] = sqrtf(br * br + bi * bi);
            if (output_phase) {
                pha[b * (FT512 + 1) + f] = atan2f(ai, ar);
                pha[b * (FT512 + 1) + f + 1] = atan2f(bi, br);
            }
        }
        __builtin_amdgcn_wave_barrier();                 // the scratch is rewritten by the wave's next pair
    }
    __syncthreads();
    const int nf = (frames_out - m0) < FT512 ? (frames_out - m0) : FT512;
    for (int e = tid; e < nbins * FT512; e += 256) {
        const int b = e / FT512, f = e - b * FT512;
        if (f < nf) {
            out[((size_t)c * nbins + b) * frames_out + m0 + f] = mag[b * (FT512 + 1) + f];
            if (output_phase) out[((size_t)(C + c) * nbins + b) * frames_out + m0 + f] = pha[b * (FT512 + 1) + f];
        }
    }
}

static int frames_total(int L, int N, int noverlap) {
    const int hop = N - noverlap;
    if (hop <= 0) return -1;
    long long Lp = (long long)L + N;                  // boundary='zeros' extension by N/2 on both sides
    long long nadd = ((-(Lp - N)) % hop + hop) % hop; // padded=True
    nadd %= N;
    long long frames = (Lp + nadd - N) / hop + 1;
    return (int)frames;
}

}  // namespace seld
using namespace seld;

extern "C" int seld_stft_frames_ex(int32_t L, int32_t nperseg, int32_t noverlap, int32_t cut_last_timeframe) {
    if (L <= 0 || nperseg <= 1 || noverlap < 0 || noverlap >= nperseg) return SELD_EINVAL;
    return frames_total(L, nperseg, noverlap) - (cut_last_timeframe ? 1 : 0);
}
extern "C" int seld_stft_frames(int32_t L, int32_t nperseg, int32_t noverlap) {
    return seld_stft_frames_ex(L, nperseg, noverlap, 1);
}

extern "C" int seld_stft_magphase_ex(const float* x, int32_t C, int32_t L, int32_t nperseg, int32_t noverlap,
                                     int32_t output_phase, int32_t cut_dc, int32_t cut_last_timeframe,
                                     const float* window, float* out, void* stream) {
    if (!x || !out || C <= 0 || L <= 0 || nperseg <= 1 || noverlap < 0 || noverlap >= nperseg) return SELD_EINVAL;
    int logN = 0;
    while ((1 << logN) < nperseg) ++logN;
    if ((1 << logN) != nperseg || nperseg > 4096 || nperseg < 4) return SELD_EUNSUPPORTED;   // power-of-two segments only
    const int frames = frames_total(L, nperseg, noverlap) - (cut_last_timeframe ? 1 : 0);
    if (frames <= 0) return SELD_EINVAL;
    const int half = nperseg / 2;
    if (nperseg == 512 && !env().stft_radix2) {
        const size_t smem512 = sizeof(float) * ((size_t)3 * 512 + (size_t)4 * 2 * SK512 + (size_t)2 * 257 * (FT512 + 1));
        hipLaunchKernelGGL(stft512_kernel, dim3((frames + FT512 - 1) / FT512, C), dim3(256), smem512, (hipStream_t)stream, x, C, L,
                           nperseg - noverlap, frames, output_phase, cut_dc ? 1 : 0, window, out);
        return check_launch();
    }
    const size_t smem = sizeof(float) * ((size_t)2 * nperseg * 2 + (size_t)half * 2 + nperseg + (size_t)2 * (half + 1) * (FT + 1));
    dim3 grid((frames + FT - 1) / FT, C);
    hipLaunchKernelGGL(stft_kernel, grid, dim3(256), smem, (hipStream_t)stream, x, C, L, nperseg, logN,
                       nperseg - noverlap, frames, output_phase, cut_dc ? 1 : 0, window, out);
    return check_launch();
}
extern "C" int seld_stft_magphase(const float* x, int32_t C, int32_t L, int32_t nperseg, int32_t noverlap,
                                  int32_t output_phase, float* out, void* stream) {
    return seld_stft_magphase_ex(x, C, L, nperseg, noverlap, output_phase, 1, 1, nullptr, out, stream);
}
