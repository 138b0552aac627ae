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
