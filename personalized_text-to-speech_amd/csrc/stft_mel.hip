// Magnitude -> mel projection -> log-clamp of the STFT (reference mel_processing.py:63-69 `sqrt(re^2 + im^2 + 1e-6)`, :73-82 / :105-111
// `log(clamp(mel_basis @ spec, 1e-5))`) as ONE kernel per direction, behind the DFT product of vits_conv1d_cl.
//
//   ri    [rows = b * frames][ld] fp32: real parts in columns [0, F), imaginary parts in [Fp, Fp + F)   (the DFT operand's layout)
//   basis [M][F] fp32 (librosa mel filter bank)
//   mel, lin [b][M][frames] fp32 (the reference's layout); lin = basis @ mag is kept for the backward
// forward : workgroup = one frame: magnitudes to LDS, wave w sums the filters m = w, w + 4, ..: lanes stride over the frequencies,
//           fixed-order butterfly (bitwise reproducible);
// backward: d lin = d mel / lin where lin >= clip (torch's clamp_min backward), d mag[f] = sum_m basis[m][f] d lin[m] (thread = f,
//           coalesced over f), d re = d mag * re / mag, d im = d mag * im / mag; the padding columns of d ri are written as zero.
#include "common.h"

namespace {

constexpr int FMAX = 2048, MMAX = 256;

__global__ __launch_bounds__(256) void stft_mel_fwd_kernel(const float* __restrict__ ri, const float* __restrict__ basis, float* __restrict__ mel,
                                                           float* __restrict__ lin, int frames, int F, int Fp, int ld, int M, float clip) {
  __shared__ float mag[FMAX];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* r = ri + (size_t)row * ld;
  for (int f = tid; f < F; f += 256) { const float re = r[f], im = r[Fp + f]; mag[f] = sqrtf(re * re + im * im + 1e-6f); }
  __syncthreads();
  const int bi = row / frames, fr = row - bi * frames;
  for (int m = wave; m < M; m += 4) {
    const float* bm = basis + (size_t)m * F;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) s = fmaf(bm[f], mag[f], s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
      const size_t o = ((size_t)bi * M + m) * frames + fr;
      if (lin) lin[o] = s;
      mel[o] = logf(s > clip ? s : clip);
    }
  }
}

__global__ __launch_bounds__(256) void stft_mel_bwd_kernel(const float* __restrict__ ri, const float* __restrict__ basis, const float* __restrict__ lin,
                                                           const float* __restrict__ dmel, float* __restrict__ dri, int frames, int F, int Fp, int ld,
                                                           int M, float clip) {
  __shared__ float dl[MMAX];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int bi = row / frames, fr = row - bi * frames;
  for (int m = tid; m < M; m += 256) {
    const size_t o = ((size_t)bi * M + m) * frames + fr;
    const float l = lin[o];
    dl[m] = l >= clip ? dmel[o] / l : 0.f;
  }
  __syncthreads();
  const float* r = ri + (size_t)row * ld;
  float* d = dri + (size_t)row * ld;
  for (int c = tid; c < ld; c += 256) {
    const int f = c < Fp ? c : c - Fp;
    if (f >= F || c >= 2 * Fp) { d[c] = 0.f; continue; }
    float s = 0.f;
    for (int m = 0; m < M; ++m) s = fmaf(basis[(size_t)m * F + f], dl[m], s);
    const float re = r[f], im = r[Fp + f];
    const float g = s / sqrtf(re * re + im * im + 1e-6f);
    d[c] = g * (c < Fp ? re : im);
  }
}

}  // namespace

extern "C" int vits_stft_mel_fwd(const float* ri, const float* basis, float* mel, float* lin, int rows, int frames, int F, int Fp, int ld,
                                 int M, float clip, void* stream) {
  if (!ri || !basis || !mel || rows <= 0 || frames <= 0 || F <= 0 || M <= 0 || Fp < F || ld < 2 * Fp || rows % frames != 0) return VITS_E_BADARG;
  if (F > FMAX || M > MMAX) return VITS_E_UNSUPPORTED;
  hipLaunchKernelGGL(stft_mel_fwd_kernel, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream), ri, basis, mel, lin, frames, F, Fp, ld, M, clip);
  return vits::check_launch("vits_stft_mel_fwd");
}

extern "C" int vits_stft_mel_bwd(const float* ri, const float* basis, const float* lin, const float* dmel, float* dri, int rows, int frames,
                                 int F, int Fp, int ld, int M, float clip, void* stream) {
  if (!ri || !basis || !lin || !dmel || !dri || rows <= 0 || frames <= 0 || F <= 0 || M <= 0 || Fp < F || ld < 2 * Fp || rows % frames != 0)
    return VITS_E_BADARG;
  if (F > FMAX || M > MMAX) return VITS_E_UNSUPPORTED;
  hipLaunchKernelGGL(stft_mel_bwd_kernel, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream), ri, basis, lin, dmel, dri, frames, F, Fp, ld, M, clip);
  return vits::check_launch("vits_stft_mel_bwd");
}
