// Atari frame preprocessing on the device: what the reference's make_env stacks on the emulator --
// deep_rl/component/envs.py:39-47 -> baselines.common.atari_wrappers (third party, absent from /root/reference):
//   MaxAndSkipEnv    observation = elementwise max of the last two raw RGB frames of a 4-frame skip
//   WarpFrame        cv2.cvtColor(RGB2GRAY), then cv2.resize(.., (84, 84), interpolation=cv2.INTER_AREA)
// restated from OpenCV's published algorithms: the 8-bit luminance is the fixed-point
//   Y = (4899 R + 9617 G + 1868 B + 8192) >> 14                       (color_rgb: R2Y, G2Y, B2Y, yuv_shift = 14)
// and INTER_AREA at a non-integer scale is the separable area average of resizeArea_: a table of (source index, weight)
// per destination index (computeResizeAreaTab: the partially covered first / last source cell weighted by its overlap,
// everything divided by the cell width), per source row  buf[dx] = sum_x S[sx] * alpha  in table order (fp32, from 0),
// per destination row  sum = beta * buf  for its first source row and  sum += beta * buf  for the others, and finally
// saturate_cast<uchar>(sum) (round half to even).  Same operation order here, -ffp-contract=off: bit-exact against
// oracle/preproc_oracle.py (PARITY UNPINNED BY THE REFERENCE: neither cv2 nor baselines is in the image).
// One thread per output pixel: 2 x <= 4 x <= 3 RGB source pixels; 201 600 B read and 7 056 B written per environment.
#include "common.h"
#include <math.h>

// Host: OpenCV's computeResizeAreaTab for one axis (cn = 1).  si / alpha: entries in order; offs[d] .. offs[d + 1]: the
// entries of destination index d.  Returns the entry count (<= 2 * ssize + dsize), or a negative error.
DRA_API int dra_resize_area_tab(int ssize, int dsize, int* si, float* alpha, int* offs, int max_entries) {
  if (ssize < 1 || dsize < 1 || !si || !alpha || !offs || max_entries < 1) return DRA_EINVAL;
  const double scale = (double)ssize / (double)dsize;
  int k = 0;
  for (int dx = 0; dx < dsize; ++dx) {
    offs[dx] = k;
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cell = fmin(scale, (double)ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
    sx1 = sx1 < sx2 ? sx1 : sx2;
    if (k + (sx2 - sx1) + 2 > max_entries) return DRA_EINVAL;
    if (sx1 - fsx1 > 1e-3) { si[k] = sx1 - 1; alpha[k++] = (float)((sx1 - fsx1) / cell); }
    for (int sx = sx1; sx < sx2; ++sx) { si[k] = sx; alpha[k++] = (float)(1.0 / cell); }
    if (fsx2 - sx2 > 1e-3) { si[k] = sx2; alpha[k++] = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell); }
  }
  offs[dsize] = k;
  return k;
}

__global__ void __launch_bounds__(256)
atari_preprocess_kernel(const uint8_t* __restrict__ raw, int n_env, int H, int W, int OH, int OW, const int* __restrict__ x_si,
                        const float* __restrict__ x_alpha, const int* __restrict__ x_off, const int* __restrict__ y_si,
                        const float* __restrict__ y_alpha, const int* __restrict__ y_off, uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_env * OH * OW) return;
  const int dx = (int)(i % OW), dy = (int)((i / OW) % OH);
  const int64_t env = i / ((int64_t)OW * OH);
  const uint8_t* f0 = raw + env * 2 * (int64_t)H * W * 3;     // the two newest raw frames of this environment
  const uint8_t* f1 = f0 + (int64_t)H * W * 3;
  const int xa = x_off[dx], xb = x_off[dx + 1], ya = y_off[dy], yb = y_off[dy + 1];
  float sum = 0.f;
  for (int yk = ya; yk < yb; ++yk) {
    const int sy = y_si[yk];
    float buf = 0.f;
    for (int xk = xa; xk < xb; ++xk) {
      const int64_t o = ((int64_t)sy * W + x_si[xk]) * 3;
      const int r = max((int)f0[o], (int)f1[o]), g = max((int)f0[o + 1], (int)f1[o + 1]), b = max((int)f0[o + 2], (int)f1[o + 2]);
      const int gray = (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14;
      buf = buf + (float)gray * x_alpha[xk];
    }
    sum = (yk == ya) ? y_alpha[yk] * buf : sum + y_alpha[yk] * buf;
  }
  const float rr = rintf(sum);             // round half to even (cvRound)
  out[i] = (uint8_t)(rr < 0.f ? 0.f : (rr > 255.f ? 255.f : rr));
}

// raw: [n_env][2][H][W][3] uint8 (device): the last two RGB frames of each environment's frame skip; out: [n_env][OH][OW]
// uint8 (device).  The six table arrays are DEVICE copies of dra_resize_area_tab's outputs for (W -> OW) and (H -> OH).
DRA_API int dra_atari_preprocess(const uint8_t* raw, int n_env, int height, int width, int out_h, int out_w, const int* x_si,
                                 const float* x_alpha, const int* x_off, const int* y_si, const float* y_alpha, const int* y_off,
                                 uint8_t* out, void* stream) {
  if (!raw || !out || !x_si || !x_alpha || !x_off || !y_si || !y_alpha || !y_off || n_env < 1 || height < 1 || width < 1 ||
      out_h < 1 || out_w < 1)
    return DRA_EINVAL;
  const int64_t n = (int64_t)n_env * out_h * out_w;
  hipLaunchKernelGGL(atari_preprocess_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, dra_stream(stream), raw, n_env, height,
                     width, out_h, out_w, x_si, x_alpha, x_off, y_si, y_alpha, y_off, out);
  DRA_LAUNCH_CHECK();
  return DRA_OK;
}
