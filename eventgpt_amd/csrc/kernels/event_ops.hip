// Event-stream front end on the GPU: rasterisation of polarity events into RGB
// frames, per-bin polarity histograms / voxel grids, and the fused CLIP
// preprocessing (antialiased bicubic resize + centre crop + normalise -> bf16).
//
// Rasterisation is a scatter with an ordering rule, so it is done with one
// 32-bit atomic per event and a colourise pass:
//   "last event wins"      atomicMax of (event_index << 1 | polarity)
//   "negative wins"        atomicOr  of a 2-bit {has_pos, has_neg} mask
// All T slices of a stream are handled by one launch (slice id from the event
// index via the equal-count boundaries).
#include "common.h"
#include "../evgpt_kernels.h"

namespace evgpt {

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

// keys: [T][H*W] int32, pre-filled with -1 (last) or 0 (negative-wins mask)
__global__ __launch_bounds__(256) void raster_scatter_kernel(
    const int* __restrict__ x, const int* __restrict__ y, const int* __restrict__ p, int n_events,
    int n_slices, int per_slice, int H, int W, int* __restrict__ keys, int mode) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_events; i += gridDim.x * 256) {
    int s = per_slice > 0 ? i / per_slice : 0;
    if (s >= n_slices) s = n_slices - 1;  // the last slice takes the remainder
    const int xi = x[i], yi = y[i];
    if (xi < 0 || xi >= W || yi < 0 || yi >= H) continue;
    int* cell = keys + static_cast<size_t>(s) * H * W + yi * W + xi;
    const int pol = p[i] != 0;
    if (mode == 0) atomicMax(cell, (i << 1) | pol);
    else atomicOr(cell, pol ? 1 : 2);
  }
}

__global__ __launch_bounds__(256) void raster_colorize_kernel(const int* __restrict__ keys,
                                                              unsigned char* __restrict__ img,
                                                              long long n_pix, int mode) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n_pix; i += gridDim.x * 256ll) {
    const int k = keys[i];
    unsigned char r = 255, g = 255, b = 255;
    if (mode == 0) {
      if (k >= 0) { if (k & 1) { g = 0; b = 0; } else { r = 0; g = 0; } }
    } else {
      if (k & 2) { r = 0; g = 0; } else if (k & 1) { g = 0; b = 0; }
    }
    img[i * 3] = r; img[i * 3 + 1] = g; img[i * 3 + 2] = b;
  }
}

void launch_event_raster(const int* x, const int* y, const int* p, int n_events, int n_slices,
                         int H, int W, int* keys, unsigned char* img, int mode, hipStream_t st) {
  const int per = n_events / n_slices;
  int blocks = (n_events + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks > 0)
    hipLaunchKernelGGL(raster_scatter_kernel, dim3(blocks), dim3(256), 0, st, x, y, p, n_events,
                       n_slices, per, H, W, keys, mode);
  const long long n_pix = static_cast<long long>(n_slices) * H * W;
  int cb = static_cast<int>((n_pix + 255) / 256);
  if (cb > 4096) cb = 4096;
  hipLaunchKernelGGL(raster_colorize_kernel, dim3(cb), dim3(256), 0, st, keys, img, n_pix, mode);
}

// ------------------------------------------------------------ voxel / histogram
// Time-binned accumulation.  t is int64 microseconds; bin = floor((t - t0) * bins / span)
// clamped to bins-1.  out_hist: [bins][2][H][W] int32 counts (may be null);
// out_vox: [bins][H][W] fp32, +1/-1 (signed) or +1 per event (may be null).
__global__ __launch_bounds__(256) void voxel_kernel(
    const int* __restrict__ x, const int* __restrict__ y, const int* __restrict__ p,
    const long long* __restrict__ t, int n_events, long long t0, double span, int bins, int H,
    int W, int* __restrict__ out_hist, float* __restrict__ out_vox, int signed_mode, int by_count) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_events; i += gridDim.x * 256) {
    const int xi = x[i], yi = y[i];
    if (xi < 0 || xi >= W || yi < 0 || yi >= H) continue;
    int b;
    if (by_count) b = static_cast<int>(static_cast<long long>(i) * bins / n_events);
    else b = static_cast<int>(static_cast<double>(t[i] - t0) / span * bins);
    b = b < 0 ? 0 : (b >= bins ? bins - 1 : b);
    const int pol = p[i] != 0;
    if (out_hist) atomicAdd(out_hist + ((static_cast<size_t>(b) * 2 + pol) * H + yi) * W + xi, 1);
    if (out_vox)
      atomicAdd(out_vox + (static_cast<size_t>(b) * H + yi) * W + xi,
                pol ? 1.f : (signed_mode ? -1.f : 1.f));
  }
}

void launch_event_voxel(const int* x, const int* y, const int* p, const long long* t, int n_events,
                        long long t0, long long span, int bins, int H, int W, int* out_hist,
                        float* out_vox, int signed_mode, int by_count, hipStream_t st) {
  int blocks = (n_events + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) return;
  const double dspan = static_cast<double>(span > 0 ? span : 1);
  hipLaunchKernelGGL(voxel_kernel, dim3(blocks), dim3(256), 0, st, x, y, p, t, n_events, t0,
                     dspan, bins, H, W, out_hist, out_vox, signed_mode, by_count);
}

// LDS-privatised variant for event streams whose bins are index ranges (binning by count, or by
// time on a time-sorted stream): bounds[b] .. bounds[b+1] are the events of bin b.  The
// per-polarity count image of one bin (2 x H x W int32) does not fit the 160 KB of LDS, so it is
// cut into bands of `rows` image rows; block (chunk, band, bin) scans its share of the bin's
// events, reads only `y` for every event (4 B) and `x`, `p` for the ~rows/H that fall into its
// band, counts them with LDS atomics, and flushes the non-zero counters: one global atomic per
// touched pixel and block instead of one per event.  Reading y once per band costs H/rows x 4 B
// per event, far less than an L2 atomic round trip per event.
constexpr int VL_THREADS = 1024;  // one block per CU (the band fills LDS): 16 waves hide the load latency
constexpr int VL_DEPTH = 4;       // y vectors in flight per thread

__global__ __launch_bounds__(VL_THREADS) void voxel_lds_kernel(
    const int* __restrict__ x, const int* __restrict__ y, const int* __restrict__ p,
    const int* __restrict__ bounds, int H, int W, int rows, int* __restrict__ out_hist,
    float* __restrict__ out_vox, int signed_mode) {
  extern __shared__ int cnt[];  // [2][rows][W]
  const int b = blockIdx.z, y0 = blockIdx.y * rows;
  const int band = min(rows, H - y0);
  const int cells = 2 * rows * W;
  for (int i = threadIdx.x; i < cells; i += VL_THREADS) cnt[i] = 0;
  __syncthreads();
  const int lo = bounds[b], hi = bounds[b + 1];
  // chunks of whole groups of 4 * VL_THREADS events: every load is a 16-byte aligned lane quad
  constexpr int GROUP = 4 * VL_THREADS;
  const int first = lo & ~3;
  const long long groups = (static_cast<long long>(hi) - first + GROUP - 1) / GROUP;
  const long long per = (groups + gridDim.x - 1) / gridDim.x;
  const long long g0 = per * blockIdx.x, g1 = min(groups, g0 + per);
  for (long long g = g0; g < g1; g += VL_DEPTH) {
    i32x4_t yv[VL_DEPTH];
    int base[VL_DEPTH];
#pragma unroll
    for (int d = 0; d < VL_DEPTH; ++d) {      // all y loads of this round first
      base[d] = first + static_cast<int>(g + d) * GROUP + threadIdx.x * 4;
      const bool in = g + d < g1 && base[d] < hi;
      if (in && base[d] + 3 < hi) {
        yv[d] = *reinterpret_cast<const i32x4_t*>(y + base[d]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) yv[d][k] = (in && base[d] + k < hi) ? y[base[d] + k] : -1;
      }
    }
#pragma unroll
    for (int d = 0; d < VL_DEPTH; ++d) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = yv[d][k] - y0, i = base[d] + k;
        if (i < lo || r < 0 || r >= band) continue;
        const int xi = x[i];
        if (xi < 0 || xi >= W) continue;
        atomicAdd(&cnt[((p[i] != 0 ? rows : 0) + r) * W + xi], 1);
      }
    }
  }
  __syncthreads();
  const int band_cells = band * W;
  for (int i = threadIdx.x; i < band_cells; i += VL_THREADS) {
    const int neg = cnt[i], pos = cnt[rows * W + i];
    if ((neg | pos) == 0) continue;
    const size_t pix = static_cast<size_t>(y0) * W + i;
    if (out_hist) {
      if (neg) atomicAdd(out_hist + (static_cast<size_t>(b) * 2) * H * W + pix, neg);
      if (pos) atomicAdd(out_hist + (static_cast<size_t>(b) * 2 + 1) * H * W + pix, pos);
    }
    if (out_vox)
      atomicAdd(out_vox + static_cast<size_t>(b) * H * W + pix,
                static_cast<float>(signed_mode ? pos - neg : pos + neg));
  }
}

// rows of one band: as many as 150 KB of LDS hold for this width (0: image too wide)
int event_voxel_band_rows(int H, int W) {
  const int rows = (150 * 1024 / 4) / (2 * W);
  return rows < H ? rows : H;
}

void launch_event_voxel_ranges(const int* x, const int* y, const int* p, const int* bounds, int n_events,
                               int bins, int H, int W, int* out_hist, float* out_vox, int signed_mode,
                               hipStream_t st, int chunks_override) {
  const int rows = event_voxel_band_rows(H, W);
  const int bands = (H + rows - 1) / rows;
  // just under three full rounds of one block per CU (a partial extra round costs a whole one:
  // 240 / 480 / 720 blocks measured 0.59 ms at 1e7 events, 320 / 800 blocks 0.70-0.72 ms), but no
  // chunk below 64 K events (zeroing and flushing a band costs as much as scanning a few thousand)
  int chunks = (3 * 256) / (bands * bins);
  const int by_size = n_events / bins / 65536;
  if (chunks > by_size) chunks = by_size;
  if (chunks < 1) chunks = 1;
  if (chunks_override > 0) chunks = chunks_override;
  const size_t smem = static_cast<size_t>(2) * rows * W * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&voxel_lds_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(voxel_lds_kernel, dim3(chunks, bands, bins), dim3(VL_THREADS), smem, st, x, y, p, bounds, H, W,
                     rows, out_hist, out_vox, signed_mode);
}

// --------------------------------------------------------- CLIP preprocessing
// img: [T][H][W][3] uint8.  Two-pass fixed-point resample identical to PIL's
// (horizontal pass with uint8 rounding, then vertical), restricted to the
// centre-crop window, then (v/255 - mean)/std -> bf16 [T][3][S][S].
// Tables (int32): start/count per output coordinate, weights [out][ksize] with
// 22 fractional bits.  One thread per output pixel; the source frame is
// L2-resident (0.9 MB), each thread touches ksize_v x ksize_h pixels.
struct PreprocParams {
  const unsigned char* img; bf16_t* out;
  const int* hstart; const int* hcount; const int* hweight;
  const int* vstart; const int* vcount; const int* vweight;
  int T, H, W, S, top, left, ksh, ksv, resize_h, resize_w;
  float mean[3], inv_std[3];
};

__device__ __forceinline__ int clip8_fix(long long acc) {
  const long long v = acc >> 22;
  return v < 0 ? 0 : (v > 255 ? 255 : static_cast<int>(v));
}

__global__ __launch_bounds__(256) void clip_preproc_kernel(PreprocParams p) {
  const long long total = static_cast<long long>(p.T) * p.S * p.S;
  const long long half = 1ll << 21;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
    const int ox = static_cast<int>(i % p.S);
    const int oy = static_cast<int>((i / p.S) % p.S);
    const int t = static_cast<int>(i / (static_cast<long long>(p.S) * p.S));
    const int rx = ox + p.left, ry = oy + p.top;  // coordinates in the resized image
    const unsigned char* frame = p.img + static_cast<size_t>(t) * p.H * p.W * 3;
    int px[3];
    const bool need_h = p.resize_w != p.W, need_v = p.resize_h != p.H;
    const int hs = need_h ? p.hstart[rx] : rx, hc = need_h ? p.hcount[rx] : 1;
    const int vs = need_v ? p.vstart[ry] : ry, vc = need_v ? p.vcount[ry] : 1;
    long long vacc[3] = {half, half, half};
    for (int j = 0; j < vc; ++j) {
      const unsigned char* row = frame + static_cast<size_t>(vs + j) * p.W * 3;
      int hval[3];
      if (need_h) {
        long long hacc[3] = {half, half, half};
        for (int k = 0; k < hc; ++k) {
          const long long w = p.hweight[rx * p.ksh + k];
          const unsigned char* s = row + (hs + k) * 3;
          hacc[0] += w * s[0]; hacc[1] += w * s[1]; hacc[2] += w * s[2];
        }
        hval[0] = clip8_fix(hacc[0]); hval[1] = clip8_fix(hacc[1]); hval[2] = clip8_fix(hacc[2]);
      } else {
        const unsigned char* s = row + hs * 3;
        hval[0] = s[0]; hval[1] = s[1]; hval[2] = s[2];
      }
      if (need_v) {
        const long long w = p.vweight[ry * p.ksv + j];
        vacc[0] += w * hval[0]; vacc[1] += w * hval[1]; vacc[2] += w * hval[2];
      } else {
        px[0] = hval[0]; px[1] = hval[1]; px[2] = hval[2];
      }
    }
    if (need_v) { px[0] = clip8_fix(vacc[0]); px[1] = clip8_fix(vacc[1]); px[2] = clip8_fix(vacc[2]); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = (static_cast<float>(px[c]) * (1.f / 255.f) - p.mean[c]) * p.inv_std[c];
      p.out[((static_cast<size_t>(t) * 3 + c) * p.S + oy) * p.S + ox] = f32_to_bf16(v);
    }
  }
}

void launch_clip_preproc(const unsigned char* img, bf16_t* out, const int* hstart,
                         const int* hcount, const int* hweight, const int* vstart,
                         const int* vcount, const int* vweight, int T, int H, int W, int S,
                         int resize_h, int resize_w, int top, int left, int ksh, int ksv,
                         const float* mean, const float* stdv, hipStream_t st) {
  PreprocParams p;
  p.img = img; p.out = out; p.hstart = hstart; p.hcount = hcount; p.hweight = hweight;
  p.vstart = vstart; p.vcount = vcount; p.vweight = vweight;
  p.T = T; p.H = H; p.W = W; p.S = S; p.top = top; p.left = left; p.ksh = ksh; p.ksv = ksv;
  p.resize_h = resize_h; p.resize_w = resize_w;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean[c]; p.inv_std[c] = 1.f / stdv[c]; }
  const long long total = static_cast<long long>(T) * S * S;
  int blocks = static_cast<int>((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(clip_preproc_kernel, dim3(blocks), dim3(256), 0, st, p);
}

}  // namespace evgpt
