// Convolution weight gradient on MFMA (gfx950):
//   dw[n, tap*C + c] = sum_p dy[p, n] * x[p @ tap, c]           (reduction over pixels p)
//
// As a GEMM the reduction dimension (pixels) is the SLOW dimension of both NHWC operands, while
// MFMA fragments want 8 (bf16) / 4 (f32) consecutive k per lane.  Each thread therefore loads an
// 8x8 (bf16) or 4x4 (f32) [pixel][channel] block with 16-byte loads, transposes it in registers
// (v_perm_b32 pairs for bf16, pure register renaming for f32) and writes [channel][pixel] rows
// with ds_write_b128 into the SAME swizzled 128-byte-row LDS tile format as the forward kernel,
// so the fragment fetch + MFMA core is identical.  One block = one (n-tile, tap, c-tile, split);
// split-K partials go to a workspace and are reduced deterministically.
#include <atomic>

#include "gdl_common.h"

namespace {

struct WArgs {
  gdl_wgrad_args a;
  int64_t P;          // B*Ho*Wo
  int ctiles;         // C / 128
  int splits;
  int64_t p_per_split;  // multiple of the K-step
  int x_dense, dy_dense;
};

template <typename T> struct WT;
template <> struct WT<float> { static constexpr int ES = 4, BKP = 32, BLK = 4; };      // 4x4 blocks
template <> struct WT<bf16_tag> { static constexpr int ES = 2, BKP = 64, BLK = 8; };   // 8x8 blocks

__device__ __forceinline__ void transpose8x8_b16(uint4 (&r)[8]) {
  // rows r[i] = 8 halfwords a[i][0..7]; result r[j] = a[0..7][j]
  uint32_t t[4][8];  // t[i][j] = (a[2i][j], a[2i+1][j])
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t x[4] = {r[2 * i].x, r[2 * i].y, r[2 * i].z, r[2 * i].w};
    const uint32_t y[4] = {r[2 * i + 1].x, r[2 * i + 1].y, r[2 * i + 1].z, r[2 * i + 1].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      t[i][2 * k] = __builtin_amdgcn_perm(y[k], x[k], 0x05040100u);      // (x.lo, y.lo)
      t[i][2 * k + 1] = __builtin_amdgcn_perm(y[k], x[k], 0x07060302u);  // (x.hi, y.hi)
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = make_uint4(t[0][j], t[1][j], t[2][j], t[3][j]);
}

__device__ __forceinline__ void transpose4x4_b32(uint4 (&r)[4]) {
  const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
  r[0] = make_uint4(a.x, b.x, c.x, d.x);
  r[1] = make_uint4(a.y, b.y, c.y, d.y);
  r[2] = make_uint4(a.z, b.z, c.z, d.z);
  r[3] = make_uint4(a.w, b.w, c.w, d.w);
}

// tile = 128 (n) x 128 (c), 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32
template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const WArgs k) {
  constexpr int ES = WT<T>::ES, BKP = WT<T>::BKP, BLK = WT<T>::BLK;
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const gdl_wgrad_args& a = k.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int n0 = blockIdx.x * BM;
  const int tap = blockIdx.y / k.ctiles, c0 = (blockIdx.y % k.ctiles) * BN;
  const int tap_r = tap / a.S, tap_s = tap % a.S;
  // grid.z = (batch index zb -> (z0, z1)) * splits + split-K index
  const int zb = blockIdx.z / k.splits;
  const int split = blockIdx.z - zb * k.splits;
  const int z0 = zb / a.nz_inner, z1 = zb % a.nz_inner;
  const unsigned char* in_p = (const unsigned char*)a.in + (z0 * a.in_sZ0 + z1 * a.in_sZ1) * ES;
  const unsigned char* dy_p = (const unsigned char*)a.dy + (z0 * a.dy_sZ0 + z1 * a.dy_sZ1) * ES;
  float* dw_p = a.dw + z0 * a.dw_sZ0 + z1 * a.dw_sZ1;
  const int64_t p_begin = (int64_t)split * k.p_per_split;
  int64_t p_end = p_begin + k.p_per_split;
  if (p_end > k.P) p_end = k.P;
  const int HoWo = a.Ho * a.Wo;

  // ---- staging roles ----
  // bf16: threads 0..127 stage dy (A), 128..255 stage x (B); one 8x8 block each.
  // f32 : every thread stages one 4x4 block of A and one of B.
  constexpr int GROUPS = 128 / BLK;  // channel groups across the tile: 16 (bf16) / 32 (f32)
  const int bt = (ES == 2) ? (tid & 127) : tid;
  const int cg = bt % GROUPS;        // channel group (BLK channels = 16 bytes)
  const int pg = bt / GROUPS;        // pixel group (BLK pixels) 0..7
  const bool do_a = (ES == 4) || tid < 128;
  const bool do_b = (ES == 4) || tid >= 128;

  uint4 ra[BLK], rb[BLK];

  auto fetch = [&](int64_t pbase) {
#pragma unroll
    for (int i = 0; i < BLK; ++i) {
      const int64_t p = pbase + pg * BLK + i;
      const bool pv = p < p_end;
      int b = 0, oy = 0, ox = 0;
      const int64_t pp = pv ? p : 0;
      if (!(k.x_dense && k.dy_dense)) {
        b = (int)(pp / HoWo);
        const int rem = (int)(pp - (int64_t)b * HoWo);
        oy = rem / a.Wo; ox = rem - oy * a.Wo;
      }
      if (do_a) {
        const int n = n0 + cg * BLK;
        const int64_t off = k.dy_dense ? pp * a.dy_sW
                                       : (int64_t)b * a.dy_sB + (int64_t)oy * a.dy_sH + (int64_t)ox * a.dy_sW;
        ra[i] = (pv && n < a.N) ? *(const uint4*)(dy_p + (off + n) * ES)
                                : make_uint4(0, 0, 0, 0);
      }
      if (do_b) {
        const int c = c0 + cg * BLK;
        bool ok = pv && c < a.C;
        int64_t off;
        if (k.x_dense) {
          off = pp * a.in_sW;
        } else {
          const int iy = oy * a.stride + tap_r - a.pad, ix = ox * a.stride + tap_s - a.pad;
          ok = ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
          off = (int64_t)b * a.in_sB + (int64_t)iy * a.in_sH + (int64_t)ix * a.in_sW;
        }
        rb[i] = ok ? *(const uint4*)(in_p + (off + c) * ES) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto stash = [&](int stage) {
    unsigned char* sa = smem + stage * STAGE_BYTES;
    unsigned char* sb = sa + BM * 128;
    if (do_a) {
      if constexpr (ES == 2) transpose8x8_b16(ra); else transpose4x4_b32(ra);
#pragma unroll
      for (int j = 0; j < BLK; ++j) {
        const int r = cg * BLK + j;  // channel row of the tile
        *(uint4*)(sa + r * 128 + ((pg ^ ((r >> 1) & 7)) << 4)) = ra[j];
      }
    }
    if (do_b) {
      if constexpr (ES == 2) transpose8x8_b16(rb); else transpose4x4_b32(rb);
#pragma unroll
      for (int j = 0; j < BLK; ++j) {
        const int r = cg * BLK + j;
        *(uint4*)(sb + r * 128 + ((pg ^ ((r >> 1) & 7)) << 4)) = rb[j];
      }
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5, fswz = (frow >> 1) & 7;
  const int a_lds0 = (wm * TM * 32 + frow) * 128;
  const int b_lds0 = BM * 128 + (wn * TN * 32 + frow) * 128;

  const int KT = (int)((p_end - p_begin + BKP - 1) / BKP);
  if (KT > 0) {
    fetch(p_begin);
    stash(0);
  }
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = kt + 1 < KT;
    if (more) fetch(p_begin + (int64_t)(kt + 1) * BKP);
    const unsigned char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int coff = (((2 * kk + fhalf) ^ fswz) << 4);
      uint4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *(const uint4*)(st + a_lds0 + i * 32 * 128 + coff);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *(const uint4*)(st + b_lds0 + j * 32 * 128 + coff);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (ES == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8_t, fa[i]), __builtin_bit_cast(bf16x8_t, fb[j]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].x), __uint_as_float(fb[j].x), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].y), __uint_as_float(fb[j].y), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].z), __uint_as_float(fb[j].z), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[i].w), __uint_as_float(fb[j].w), acc[i][j], 0, 0, 0);
          }
        }
    }
    if (more) stash((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: rows = n, cols = c ----
  float* dst;
  int64_t ld;
  if (k.splits > 1) {
    ld = (int64_t)a.R * a.S * a.C;
    dst = a.workspace + ((int64_t)zb * k.splits + split) * a.N * ld;
  } else {
    ld = a.dw_sN;
    dst = dw_p;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
      if (n >= a.N) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c = c0 + (wn * TN + j) * 32 + frow;
        if (c >= a.C) continue;
        float* q = dst + (int64_t)n * ld + (int64_t)tap * a.C + c;
        const float v = acc[i][j][r];
        *q = (k.splits == 1 && a.accumulate) ? *q + v : v;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 production kernel (v2): no register staging at all.
//  * dy / x tiles are DMA'd (global_load_lds_dwordx4) as they lie in memory, [pixel][channel]:
//    per operand 2 panels of [64 pixels][64 channels] = 128-byte rows; out-of-range chunks (halo,
//    pixel tail, channel tail) are sourced from a zero page.
//  * MFMA fragments need 8 consecutive PIXELS per lane: the transpose is done by the LDS itself,
//    ds_read_b64_tr_b16.  Probed semantics (tools/probes/tr_read_probe.hip): within a 16-lane
//    group, result[lane L][j] = the 8-byte piece supplied by lane (4j + L/4), element L%4.  So
//    lane s pointing at LDS[pixel k0 + s/4][channel c0 + 4(s%4)] gives lane L channel c0+L for
//    pixels k0..k0+3; two such reads build one 32x16 fragment row.
//  * Swizzle (source side, because the DMA writes lane-linearly): slot = chunk ^ swz(row) with
//    swz(row) = ((row>>1)&1)<<2 | (row>>2)&3 -- the 4 rows x 64 bytes a half-wave touches per
//    read land in 16 distinct 16-byte slots of the 256-byte bank line.
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

__device__ uint4 g_wgrad_zero_page[8];

__device__ __forceinline__ int tr_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

__global__ __launch_bounds__(256) void wgrad_tr_kernel(const WArgs k) {
  constexpr int BKP = 64, TM = 2, TN = 2;
  constexpr int PANEL = 64 * 128;             // one [64 pixels][64 channels] panel
  constexpr int STAGE_BYTES = 4 * PANEL;      // dy panels 0,1 | x panels 0,1
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const gdl_wgrad_args& a = k.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int n0 = blockIdx.x * 128;
  const int tap = blockIdx.y / k.ctiles, c0 = (blockIdx.y % k.ctiles) * 128;
  const int tap_r = tap / a.S, tap_s = tap % a.S;
  const int zb = blockIdx.z / k.splits;
  const int split = blockIdx.z - zb * k.splits;
  const int z0 = zb / a.nz_inner, z1 = zb % a.nz_inner;
  const unsigned char* in_p = (const unsigned char*)a.in + (z0 * a.in_sZ0 + z1 * a.in_sZ1) * 2;
  const unsigned char* dy_p = (const unsigned char*)a.dy + (z0 * a.dy_sZ0 + z1 * a.dy_sZ1) * 2;
  float* dw_p = a.dw + z0 * a.dw_sZ0 + z1 * a.dw_sZ1;
  const int64_t p_begin = (int64_t)split * k.p_per_split;
  int64_t p_end = p_begin + k.p_per_split;
  if (p_end > k.P) p_end = k.P;
  const int HoWo = a.Ho * a.Wo;
  const unsigned char* zero = (const unsigned char*)g_wgrad_zero_page;

  // ---- DMA geometry: this lane feeds tile rows rr[0], rr[1] (pixels) of all four panels
  const int lrow = lane >> 3, lslot = lane & 7;
  int rr[2], chunk[2], pb[2], py[2], px[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    rr[u] = (wave + 4 * u) * 8 + lrow;
    chunk[u] = lslot ^ tr_swz(rr[u]);
    const int64_t p = p_begin + rr[u];
    const int b = (int)(p / HoWo);
    const int rem = (int)(p - (int64_t)b * HoWo);
    pb[u] = b; py[u] = rem / a.Wo; px[u] = rem - py[u] * a.Wo;
  }

  auto issue = [&](int stage, int64_t pbase) {
    unsigned char* st = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t p = pbase + rr[u];
      const bool pv = p < p_end;
      const int64_t dyo = k.dy_dense ? p * a.dy_sW
                                     : (int64_t)pb[u] * a.dy_sB + (int64_t)py[u] * a.dy_sH + (int64_t)px[u] * a.dy_sW;
      bool xv = pv;
      int64_t xo;
      if (k.x_dense) {
        xo = p * a.in_sW;
      } else {
        const int iy = py[u] * a.stride + tap_r - a.pad, ix = px[u] * a.stride + tap_s - a.pad;
        xv = xv && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        xo = (int64_t)pb[u] * a.in_sB + (int64_t)iy * a.in_sH + (int64_t)ix * a.in_sW;
      }
      const int rowgroup = wave + 4 * u;
#pragma unroll
      for (int pn = 0; pn < 2; ++pn) {
        const int nn = n0 + pn * 64 + chunk[u] * 8;
        const unsigned char* sa = (pv && nn < a.N) ? dy_p + (dyo + nn) * 2 : zero;
        dma16_to_lds(sa, st + pn * PANEL + rowgroup * 1024);
        const int cc = c0 + pn * 64 + chunk[u] * 8;
        const unsigned char* sb = (xv && cc < a.C) ? in_p + (xo + cc) * 2 : zero;
        dma16_to_lds(sb, st + (2 + pn) * PANEL + rowgroup * 1024);
      }
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      px[u] += BKP;
      while (px[u] >= a.Wo) { px[u] -= a.Wo; ++py[u]; }
      while (py[u] >= a.Ho) { py[u] -= a.Ho; ++pb[u]; }
    }
  };

  // ---- fragment addressing (ds_read_b64_tr_b16); see header comment
  const int g = lane >> 4, s = lane & 15;
  const int R0 = 8 * (g >> 1) + (s >> 2);
  int foff[2][2];  // [tile i][read t] byte offset inside a panel (kk adds kk*16*128)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 4 * t + R0;
      const int channel = i * 32 + 16 * (g & 1) + 4 * (s & 3);
      foff[i][t] = row * 128 + (((channel >> 3) ^ tr_swz(row)) << 4) + (channel & 7) * 2;
    }

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = (int)((p_end - p_begin + BKP - 1) / BKP);
  if (KT > 0) issue(0, p_begin);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) { advance(); issue((kt + 1) & 1, p_begin + (int64_t)(kt + 1) * BKP); }
    const unsigned char* st = smem + (kt & 1) * STAGE_BYTES;
    const unsigned char* pa = st + wm * PANEL;         // this wave's 64 n-channels = dy panel wm
    const unsigned char* pbx = st + (2 + wn) * PANEL;  // and its 64 c-channels = x panel wn
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8_t fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + kk * 2048 + foff[i][0]));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + kk * 2048 + foff[i][1]));
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fa[i] = __builtin_bit_cast(bf16x8_t, v);
        const s16x4_t lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pbx + kk * 2048 + foff[i][0]));
        const s16x4_t hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pbx + kk * 2048 + foff[i][1]));
        const s16x8_t v2 = {lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
        fb[i] = __builtin_bit_cast(bf16x8_t, v2);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: rows = n, cols = c (coalesced along c) ----
  const int frow = lane & 31, fhalf = lane >> 5;
  float* dst;
  int64_t ld;
  if (k.splits > 1) {
    ld = (int64_t)a.R * a.S * a.C;
    dst = a.workspace + ((int64_t)zb * k.splits + split) * a.N * ld;
  } else {
    ld = a.dw_sN;
    dst = dw_p;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
      if (n >= a.N) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c = c0 + (wn * TN + j) * 32 + frow;
        if (c >= a.C) continue;
        float* q = dst + (int64_t)n * ld + (int64_t)tap * a.C + c;
        const float v = acc[i][j][r];
        *q = (k.splits == 1 && a.accumulate) ? *q + v : v;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 wide-layer kernel: 256 (n) x 256 (c) output tile, 8 waves (2 x 4), wave tile 128 x 64, K-step = 64 pixels.
// Twice the arithmetic intensity of the 128^2 kernel per staged byte (the 128^2 kernel issues 8 KiB of LDS-DMA per
// 16 MFMAs and is bound by DMA issue).  Operands arrive through buffer descriptors (32-bit offsets, out-of-range
// chunk -> hardware zero fill), fragments through ds_read_b64_tr_b16 with double-buffered registers; the K loop is
// the software-pipelined single-barrier loop of conv_gemm.hip.
struct W256 {
  WArgs w;
  unsigned in_span, dy_span;   // bytes, < 2 GiB
  int tiles_n, tiles_y;        // n tiles, taps * c tiles
  int xcd_group;               // 8 = workgroup ids as dealt by the hardware
};


__global__ __launch_bounds__(512) void wgrad_tr256_kernel(const W256 kk) {
  constexpr int BKP = 64, TM = 4, TN = 2;
  constexpr int PANEL = 64 * 128;             // one [64 pixels][64 channels] panel
  constexpr int STAGE_BYTES = 8 * PANEL;      // dy panels 0..3 | x panels 0..3
  constexpr unsigned kOob = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WArgs& k = kk.w;
  const gdl_wgrad_args& a = k.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // 1-D grid in (n-tile, tap * c-tile, split) order, XCD-major since round 6: the tiles of one pixel range sit on ONE XCD and
  // share its L2 (the hardware deals consecutive workgroup ids to the eight XCDs in turn, so every L2 used to fetch every panel:
  // PMC 1.28 GB fetched per launch at batch 64, L2 hit 0.32).  The same remap was 5 % slower under the round-3 split rule;
  // with round 4's fewer, longer splits three same-box A/B runs of the whole step give +0.45 % (991.2 -> 995.7 train tiles/s;
  // 2 / 4 XCDs per range: +0.2 / +0.3 %).  xcd_group = 8 (gdl_debug_set_wgrad_xcd_group) restores the hardware order.
  int lid = blockIdx.x;
  if (kk.xcd_group < 8) {
    const int G = kk.xcd_group, nb = (int)gridDim.x, q8 = nb >> 3, r8 = nb & 7;
    const int x = lid & 7, i = lid >> 3, s = x / G, m = x - s * G;
    int base = 0;
    for (int xx = 0; xx < s * G; ++xx) base += q8 + (xx < r8 ? 1 : 0);
    lid = base + i * G + m;
  }
  const int per_split = kk.tiles_n * kk.tiles_y;
  const int bz = lid / per_split, bt = lid - bz * per_split;
  const int by = bt / kk.tiles_n, bx = bt - by * kk.tiles_n;
  const int n0 = bx * 256;
  const int tap = by / k.ctiles, c0 = (by % k.ctiles) * 256;
  const int tap_r = tap / a.S, tap_s = tap % a.S;
  const int zb = bz / k.splits;
  const int split = bz - zb * k.splits;
  const int z0 = zb / a.nz_inner, z1 = zb % a.nz_inner;
  const srd_t srd_x = make_srd((const unsigned char*)a.in + (z0 * a.in_sZ0 + z1 * a.in_sZ1) * 2, kk.in_span);
  const srd_t srd_dy = make_srd((const unsigned char*)a.dy + (z0 * a.dy_sZ0 + z1 * a.dy_sZ1) * 2, kk.dy_span);
  float* dw_p = a.dw + z0 * a.dw_sZ0 + z1 * a.dw_sZ1;
  const int64_t p_begin = (int64_t)split * k.p_per_split;
  int64_t p_end = p_begin + k.p_per_split;
  if (p_end > k.P) p_end = k.P;
  const int HoWo = a.Ho * a.Wo;
  const unsigned lds_base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);

  // ---- DMA geometry ("ping-pong": the two waves of a SIMD, w and w+4, alternate as the loader of a K-step, so
  // one of them is always in its MFMA phase): loader wave (w&3) feeds pixel rows 8(w&3)+64u.. of all eight panels
  // for u = 0,1 -- i.e. row groups (w&3) and (w&3)+4; lane l row (l>>3), slot (l&7)
  const int lrow = lane >> 3, lslot = lane & 7;
  const int half = wave >> 2, lwave = wave & 3;
  int rr[2], chunk[2], pb[2], py[2], px[2];   // (batch, oy, ox) of this lane's pixels in the NEXT tile to fetch
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    rr[u] = (lwave + 4 * u) * 8 + lrow;
    chunk[u] = lslot ^ tr_swz(rr[u]);
    const int64_t p = p_begin + rr[u];
    pb[u] = (int)(p / HoWo);
    const int rem = (int)(p - (int64_t)pb[u] * HoWo);
    py[u] = rem / a.Wo;
    px[u] = rem - py[u] * a.Wo;
  }
  int64_t pnext = p_begin;   // first pixel of the NEXT tile to fetch

  auto issue = [&](int stage, bool load) {
    if (load) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const unsigned lds = lds_base + stage * STAGE_BYTES + (lwave + 4 * u) * 1024;
        const bool pv = pnext + rr[u] < p_end;
        unsigned dyo, xo;
        bool xv = pv;
        if (k.dy_dense) dyo = (unsigned)((pnext + rr[u]) * a.dy_sW * 2);
        else dyo = (unsigned)((pb[u] * a.dy_sB + py[u] * a.dy_sH + px[u] * a.dy_sW) * 2);
        if (k.x_dense) {
          xo = (unsigned)((pnext + rr[u]) * a.in_sW * 2);
        } else {
          const int iy = py[u] * a.stride + tap_r - a.pad, ix = px[u] * a.stride + tap_s - a.pad;
          xv = xv && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
          xo = (unsigned)((pb[u] * a.in_sB + iy * a.in_sH + ix * a.in_sW) * 2);
        }
#pragma unroll
        for (int pn = 0; pn < 4; ++pn) {
          const int nn = n0 + pn * 64 + chunk[u] * 8, cc = c0 + pn * 64 + chunk[u] * 8;
          const unsigned vd = (pv && nn < a.N) ? dyo + (unsigned)nn * 2u : kOob;
          dma16_buf(vd, srd_dy, 0u, lds + pn * PANEL);
          const unsigned vx = (xv && cc < a.C) ? xo + (unsigned)cc * 2u : kOob;
          dma16_buf(vx, srd_x, 0u, lds + (4 + pn) * PANEL);
        }
      }
    }
    pnext += BKP;   // every wave tracks the position, loader or not
    if (!(k.dy_dense && k.x_dense)) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        px[u] += BKP;
        while (px[u] >= a.Wo) { px[u] -= a.Wo; ++py[u]; }
        while (py[u] >= a.Ho) { py[u] -= a.Ho; ++pb[u]; }
      }
    }
  };

  // ---- fragment addressing (ds_read_b64_tr_b16; see the 128^2 kernel's header comment)
  const int g = lane >> 4, s = lane & 15;
  const int R0 = 8 * (g >> 1) + (s >> 2);
  int foff[2][2];   // [32-channel tile inside a 64-channel panel][read t]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 4 * t + R0;
      const int channel = i * 32 + 16 * (g & 1) + 4 * (s & 3);
      foff[i][t] = row * 128 + (((channel >> 3) ^ tr_swz(row)) << 4) + (channel & 7) * 2;
    }
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  bf16x8_t fa[2][TM], fb[2][TN];
  auto fetch = [&](const unsigned char* st, int kq, int buf) {
    const unsigned char* pa = st + (wm * 2) * PANEL + kq * 2048;   // this wave's 128 n = dy panels 2wm, 2wm+1
    const unsigned char* px_ = st + (4 + wn) * PANEL + kq * 2048;  // its 64 c = x panel wn
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned char* q = pa + (i >> 1) * PANEL;
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q + foff[i & 1][0]));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q + foff[i & 1][1]));
      const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      fa[buf][i] = __builtin_bit_cast(bf16x8_t, v);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(px_ + foff[j][0]));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(px_ + foff[j][1]));
      const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      fb[buf][j] = __builtin_bit_cast(bf16x8_t, v);
    }
  };
  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto mfmas = [&](int buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][i], fb[buf][j], acc[i][j], 0, 0, 0);
  };

  const int KT = (int)((p_end - p_begin + BKP - 1) / BKP);
  if (KT > 0) {
    issue(0, half == 0);                                 // tile t is loaded by half (t & 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (KT > 1) issue(1, half == 1);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): no scalar load is pending past this point (see conv_gemm.hip)
    fetch(smem, 0, 0);
    for (int kt = 0; kt < KT; ++kt) {
      const unsigned char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
      for (int kq = 0; kq < 3; ++kq) {
        fetch(st, kq + 1, (kq + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(kq & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kt + 1 < KT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < KT) issue(kt & 1, half == (kt & 1));
        fetch(smem + ((kt + 1) & 1) * STAGE_BYTES, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfmas(1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: rows = n (registers), cols = c (lane): 128-byte coalesced f32 rows
  const int frow = lane & 31, fhalf = lane >> 5;
  float* dst;
  int64_t ld;
  if (k.splits > 1) {
    ld = (int64_t)a.R * a.S * a.C;
    dst = a.workspace + ((int64_t)zb * k.splits + split) * a.N * ld;
  } else {
    ld = a.dw_sN;
    dst = dw_p;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
      if (n >= a.N) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c = c0 + (wn * TN + j) * 32 + frow;
        if (c >= a.C) continue;
        float* q = dst + (int64_t)n * ld + (int64_t)tap * a.C + c;
        const float v = acc[i][j][r];
        *q = (k.splits == 1 && a.accumulate) ? *q + v : v;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 row-segment kernel for 3x3 / stride 1 / pad 1 convolutions on feature maps at least 32 pixels wide
// (UNet++ decoder, ResNet / MiT high-resolution levels).  The per-tap kernels above re-stage the SAME input pixels for
// each of the nine taps and, on 64-channel layers, fill half of every 128-wide tile with zeros; here one block owns
// (64 n) x (64 c) x ALL NINE taps and walks row segments (b, y, x0..x0+len-1) of up to 64 pixels:
//   LDS stage = dy [64 px][64 n]  +  x rows y-1, y, y+1 as [66 px: x0-1 .. x0+64][64 c]      (35 KiB, two stages)
//   the nine taps are the three staged rows read at pixel shifts 0, 1, 2, so every staged byte feeds 9 (dy) / 3 (x)
//   MFMA operands: 144 MFMAs per 33.5 KiB staged instead of 16 per 32 KiB.
// Image borders are zero-filled by the buffer descriptor (voffset = OOB); the x descriptor starts one pixel BEFORE
// the tensor so that panel pixel 0 (x0 - 1) has a non-negative offset (it is masked whenever x0 == 0).
struct WRows {
  WArgs w;
  unsigned in_span, dy_span;   // bytes, < 2 GiB
  int ntiles, cchunks;         // ceil(N / 64), ceil(C / 64)
  int seglen, segs_per_row;    // pixels per row segment (multiple of 16, <= 64), ceil(W / seglen)
  int nsegs, segs_per_split;
  int xcd_group;               // 1: all (n, c) tiles of one pixel range are dealt to ONE XCD (grid.y padded to a multiple of 8)
};

__device__ __forceinline__ int k_splits_of(const WRows& kk) { return kk.w.splits; }

// Work split: 4 waves, wave (i, j) owns the 32 (n) x 32 (c) quadrant of the 64 x 64 tile for ALL nine taps
// (9 accumulators = 144 registers), so all four SIMDs carry MFMA work and the kernel stays under 256 registers: two
// blocks share a CU (2 x 70 KiB of LDS) and one block's DMA issue / barrier waits are covered by the other's MFMAs.
// (A 3-wave variant -- one wave per kernel row, 192 accumulator registers, one block per CU -- measured 5-45 % slower.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_rows_kernel(const WRows kk) {
  constexpr int DYB = 64 * 128;
  constexpr int XROW = 72 * 128;
  constexpr int STAGE_BYTES = DYB + 3 * XROW;
  constexpr unsigned kOob = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WArgs& k = kk.w;
  const gdl_wgrad_args& a = k.a;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = wave & 1, wj = wave >> 1;            // n half, c half
  // Round 5: the ntiles x cchunks tiles of one pixel range read the SAME dy panels (shared by the c chunks) and input rows (shared
  // by the n tiles).  Dealt out in launch order they land on all eight XCDs and every XCD's L2 fetches every panel again (PMC,
  // round 4: 3.5 GB fetched per launch of the 256 -> 256 layer at 144^2 for 0.68 GB of operands, L2 hit rate 37 %, the kernel
  // HBM-bound at 4.85 TB/s).  With xcd_group the hardware's round-robin (workgroup id % 8 = XCD) is inverted: XCD x owns pixel
  // ranges x, x + 8, ... and runs all their tiles together.
  int tile = blockIdx.x, split = blockIdx.y;
  if (kk.xcd_group) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int j = lin >> 3;
    tile = j % (int)gridDim.x;
    split = (j / (int)gridDim.x) * 8 + (lin & 7);
    if (split >= k_splits_of(kk)) return;
  }
  const int n0 = (tile % kk.ntiles) * 64, c0 = (tile / kk.ntiles) * 64;
  const unsigned xpix = (unsigned)(a.in_sW * 2), dypix = (unsigned)(a.dy_sW * 2);
  const srd_t srd_x = make_srd((const unsigned char*)a.in - xpix, kk.in_span + xpix);
  const srd_t srd_dy = make_srd((const unsigned char*)a.dy, kk.dy_span);
  const int seg_begin = split * kk.segs_per_split;
  const int seg_end = seg_begin + kk.segs_per_split < kk.nsegs ? seg_begin + kk.segs_per_split : kk.nsegs;
  const unsigned lds_base =
      __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);

  // ---- DMA: 9 pieces per wave and stage.  Waves 0..2 stage input rows y-1, y, y+1; wave 3 the dy panel (8 pieces,
  // the last one twice so that every wave has the same number of loads in flight).
  const int lrow = lane >> 3, lslot = lane & 7;
  const int chunk_e = lslot ^ tr_swz(lrow), chunk_o = lslot ^ tr_swz(8 + lrow);
  const bool is_dy = wave == 3;
  const int ch0 = is_dy ? n0 : c0, chn = is_dy ? a.N : a.C;
  const unsigned pix = is_dy ? dypix : xpix;
  const unsigned v_e = ch0 + chunk_e * 8 < chn ? lrow * pix + (unsigned)(ch0 + chunk_e * 8) * 2u : kOob;
  const unsigned v_o = ch0 + chunk_o * 8 < chn ? lrow * pix + (unsigned)(ch0 + chunk_o * 8) * 2u : kOob;
  int sx = seg_begin % kk.segs_per_row;
  int sy = (seg_begin / kk.segs_per_row) % a.H;
  int sb = seg_begin / kk.segs_per_row / a.H;

  // A segment covers pixels x0 .. x0+len-1 of one image row, len <= seglen (a multiple of 16; the last segment of a
  // row may be shorter).  It is multiplied in nkq = ceil(len / 16) groups of 16 pixels, so the panels are filled up
  // to 16 * nkq (+2 halo) pixels and everything outside the image / past len arrives as zeros.
  auto seg_nkq = [&](int sxi) {
    const int rem = a.W - sxi * kk.seglen;
    return ((rem < kk.seglen ? rem : kk.seglen) + 15) >> 4;
  };
  auto issue = [&](int stage) {
    const unsigned lds = lds_base + stage * STAGE_BYTES;
    const int x0 = sx * kk.seglen;
    const int nkq = seg_nkq(sx);
    if (is_dy) {
      const unsigned soff = (unsigned)((sb * a.dy_sB + sy * a.dy_sH + (int64_t)x0 * a.dy_sW) * 2);
      const int rem = a.W - x0 - lrow;                    // piece * 8 < rem <=> the lane's pixel is inside the row
      const int len = (a.W - x0 < kk.seglen ? a.W - x0 : kk.seglen) - lrow;
#pragma unroll
      for (int piece = 0; piece < 8; ++piece) {
        if (piece < 2 * nkq) {
          unsigned v = (piece & 1) ? v_o : v_e;
          if (piece * 8 >= rem || piece * 8 >= len) v = kOob;
          dma16_buf(v, srd_dy, soff + piece * 8 * dypix, lds + piece * 1024);
        }
      }
    } else {
      const int iy = sy + wave - 1;
      const bool rowok = (unsigned)iy < (unsigned)a.H;
      const unsigned soff = (unsigned)((sb * a.in_sB + iy * a.in_sH + (int64_t)x0 * a.in_sW) * 2);
      const unsigned ldx = lds + DYB + wave * XROW;
      const int ixl = x0 - 1 + lrow;                      // image x of the lane's pixel in piece 0
#pragma unroll
      for (int piece = 0; piece < 9; ++piece) {
        if (piece < 2 * nkq + 1) {
          unsigned v = (piece & 1) ? v_o : v_e;
          if (!rowok || (unsigned)(ixl + piece * 8) >= (unsigned)a.W) v = kOob;
          dma16_buf(v, srd_x, soff + piece * 8 * xpix, ldx + piece * 1024);
        }
      }
    }
    if (++sx == kk.segs_per_row) { sx = 0; if (++sy == a.H) { sy = 0; ++sb; } }
  };
  int csx = sx;                                           // segment-in-row index of the segment being multiplied

  // ---- fragment addressing: this wave's 32-channel tile of the dy panel (wi) and of the x rows (wj)
  const int g = lane >> 4, s = lane & 15;
  const int R0 = 8 * (g >> 1) + (s >> 2);
  int foff_dy[2], foff_x[3][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = 4 * t + R0;
    const int ch = wi * 32 + 16 * (g & 1) + 4 * (s & 3);
    foff_dy[t] = row * 128 + (((ch >> 3) ^ tr_swz(row)) << 4) + (ch & 7) * 2;
#pragma unroll
    for (int sh = 0; sh < 3; ++sh) {
      const int rx = row + sh;
      const int cx = wj * 32 + 16 * (g & 1) + 4 * (s & 3);
      foff_x[sh][t] = DYB + rx * 128 + (((cx >> 3) ^ tr_swz(rx)) << 4) + (cx & 7) * 2;
    }
  }
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  auto frag = [&](const unsigned char* q, int o0, int o1) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q + o0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(q + o1));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  f32x16_t acc[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int sh = 0; sh < 3; ++sh)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][sh][q] = 0.f;

  const int KT = seg_end - seg_begin;
  if (KT > 0) issue(0);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // segment kt landed everywhere; stage (kt+1)&1 fully read
    if (kt + 1 < KT) issue((kt + 1) & 1);
    const unsigned char* st = smem + (kt & 1) * STAGE_BYTES;
    // 12 groups (kq, r) of three MFMAs; the fragments of group g+1 are fetched while group g multiplies
    bf16x8_t fa[2], fb[2][3];
    auto fetch = [&](int grp) {
      const int kq = grp / 3, r = grp - kq * 3;
      const unsigned char* q = st + kq * 2048;
      if (r == 0) fa[kq & 1] = frag(q, foff_dy[0], foff_dy[1]);
#pragma unroll
      for (int sh = 0; sh < 3; ++sh) fb[grp & 1][sh] = frag(q + r * XROW, foff_x[sh][0], foff_x[sh][1]);
    };
    const int ngrp = 3 * seg_nkq(csx);
    if (++csx == kk.segs_per_row) csx = 0;
    fetch(0);
#pragma unroll
    for (int grp = 0; grp < 12; ++grp) {
      if (grp < ngrp) {
        if (grp + 1 < 12 && grp + 1 < ngrp) fetch(grp + 1);
        __builtin_amdgcn_sched_barrier(0);
        const int kq = grp / 3, r = grp - kq * 3;
#pragma unroll
        for (int sh = 0; sh < 3; ++sh)
          acc[r][sh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kq & 1], fb[grp & 1][sh], acc[r][sh], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue: rows = n (registers), cols = c (lane)
  const int frow = lane & 31, fhalf = lane >> 5;
  float* dst;
  int64_t ld;
  if (k.splits > 1) {
    ld = (int64_t)9 * a.C;
    dst = a.workspace + (int64_t)split * a.N * ld;
  } else {
    ld = a.dw_sN;
    dst = a.dw;
  }
  const int c = c0 + wj * 32 + frow;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int sh = 0; sh < 3; ++sh) {
      const int tap = r * 3 + sh;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = n0 + wi * 32 + (q & 3) + 8 * (q >> 2) + 4 * fhalf;
        if (n >= a.N || c >= a.C) continue;
        float* o = dst + (int64_t)n * ld + (int64_t)tap * a.C + c;
        const float v = acc[r][sh][q];
        *o = (k.splits == 1 && a.accumulate) ? *o + v : v;
      }
    }
}

// split-K reduction for many partials: 32 outputs x 8 partial groups per block, fixed summation order
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ ws, int splits, int N,
                                                                int64_t K, float* dw, int64_t dw_sN, int accumulate) {
  __shared__ float part[8][32];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t total = (int64_t)N * K;
  const int64_t i = (int64_t)blockIdx.x * 32 + col;
  float s = 0.f;
  if (i < total) s = ordered_sum8<float>(grp, splits, 8, [&](int sp) { return ws[(int64_t)sp * total + i]; });
  part[grp][col] = s;
  __syncthreads();
  if (grp != 0 || i >= total) return;
  s = ((part[0][col] + part[1][col]) + (part[2][col] + part[3][col])) +
      ((part[4][col] + part[5][col]) + (part[6][col] + part[7][col]));
  const int64_t n = i / K, kq = i - n * K;
  float* q = dw + n * dw_sN + kq;
  *q = accumulate ? *q + s : s;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int N,
                                                           int64_t K, float* dw, int64_t dw_sN, int accumulate,
                                                           int nz_inner, int64_t dw_sZ0, int64_t dw_sZ1) {
  const int64_t total = (int64_t)N * K;
  const int zb = blockIdx.y;
  ws += (int64_t)zb * splits * total;
  dw += (zb / nz_inner) * dw_sZ0 + (zb % nz_inner) * dw_sZ1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float s = ordered_sum8<float>(0, splits, 1, [&](int sp) { return ws[(int64_t)sp * total + i]; });
    const int64_t n = i / K, kk = i - n * K;
    float* q = dw + n * dw_sN + kk;
    *q = accumulate ? *q + s : s;
  }
}

int64_t span_bytes(int B, int H, int W, int C, int64_t sB, int64_t sH, int64_t sW, int es) {
  return (((int64_t)B - 1) * sB + ((int64_t)H - 1) * sH + ((int64_t)W - 1) * sW + C) * es;
}

std::atomic<int> g_wgrad_old_splits{0};    // A/B hook: the round-3 split-K rule of the 256^2 kernel
// the row-segment kernel runs all (n, c) tiles of a pixel range on one XCD (round 5, profiles/r05h_*: 256 -> 256 at 144^2
// 783 -> 745 us, at 128^2 681 -> 607 us, no layer slower; bit-identical).  0 = launch order (GDL_WGRAD_ROWS_XCD=0)
std::atomic<int> g_wgrad_rows_xcd{1};
int wgrad_num_cus() {
  static int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    return cus;
  }();
  return n;
}
std::atomic<int> g_wgrad_force_small{0};   // A/B hook: bit 0 = never use the 256^2 kernel, bit 1 = never use the row-segment kernel, bit 3 = N, C >= 256 layers on the 256^2 kernel

// 256^2 tiles for wide bf16 layers whose operands fit 32-bit buffer offsets
int wgrad_tile(const gdl_wgrad_args& a) {
  if (a.dtype != GDL_BF16 || (g_wgrad_force_small & 1) || a.N < 256 || a.C < 256) return 128;
  if (span_bytes(a.B, a.H, a.W, a.C, a.in_sB, a.in_sH, a.in_sW, 2) > 0x7ffffff0ll ||
      span_bytes(a.B, a.Ho, a.Wo, a.N, a.dy_sB, a.dy_sH, a.dy_sW, 2) > 0x7ffffff0ll)
    return 128;
  return 256;
}

int rows_seglen(int W);
// row-segment kernel: every 3x3 / stride 1 / pad 1 layer on a map at least 32 pixels wide
bool wgrad_rows_ok(const gdl_wgrad_args& a) {
  if (a.dtype != GDL_BF16 || (g_wgrad_force_small & 2)) return false;
  if ((g_wgrad_force_small & 8) && wgrad_tile(a) == 256) return false;   // A/B hook: wide layers on the 256^2 per-tap kernel
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.H != a.Ho || a.W != a.Wo || a.W < 32 || a.nz != 1)
    return false;
  // rows are staged in 16-pixel units: a 36-wide map pays for 48 pixels per row.  Where the per-tap 256^2 kernel exists (N, C >=
  // 256) it is faster from 20 % padding on (768 -> 768 at 36^2: 563 -> 411 us, profiles/r04e_bench_wgrad_split_rule.txt)
  if (!g_wgrad_old_splits && wgrad_tile(a) == 256) {
    const int seglen = rows_seglen(a.W);
    if (5 * seglen * ((a.W + seglen - 1) / seglen) >= 6 * a.W) return false;
    // ... and, with the round-4 split rule, for the widest layers at any width (768 -> 768: 1.10-1.15 PF against 0.97-1.02,
    // 1024 -> 256: 1.14-1.16 against 1.05-1.11; 256 -> 256 stays 4-16 % faster on the row segments)
    if ((int64_t)a.N * a.C >= 256 * 1024) return false;
  }
  return span_bytes(a.B, a.H, a.W, a.C, a.in_sB, a.in_sH, a.in_sW, 2) + a.in_sW * 2 <= 0x7ffffff0ll &&
         span_bytes(a.B, a.Ho, a.Wo, a.N, a.dy_sB, a.dy_sH, a.dy_sW, 2) <= 0x7ffffff0ll;
}

// rows are cut into ceil(W / 64) segments of equal length in 16-pixel units: 144 -> 3 x 48, 72 -> 48 + 24, 128 -> 2 x 64
int rows_seglen(int W) {
  const int u = (W + 15) / 16, n = (W + 63) / 64;
  return 16 * ((u + n - 1) / n);
}

int choose_splits(const gdl_wgrad_args& a, int64_t P, int bkp) {
  if (wgrad_rows_ok(a)) {
    // two blocks per CU: ~512 blocks, >= 8 segments per split
    const int64_t tiles = (int64_t)((a.N + 63) / 64) * ((a.C + 63) / 64);
    int64_t want = 512 / tiles;
    const int seglen = rows_seglen(a.W);
    const int64_t max_by_k = (int64_t)a.B * a.H * ((a.W + seglen - 1) / seglen) / 8;
    if (want > max_by_k) want = max_by_k;
    if (want > 512) want = 512;
    if (want < 1) want = 1;
    return (int)want;
  }
  const int t = wgrad_tile(a);
  const int64_t tiles = (int64_t)((a.N + t - 1) / t) * a.R * a.S * ((a.C + t - 1) / t) * (a.nz > 1 ? a.nz : 1);
  const int64_t max_by_k = P / (8 * bkp);              // keep >= 8 K-steps per split
  if (a.dtype == GDL_BF16 && !g_wgrad_old_splits) {
    // Round 4: cost model instead of "about 512 (256^2) / 1024 (128^2) workgroups".  The 256^2 kernel holds a CU alone (512
    // threads, 128 KiB of LDS), the 128^2 kernel shares it with one other workgroup (64 KiB each), so the split count decides
    // how the launch falls into rounds of `slots` workgroups -- and the old rule always produced slots * 2 .. slots * 2 + tiles
    // - 1 of them: two full rounds plus a nearly empty third (lateral 768 -> 768 at 36^2: 9 tiles x 57 splits = 513 workgroups,
    // 120 us for 49 GF; now 28 splits = 252 workgroups, 89 us).  Per candidate: rounds x (K-steps of a split x cycles per
    // K-step + prologue and f32 partial-tile store) + the reduction pass over `splits` partial matrices at ~3 TB/s.
    const int64_t slots = (int64_t)wgrad_num_cus() * (t == 256 ? 1 : 2);
    const double step_cyc = t == 256 ? 3000.0 : 1700.0, fixed_cyc = t == 256 ? 25000.0 : 11000.0;
    const int64_t ksteps = (P + bkp - 1) / bkp;
    const double out_bytes = (double)a.N * a.R * a.S * a.C * 4.0 * (a.nz > 1 ? a.nz : 1);
    const int64_t smax = std::max<int64_t>(1, std::min<int64_t>(max_by_k, a.nz > 1 ? 64 : 512));
    auto cost_us = [&](int64_t sp) {
      const int64_t rounds = (tiles * sp + slots - 1) / slots;
      const double main_us = rounds * ((double)((ksteps + sp - 1) / sp) * step_cyc + fixed_cyc) / 1800.0;
      return main_us + (sp > 1 ? sp * out_bytes / 3.0e6 + 8.0 : 0.0);
    };
    double best_us = 1e30;
    for (int64_t sp = 1; sp <= smax; ++sp) best_us = std::min(best_us, cost_us(sp));
    for (int64_t sp = 1; sp <= smax; ++sp)            // the fewest splits within 2 % of the best: less workspace, less reduction traffic
      if (cost_us(sp) <= 1.02 * best_us) return (int)sp;
    return 1;
  }
  int64_t want = ((t == 256 ? 512 : 1024) + tiles - 1) / tiles;   // aim for ~1024 (512 big-tile) blocks
  if (want > max_by_k) want = max_by_k;
  // few-tile layers (narrow linears over many pixels: MiT stage 1-2) need many splits to fill 256 CUs; the wide
  // reduction kernel makes them cheap.  Batched calls (attention dK / dV) already have nz-fold parallelism.
  const int64_t cap = a.nz > 1 ? 64 : 512;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

}  // namespace

static std::atomic<int> g_wgrad_force_v1{0};
extern "C" void gdl_debug_set_wgrad_old_splits(int on) { g_wgrad_old_splits = on; }  // A/B hook: round-3 split-K rule
static int g_wgrad_xcd_group = 1;
extern "C" void gdl_debug_set_wgrad_xcd_group(int g) { g_wgrad_xcd_group = (g == 1 || g == 2 || g == 4) ? g : 8; }   // A/B hook: XCDs a pixel range's tiles are dealt to
extern "C" void gdl_debug_set_wgrad_rows_xcd(int on) { g_wgrad_rows_xcd = on; }      // A/B hook: XCD grouping of the row-segment kernel
extern "C" void gdl_debug_force_wgrad_small(int on) { g_wgrad_force_small = on; }  // A/B hook: 128^2 tiles only
extern "C" void gdl_debug_force_wgrad_v1(int on) { g_wgrad_force_v1 = on; }  // A/B hook: register-transpose kernel

extern "C" int64_t gdl_conv_wgrad_workspace(const gdl_wgrad_args* ap) {
  if (!ap) return 0;
  const gdl_wgrad_args& a = *ap;
  const int64_t P = (int64_t)a.B * a.Ho * a.Wo;
  const int bkp = a.dtype == GDL_BF16 ? 64 : 32;
  const int splits = choose_splits(a, P, bkp);
  if (splits <= 1) return 0;
  return (int64_t)splits * (a.nz > 1 ? a.nz : 1) * a.N * a.R * a.S * a.C * (int64_t)sizeof(float);
}

extern "C" int gdl_conv_wgrad(const gdl_wgrad_args* ap, gdl_stream_t stream) {
  GDL_CHECK_ARG(ap != nullptr, "gdl_conv_wgrad: null args");
  const gdl_wgrad_args& a = *ap;
  GDL_CHECK_ARG(a.dtype == GDL_F32 || a.dtype == GDL_BF16, "gdl_conv_wgrad: bad dtype");
  GDL_CHECK_ARG(a.in && a.dy && a.dw, "gdl_conv_wgrad: null tensor");
  const int es = (int)gdl_elem_size(a.dtype), al = 16 / es;
  GDL_CHECK_ARG(a.C % al == 0 && a.N % al == 0, "gdl_conv_wgrad: C and N must be multiples of %d", al);
  GDL_CHECK_ARG(a.in_sB % al == 0 && a.in_sH % al == 0 && a.in_sW % al == 0 && a.dy_sB % al == 0 &&
                    a.dy_sH % al == 0 && a.dy_sW % al == 0, "gdl_conv_wgrad: strides must keep 16-byte alignment");
  GDL_CHECK_ARG(((uintptr_t)a.in % 16 == 0) && ((uintptr_t)a.dy % 16 == 0), "gdl_conv_wgrad: pointers must be 16-byte aligned");
  GDL_CHECK_ARG(a.nz >= 1 && a.nz_inner >= 1 && a.nz <= 65535, "gdl_conv_wgrad: nz/nz_inner must be >= 1");
  GDL_CHECK_ARG(a.in_sZ0 % al == 0 && a.in_sZ1 % al == 0 && a.dy_sZ0 % al == 0 && a.dy_sZ1 % al == 0,
                "gdl_conv_wgrad: batch strides must keep 16-byte alignment");
  WArgs k;
  k.a = a;
  k.P = (int64_t)a.B * a.Ho * a.Wo;
  k.ctiles = (a.C + 127) / 128;
  const int bkp = a.dtype == GDL_BF16 ? 64 : 32;
  k.splits = choose_splits(a, k.P, bkp);
  if (k.splits > 1) {
    const int64_t need = (int64_t)k.splits * a.nz * a.N * a.R * a.S * a.C * (int64_t)sizeof(float);
    GDL_CHECK_ARG(a.workspace && a.workspace_bytes >= need, "gdl_conv_wgrad: workspace too small (%lld needed)", (long long)need);
  }
  int64_t per = (k.P + k.splits - 1) / k.splits;
  per = (per + bkp - 1) / bkp * bkp;
  k.p_per_split = per;
  k.x_dense = (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0 && a.H == a.Ho && a.W == a.Wo &&
               a.in_sH == (int64_t)a.W * a.in_sW && a.in_sB == (int64_t)a.H * a.in_sH);
  k.dy_dense = (a.dy_sH == (int64_t)a.Wo * a.dy_sW && a.dy_sB == (int64_t)a.Ho * a.dy_sH);
  hipStream_t s = (hipStream_t)stream;
  GDL_CHECK_ARG((int64_t)a.nz * k.splits <= 65535, "gdl_conv_wgrad: nz * splits too large");
  dim3 grid((a.N + 127) / 128, a.R * a.S * k.ctiles, a.nz * k.splits);
  const size_t lds = 2 * (128 + 128) * 128;
  if (wgrad_rows_ok(a) && !g_wgrad_force_v1) {
    WRows kr;
    kr.w = k;
    kr.in_span = (unsigned)span_bytes(a.B, a.H, a.W, a.C, a.in_sB, a.in_sH, a.in_sW, 2);
    kr.dy_span = (unsigned)span_bytes(a.B, a.Ho, a.Wo, a.N, a.dy_sB, a.dy_sH, a.dy_sW, 2);
    kr.ntiles = (a.N + 63) / 64;
    kr.cchunks = (a.C + 63) / 64;
    kr.seglen = rows_seglen(a.W);
    kr.segs_per_row = (a.W + kr.seglen - 1) / kr.seglen;
    kr.nsegs = a.B * a.H * kr.segs_per_row;
    kr.segs_per_split = (kr.nsegs + k.splits - 1) / k.splits;
    kr.xcd_group = g_wgrad_rows_xcd && k.splits >= 8 ? 1 : 0;
    const unsigned gy = kr.xcd_group ? (unsigned)((k.splits + 7) / 8 * 8) : (unsigned)k.splits;
    GDL_SET_MAX_LDS_ONCE(wgrad_rows_kernel, 2 * 35840);
    hipLaunchKernelGGL(wgrad_rows_kernel, dim3(kr.ntiles * kr.cchunks, gy), dim3(256), 2 * 35840, s, kr);
  } else if (wgrad_tile(a) == 256 && !g_wgrad_force_v1) {
    W256 kb;
    kb.w = k;
    kb.w.ctiles = (a.C + 255) / 256;
    kb.in_span = (unsigned)span_bytes(a.B, a.H, a.W, a.C, a.in_sB, a.in_sH, a.in_sW, 2);
    kb.dy_span = (unsigned)span_bytes(a.B, a.Ho, a.Wo, a.N, a.dy_sB, a.dy_sH, a.dy_sW, 2);
    GDL_SET_MAX_LDS_ONCE(wgrad_tr256_kernel, 128 * 1024);
    kb.tiles_n = (a.N + 255) / 256;
    kb.tiles_y = a.R * a.S * kb.w.ctiles;
    kb.xcd_group = g_wgrad_xcd_group;
    dim3 gridb((unsigned)((int64_t)kb.tiles_n * kb.tiles_y * a.nz * k.splits));
    hipLaunchKernelGGL(wgrad_tr256_kernel, gridb, dim3(512), 128 * 1024, s, kb);
  } else if (a.dtype == GDL_BF16 && !g_wgrad_force_v1) hipLaunchKernelGGL(wgrad_tr_kernel, grid, dim3(256), lds, s, k);
  else if (a.dtype == GDL_BF16) hipLaunchKernelGGL(wgrad_kernel<bf16_tag>, grid, dim3(256), lds, s, k);
  else hipLaunchKernelGGL(wgrad_kernel<float>, grid, dim3(256), lds, s, k);
  if (k.splits > 1) {
    const int64_t K = (int64_t)a.R * a.S * a.C;
    int64_t g = ((int64_t)a.N * K + 255) / 256;
    if (g > 8192) g = 8192;
    if (k.splits >= 16 && a.nz == 1)
      hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)(((int64_t)a.N * K + 31) / 32)), dim3(256), 0, s,
                         a.workspace, k.splits, a.N, K, a.dw, a.dw_sN, a.accumulate);
    else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g, a.nz), dim3(256), 0, s, a.workspace, k.splits, a.N, K,
                       a.dw, a.dw_sN, a.accumulate, a.nz_inner, a.dw_sZ0, a.dw_sZ1);
  }
  GDL_CHECK_LAUNCH("gdl_conv_wgrad");
  return GDL_OK;
}
