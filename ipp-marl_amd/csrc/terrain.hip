// Synthetic random-field terrain: the two device ends of the spectral synthesis the reference runs for every
// episode (mapping/ground_truths.py:16-40: white noise -> FFT -> sqrt(P(k)) -> inverse FFT -> min/max normalise
// -> >= 0.5).  The FFTs between the two kernels are rocFFT calls made by the host side (ippmarl/terrain.py).
#include "ippm_internal.h"

#define IPPM_DOMAIN_TERRAIN 3u
#ifndef IPPM_TERRAIN_ABL   // variant builds: timing ablations (1: no spectrum draw, 2 / 4: no first / second register FFT, 8: no stores)
#define IPPM_TERRAIN_ABL 0
#endif

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Standard-normal white noise, four cells per Philox block (two Box-Muller pairs).  The stream depends on
// (seed, episode, cell) only, so an episode's terrain does not depend on the batch it is generated in.
__global__ __launch_bounds__(256) void k_terrain_noise(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
                                                       float* __restrict__ noise, size_t cells) {
  const int e = blockIdx.y;
  const uint64_t ep = (uint64_t)episode[e];
  const uint32_t k0 = (uint32_t)c->philox_seed, k1 = (uint32_t)(c->philox_seed >> 32);
  const uint32_t sw = ippm_stream_word(0u, 0u, IPPM_DOMAIN_TERRAIN);
  float* out = noise + (size_t)e * cells;
  const size_t groups = (cells + 3) >> 2;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
    Philox4 ph = ippm_philox((uint32_t)g, (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // u1 in (0,1], u2 in [0,1)
      const float u1 = ((float)(ph.v[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);
      const float u2 = (float)(ph.v[2 * h + 1] >> 8) * (1.0f / 16777216.0f);
      const float r = sqrtf(-2.0f * logf(u1));
      float s, co;
      sincospif(2.0f * u2, &s, &co);
      z[2 * h] = r * co;
      z[2 * h + 1] = r * s;
    }
    const size_t base = g << 2;
    if (base + 3 < cells && (cells & 3) == 0) {
      *reinterpret_cast<float4*>(out + base) = make_float4(z[0], z[1], z[2], z[3]);
    } else {
      for (int q = 0; q < 4 && base + q < cells; ++q) out[base + q] = z[q];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Power-of-two grids: the spectrum is drawn directly (the FFT of real white noise is Hermitian complex white noise, so
// the forward transform of ground_truths.py:26 is not needed) and the inverse transform is done here, in two passes
// through LDS, because rocFFT's batched 2-D real transforms run at ~0.5 TB/s on this shape (tools/terrain_probe.py:
// 446 + 626 us for 1024 fields of 256 x 256 against ~150 us for the two passes below).
//   pass X: one workgroup per (env, tile of CW ky-columns): bins -> LDS (bit-reversed kx) -> radix-2 DIT along x ->
//           T[e, ky, x] (x contiguous: fully coalesced writes)
//   pass Y: one workgroup per (env, XW x-rows): T[e, 0..gy/2, x] -> LDS rows, Hermitian-extended to gy bins
//           (bit-reversed ky) -> radix-2 DIT along y -> real part -> field[e, x, y]
// Only the half spectrum ky in [0, gy/2] is ever stored (the real field's other half is its conjugate mirror).

// Complex product with the contraction FIXED (one product rounded on its own, the other fused into the sum): left to the compiler,
// which of the two products of a component goes into the fma differed between instantiations of one kernel -- pass Y's (min, max)
// launch and its threshold launch disagreed in the last bit of a row now and then (one truth bit in a few hundred fields, found when
// the one-launch form was compared with them in round 6).  Every form of the passes now computes the same bits.
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
#pragma clang fp contract(off)
  const float p = a.y * b.y, q = a.y * b.x;
  return make_float2(__builtin_fmaf(a.x, b.x, -p), __builtin_fmaf(a.x, b.y, q));
}
__device__ __forceinline__ uint32_t bitrev(uint32_t v, int bits) { return __brev(v) >> (32 - bits); }

// Bin (kx, ky) of the half spectrum of N(0,1) white noise, up to a common factor: generic bins are a + ib with
// a, b ~ N(0,1); on the self-mirrored columns ky in {0, gy/2} bin (gx - kx) is the conjugate of bin kx and the four
// self-conjugate bins are real with twice the variance.
//
// Randomness: one Philox call serves TWO bins -- bins kx and kx + gx/2 of a column share call number (kx mod gx/2) * (gy/2 + 1) + ky,
// the lower one takes words (0, 1), the upper one words (2, 3) -- because the call is most of a bin's cost (45 of ~60 instructions)
// and pass X is issue-bound; the pairing is by wave number, not by thread, so any kernel geometry draws the same spectrum (a
// thread of pass X holds kx = i1 * N2 + i2 for all i1: both partners).
__device__ __forceinline__ uint32_t terrain_call(int gx, int gy, int cx, int ky) {   // (cx: the bin whose numbers are drawn)
  return (uint32_t)(cx & (gx / 2 - 1)) * (uint32_t)(gy / 2 + 1) + (uint32_t)ky;
}
__device__ __forceinline__ float2 terrain_bin_from(uint32_t w1, uint32_t w2, bool self, bool conj, float amp) {
  const float u1 = ((float)(w1 >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(w2 >> 8) * (1.0f / 16777216.0f);
  // hardware transcendentals (v_log_f32, v_sin_f32 / v_cos_f32 take revolutions): a few ulp off the libm forms, which a
  // noise generator does not care about -- what is checked is the transform of whatever spectrum is drawn
  const float r = __builtin_sqrtf(-1.3862943611f * __builtin_amdgcn_logf(u1));   // -2 ln u = -2 ln2 log2 u
  const float sn = __builtin_amdgcn_sinf(u2), cs = __builtin_amdgcn_cosf(u2);
  if (self) return make_float2(amp * 1.41421356237f * r * cs, 0.0f);
  return make_float2(amp * r * cs, conj ? -amp * r * sn : amp * r * sn);
}
// one bin on its own (the spectrum kernel of the tests; pass X draws pairs, below)
__device__ __forceinline__ float2 terrain_bin(uint64_t ep, uint32_t k0, uint32_t k1, int gx, int gy, int kx, int ky, float amp) {
  const bool edge = ky == 0 || 2 * ky == gy;
  int cx = kx;
  bool conj = false;
  if (edge && 2 * kx > gx) { cx = gx - kx; conj = true; }
  const bool self = edge && (cx == 0 || 2 * cx == gx);
  Philox4 ph = ippm_philox(terrain_call(gx, gy, cx, ky), (uint32_t)ep, ippm_stream_word(0u, 1u, IPPM_DOMAIN_TERRAIN), (uint32_t)(ep >> 32), k0, k1);
  const bool upper = 2 * cx >= gx;
  return terrain_bin_from(upper ? ph.v[2] : ph.v[0], upper ? ph.v[3] : ph.v[1], self, conj, amp);
}
// bins kx (< gx/2) and kx + gx/2 of a GENERIC column ky (0 < ky < gy/2) from one call.  (On the self-mirrored columns ky in
// {0, gy/2} the upper bin is the conjugate of bin gx/2 - kx, which belongs to another call: pass X gives those two columns a
// workgroup of their own that draws bin by bin -- inside a generic workgroup their one lane in sixteen made every wavefront
// run both forms.)
__device__ __forceinline__ void terrain_bin_pair(uint64_t ep, uint32_t k0, uint32_t k1, int gx, int gy, int kx, int ky, float amp_lo,
                                                 float amp_hi, float2& lo, float2& hi) {
  Philox4 ph = ippm_philox(terrain_call(gx, gy, kx, ky), (uint32_t)ep, ippm_stream_word(0u, 1u, IPPM_DOMAIN_TERRAIN), (uint32_t)(ep >> 32), k0, k1);
  lo = terrain_bin_from(ph.v[0], ph.v[1], false, false, amp_lo);
  hi = terrain_bin_from(ph.v[2], ph.v[3], false, false, amp_hi);
}

__global__ __launch_bounds__(256) void k_terrain_spectrum(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
                                                          const float* __restrict__ amp, float2* __restrict__ spec) {
  const int e = blockIdx.y, gx = c->grid_x, gy = c->grid_y, hy = gy / 2 + 1;
  const uint64_t ep = (uint64_t)episode[e];
  const uint32_t k0 = (uint32_t)c->philox_seed, k1 = (uint32_t)(c->philox_seed >> 32);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < gx * hy; i += gridDim.x * blockDim.x)
    spec[(size_t)e * gx * hy + i] = terrain_bin(ep, k0, k1, gx, gy, i / hy, i % hy, amp[i]);
}

// ---- register-resident FFT pieces ----------------------------------------------------------------------------
__device__ constexpr float TWC[16] = {1.000000000f, 0.980785280f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f,
                                      0.382683432f, 0.195090322f, 0.000000000f, -0.195090322f, -0.382683432f, -0.555570233f,
                                      -0.707106781f, -0.831469612f, -0.923879533f, -0.980785280f};
__device__ constexpr float TWS[16] = {0.000000000f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f,
                                      0.923879533f, 0.980785280f, 1.000000000f, 0.980785280f, 0.923879533f, 0.831469612f,
                                      0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f};

// inverse (e^{+i..}) FFT of R = 8/16/32 points held in registers, natural order in and out; every index is a
// compile-time constant after unrolling, so the array never leaves the VGPRs
template <int R>
__device__ __forceinline__ void fft_reg(float2 (&a)[R]) {
  constexpr int LOG = R == 8 ? 3 : (R == 16 ? 4 : 5);
  static_assert((1 << LOG) == R, "fft_reg: R must be 8, 16 or 32");
#pragma unroll
  for (int i = 0; i < R; ++i) {
    int j = 0;
#pragma unroll
    for (int b = 0; b < LOG; ++b) j |= ((i >> b) & 1) << (LOG - 1 - b);
    if (i < j) {
      const float2 t = a[i];
      a[i] = a[j];
      a[j] = t;
    }
  }
#pragma unroll
  for (int s = 0; s < LOG; ++s) {
#pragma unroll
    for (int j = 0; j < R / 2; ++j) {
      const int half = 1 << s, pos = j & (half - 1), i0 = ((j >> s) << (s + 1)) + pos, i1 = i0 + half, t = pos * (16 >> s);
      const float2 u = a[i0], v = cmul(a[i1], make_float2(TWC[t], TWS[t]));
      a[i0] = make_float2(u.x + v.x, u.y + v.y);
      a[i1] = make_float2(u.x - v.x, u.y - v.y);
    }
  }
}

// Four-step FFT of Q sequences of n = N1 * N2 points per workgroup.  In: thread (q_in, i2) holds x[i1 * N2 + i2],
// i1 = 0..N1-1.  N1-point transforms in registers, twiddle by W_n^(i2 k1), one exchange through LDS, N2-point
// transforms in registers.  Out: thread (k1, q_out) holds X[k1 + N1 * k2], k2 = 0..N2-1.
template <int N1, int N2, int Q>
struct FourStep {
  static constexpr int SEQ = N1 * (N2 + 1) + 1;  // padded so that neither side of the exchange piles onto few LDS banks
  static constexpr int THREADS = Q * (N1 > N2 ? N1 : N2);
  float2 xbuf[Q * SEQ];
  float2 twn[N1 * N2];

  // W_n^i = e^{2 pi i / n} from the context's table of the 1024-th roots of unity (float64 on the host, rounded once): a
  // sincospif per thread and workgroup was 60 instructions in front of every workgroup's first load
  __device__ __forceinline__ void init_twiddles(const float2* __restrict__ roots1024) {
    for (int i = threadIdx.x; i < N1 * N2; i += THREADS) twn[i] = roots1024[i * (1024 / (N1 * N2))];
    __syncthreads();
  }
  __device__ __forceinline__ void run(float2 (&v)[N1], float2 (&o)[N2], int q_in, int i2, bool in_active, int q_out, int k1,
                                      bool out_active) {
    if (in_active) {
#if !(IPPM_TERRAIN_ABL & 2)
      fft_reg<N1>(v);
#endif
#pragma unroll
      for (int k = 0; k < N1; ++k) xbuf[q_in * SEQ + k * (N2 + 1) + i2] = cmul(v[k], twn[i2 * k]);
    }
    __syncthreads();
    if (out_active) {
#pragma unroll
      for (int i = 0; i < N2; ++i) o[i] = xbuf[q_out * SEQ + k1 * (N2 + 1) + i];
#if !(IPPM_TERRAIN_ABL & 4)
      fft_reg<N2>(o);
#endif
    }
  }
};

// float <-> uint32 key with the same ordering, so that min/max of floats can use integer atomics
__device__ __forceinline__ uint32_t order_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// pass X: sequences are the ky columns of the half spectrum (length gx = N1 * N2), Q columns per workgroup.  Workgroups
// 0 .. gridDim.x - 2 take the generic columns 1 .. gy/2 - 1, the last one the two self-mirrored columns 0 and gy/2.
template <int N1, int N2, int Q, bool GEN>
__global__ __launch_bounds__((FourStep<N1, N2, Q>::THREADS)) void k_terrain_fft_x(const ippm_config* __restrict__ c,
                                                                               const int64_t* __restrict__ episode,
                                                                               const float* __restrict__ amp,
                                                                               const float2* __restrict__ spec,
                                                                               float2* __restrict__ work,
                                                                               uint32_t* __restrict__ range_keys,
                                                                               const float2* __restrict__ roots1024, int key_stride) {
  __shared__ FourStep<N1, N2, Q> fs;
  const int e = blockIdx.y, gx = N1 * N2, gy = c->grid_y, hy = gy / 2 + 1, tid = threadIdx.x;
  const bool edge_wg = blockIdx.x == gridDim.x - 1;
  if (range_keys && blockIdx.x == 0 && tid == 0) {   // pass Y accumulates the field's (min, max) here
    range_keys[key_stride * e] = 0xFFFFFFFFu;
    range_keys[key_stride * e + 1] = 0u;
    if (key_stride == 4) {    // the one-launch form of pass Y (MODE 3): its workgroups' arrivals per env, the env's fault word, and
      range_keys[4 * e + 2] = 0u;                    // (behind the last env's record) the launch's ticket counter
      range_keys[4 * e + 3] = 0u;
      if (e == 0) range_keys[4 * gridDim.y] = 0u;
    }
  }
  fs.init_twiddles(roots1024);
  const int q_in = tid % Q, i2 = tid / Q, k1 = tid % N1, q_out = tid / N1;
  const bool in_active = i2 < N2, out_active = q_out < Q;
  // column of slot q: -1 = none
  auto column = [&](int q) { return edge_wg ? (q == 0 ? 0 : (q == 1 ? gy / 2 : -1)) : (1 + (int)blockIdx.x * Q + q < gy / 2 ? 1 + (int)blockIdx.x * Q + q : -1); };
  float2 v[N1], o[N2];
  if (in_active) {
    const int ky = column(q_in);
    const uint64_t ep = GEN ? (uint64_t)episode[e] : 0;
    const uint32_t k0 = (uint32_t)c->philox_seed, k1s = (uint32_t)(c->philox_seed >> 32);
#pragma unroll
    for (int i1 = 0; i1 < N1; ++i1) v[i1] = make_float2(0.0f, 0.0f);
    if (ky >= 0) {
      if (GEN && edge_wg) {     // (workgroup-uniform) bin by bin: conjugate pairs and the four real bins
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1) {
          const int kx = i1 * N2 + i2;
          v[i1] = terrain_bin(ep, k0, k1s, gx, gy, kx, ky, amp[(size_t)kx * hy + ky]);
        }
      } else if (GEN) {
#pragma unroll
        for (int i1 = 0; i1 < N1 / 2; ++i1) {   // bins kx and kx + gx/2 come from one Philox call
          const int kx = i1 * N2 + i2;
          terrain_bin_pair(ep, k0, k1s, gx, gy, kx, ky, amp[(size_t)kx * hy + ky], amp[(size_t)(kx + gx / 2) * hy + ky], v[i1], v[i1 + N1 / 2]);
        }
      } else {
#pragma unroll
        for (int i1 = 0; i1 < N1; ++i1) v[i1] = spec[((size_t)e * gx + i1 * N2 + i2) * hy + ky];
      }
    }
  }
  fs.run(v, o, q_in, i2, in_active, q_out, k1, out_active);
  const int ky_out = out_active ? column(q_out) : -1;
  if (ky_out >= 0) {
    float2* dst = work + ((size_t)e * hy + ky_out) * gx + k1;
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) dst[N1 * k2] = o[k2];
  }
}

// pass Y: sequences are the x rows (length gy = N1 * N2 after the Hermitian extension).  The rows are real, so ONE complex
// transform yields TWO of them: with A, B the Hermitian-extended spectra of rows xa and xb, the inverse transform of A + iB is
// a + ib with a, b the two real rows.  A workgroup takes 2 Q rows, slot q the NEIGHBOURS xa = x0 + 2 q, xb = xa + 1 (their
// spectra are one 16-byte load per bin): half the transforms of the row-by-row form.
// MODE 0: the field is written and its (min, max) accumulated (ippm_terrain_field).  The episode reset never needs the field
// itself -- only its threshold bits, and those need the field's (min, max) first -- so it runs the pass twice instead of writing
// 268 MB and reading them back: MODE 1 accumulates (min, max) and stores nothing, MODE 2 recomputes the rows (bit for bit the
// same arithmetic) and writes the truth bits (f - min) / (max - min) >= 0.5 straight into the packed plane.
template <int N1, int N2, int Q, int MODE>
__global__ __launch_bounds__((FourStep<N1, N2, Q>::THREADS)) void k_terrain_fft_y(const ippm_config* __restrict__ c,
                                                                               const float2* __restrict__ work,
                                                                               float* __restrict__ field,
                                                                               uint32_t* __restrict__ range_keys,
                                                                               const float2* __restrict__ roots1024,
                                                                               uint32_t* __restrict__ truth32, int truth_words, int key_stride) {
  __shared__ FourStep<N1, N2, Q> fs;
  const int tid = threadIdx.x, gx = c->grid_x, gy = N1 * N2, hy = gy / 2 + 1;
  int e = blockIdx.y, part = blockIdx.x;
  __shared__ uint32_t s_ticket, s_keys[2];
  if (MODE == 3) {
    // One launch for (min, max) AND the threshold bits: the workgroups of an env meet at a counter once their rows are in
    // registers.  Waiting for a workgroup that has not started yet would hang the device, so which (env, rows) a workgroup takes
    // follows the ORDER IN WHICH WORKGROUPS START (a ticket), not its index: the tickets taken so far are exactly the workgroups
    // that have started, every env below the one in progress is complete and leaves, and its slots go to the next tickets --
    // whatever order the hardware dispatches in.  (The ticket's round trip runs beside the twiddle loads below.)
    if (tid == 0) s_ticket = atomicAdd(range_keys + 4 * gridDim.y, 1u);
  }
  fs.init_twiddles(roots1024);     // (ends with a workgroup barrier)
  if (MODE == 3) {
    const uint32_t tk = s_ticket;
    e = (int)(tk / gridDim.x);
    part = (int)(tk - (uint32_t)e * gridDim.x);
  }
  const int x0 = part * 2 * Q;
  const int q_in = tid % Q, i2 = tid / Q, k1 = tid % N1, q_out = tid / N1;
  const bool in_active = i2 < N2, out_active = q_out < Q;
  float2 v[N1], o[N2];
  if (in_active) {
    const float2* src = work + (size_t)e * hy * gx + x0 + 2 * q_in;   // (16-byte aligned: x0 and gx are even)
#pragma unroll
    for (int i1 = 0; i1 < N1; ++i1) {
      const int i = i1 * N2 + i2, m = 2 * i > gy ? gy - i : i;   // bins above gy/2 are the conjugate mirror
      const float4 ab = *reinterpret_cast<const float4*>(src + (size_t)m * gx);
      float2 a = make_float2(ab.x, ab.y), b = make_float2(ab.z, ab.w);
      if (2 * i > gy) { a.y = -a.y; b.y = -b.y; }
      if (i == 0 || 2 * i == gy) { a.y = 0.0f; b.y = 0.0f; }     // self-mirrored bins of a real transform
      v[i1] = make_float2(a.x - b.y, a.y + b.x);                 // A + iB
    }
  }
  fs.run(v, o, q_in, i2, in_active, q_out, k1, out_active);
  if (MODE != 2) {
    float lo = INFINITY, hi = -INFINITY;
    if (out_active) {
      float* dst = field + ((size_t)e * gx + x0 + 2 * q_out) * gy + k1;
#pragma unroll
      for (int k2 = 0; k2 < N2; ++k2) {
        if (MODE == 0) {
          dst[N1 * k2] = o[k2].x;
          dst[(size_t)gy + N1 * k2] = o[k2].y;
        }
        lo = fminf(lo, fminf(o[k2].x, o[k2].y));
        hi = fmaxf(hi, fmaxf(o[k2].x, o[k2].y));
      }
    }
    if (range_keys) {   // wavefront, then workgroup reduction; one atomic pair per workgroup
#pragma unroll
      for (int m = 32; m > 0; m >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, m, 64));
        hi = fmaxf(hi, __shfl_xor(hi, m, 64));
      }
      __shared__ float s_lo[8], s_hi[8];
      constexpr int NW = FourStep<N1, N2, Q>::THREADS / 64;
      if ((tid & 63) == 0) { s_lo[tid >> 6] = lo; s_hi[tid >> 6] = hi; }
      __syncthreads();
      if (tid == 0) {
        for (int w = 1; w < NW; ++w) { lo = fminf(lo, s_lo[w]); hi = fmaxf(hi, s_hi[w]); }
        if (MODE != 3) {
          atomicMin(range_keys + key_stride * e, order_key(lo));
          atomicMax(range_keys + key_stride * e + 1, order_key(hi));
        } else {
          // My (min, max) go in BEFORE my arrival does, and everything that crosses workgroups here is a device-scope atomic
          // read-modify-write, carried out where all XCDs see it -- no fence: an agent-scope release / acquire writes back and
          // invalidates the whole L2 of the XCD on gfx950, which made this launch 850 us instead of 80.  The two RETURNING atomics
          // are waited for (their results are operands of the wait) before the arrival is issued.
          uint32_t* rec = range_keys + 4 * e;
          const uint32_t o1 = __hip_atomic_fetch_min(rec, order_key(lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t o2 = __hip_atomic_fetch_max(rec + 1, order_key(hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" : : "v"(o1), "v"(o2) : "memory");
          __hip_atomic_fetch_add(rec + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // wait for the env's other workgroups (all of them have started: ticket order).  The spin is bounded all the same: a wait
          // that long means the premise broke -- word 3 of the env's record then says so (VecEnv.check_faults and the tests look at
          // it) and the workgroup goes on with what there is.
          // The counter is READ by a read-modify-write too (a compare-and-swap that never succeeds): an atomic LOAD at agent scope
          // (global_load sc1) can be served from a stale line of this XCD's L2 -- measured: a workgroup at 1024^2 waited out the
          // whole bound while its env's counter stood at 128 of 128, and at 256^2 the launch took 210 us for wake-ups that late.
          unsigned spins = 0;
          while (atomicCAS(rec + 2, 0xFFFFFFFFu, 0xFFFFFFFEu) < gridDim.x) {
            __builtin_amdgcn_s_sleep(64);      // (~1.7 us between two looks: the counter's address is shared with every peer)
            if (++spins > (1u << 21)) {
              atomicExch(rec + 3, 0xDEAD0000u | (unsigned)part);
              break;
            }
          }
          // the final keys, read where they were written (neither key can be all ones: that is a NaN's key)
          s_keys[0] = atomicCAS(rec, 0xFFFFFFFFu, 0xFFFFFFFEu);
          s_keys[1] = atomicCAS(rec + 1, 0xFFFFFFFFu, 0xFFFFFFFEu);
        }
      }
    }
    if (MODE != 3) return;
    __syncthreads();     // the env's (min, max) are final
  }
  {
    // threshold and pack: for one k2 a wavefront holds columns k1 + N1 k2 of 64 / N1 rows, i.e. N1 consecutive bits of each of
    // those rows' bit strings; lane L < 64 / N1 collects row L's chunks into 32-bit words and stores every completed word
    static_assert(N1 == 8 || N1 == 16 || N1 == 32, "a chunk must not straddle a 32-bit word");
    // (MODE 3: written by other workgroups of this launch, possibly on other XCDs -- thread 0 fetched them at device scope)
    const uint32_t key_lo = MODE == 3 ? s_keys[0] : range_keys[key_stride * e];
    const uint32_t key_hi = MODE == 3 ? s_keys[1] : range_keys[key_stride * e + 1];
    const float lo = key_value(key_lo), span = key_value(key_hi) - lo;
    const int lane = tid & 63, rows_per_wave = 64 / N1;
    const int my_q = (tid >> 6) * rows_per_wave + lane;          // the row (slot) this lane stores for, if lane < rows_per_wave
    const bool storer = lane < rows_per_wave && my_q < Q;
    uint32_t* out = truth32 + (size_t)e * truth_words;
    const size_t row_a = ((size_t)(x0 + 2 * my_q) * gy) >> 5, row_b = ((size_t)(x0 + 2 * my_q + 1) * gy) >> 5;
    uint32_t wa = 0, wb = 0;
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) {
      const uint64_t ba = __ballot(out_active && (o[k2].x - lo) / span >= 0.5f);
      const uint64_t bb = __ballot(out_active && (o[k2].y - lo) / span >= 0.5f);
      const uint32_t mask = N1 == 32 ? 0xFFFFFFFFu : ((1u << (N1 & 31)) - 1u);
      const int sh = (k2 * N1) & 31;
      wa |= ((uint32_t)(ba >> ((N1 * lane) & 63)) & mask) << sh;
      wb |= ((uint32_t)(bb >> ((N1 * lane) & 63)) & mask) << sh;
      if (((k2 + 1) * N1) % 32 == 0) {   // (compile time) a word is complete
        if (storer) { out[row_a + ((k2 * N1) >> 5)] = wa; out[row_b + ((k2 * N1) >> 5)] = wb; }
        wa = 0; wb = 0;
      }
    }
  }
}

// IPPM_PACK_PARTS workgroups per env: each one reduces min/max over the whole env field (256 KiB at 256 x 256: the
// repeats are L2 hits) and then writes its share of truth bits, bit = (f - min)/(max - min) >= 0.5, packed 64 cells
// per wavefront ballot in the library's truth layout (bit lin & 7 of byte lin >> 3).
#define IPPM_PACK_PARTS 4
__global__ __launch_bounds__(1024) void k_terrain_pack(const float* __restrict__ field, uint8_t* __restrict__ truth, size_t cells,
                                                       size_t truth_bytes) {
  __shared__ float s_lo[16], s_hi[16];
  const int e = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  const float* f = field + (size_t)e * cells;
  float lo = INFINITY, hi = -INFINITY;
  if ((cells & 3) == 0) {
    const float4* f4 = reinterpret_cast<const float4*>(f);
    const size_t n4 = cells >> 2;
#pragma unroll 4
    for (size_t i = tid; i < n4; i += blockDim.x) {
      const float4 v = f4[i];
      lo = fminf(fminf(lo, fminf(v.x, v.y)), fminf(v.z, v.w));
      hi = fmaxf(fmaxf(hi, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
  } else {
    for (size_t i = tid; i < cells; i += blockDim.x) {
      const float v = f[i];
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; }
  __syncthreads();
  lo = s_lo[0];
  hi = s_hi[0];
  for (int w = 1; w < nwave; ++w) {
    lo = fminf(lo, s_lo[w]);
    hi = fmaxf(hi, s_hi[w]);
  }
  const float span = hi - lo;
  uint32_t* out = reinterpret_cast<uint32_t*>(truth + (size_t)e * truth_bytes);  // truth_bytes is a multiple of 4
  const size_t words = truth_bytes >> 2;
  const size_t chunks = (cells + 63) >> 6;
  for (size_t ch = (size_t)blockIdx.x * nwave + wave; ch < chunks; ch += (size_t)gridDim.x * nwave) {
    const size_t i = (ch << 6) + lane;
    const bool one = i < cells && (f[i] - lo) / span >= 0.5f;
    const uint64_t bits = __ballot(one);
    if (lane < 2 && 2 * ch + lane < words) out[2 * ch + lane] = (uint32_t)(bits >> (32 * lane));
  }
}

// same thresholding with the (min, max) already known (ippm_terrain_field's range_keys): one flat pass
__global__ __launch_bounds__(256) void k_terrain_pack_keys(const float* __restrict__ field, const uint32_t* __restrict__ range_keys,
                                                           uint8_t* __restrict__ truth, size_t cells, size_t truth_bytes) {
  const int e = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const float* f = field + (size_t)e * cells;
  const float lo = key_value(range_keys[2 * e]), span = key_value(range_keys[2 * e + 1]) - lo;
  uint32_t* out = reinterpret_cast<uint32_t*>(truth + (size_t)e * truth_bytes);
  const size_t words = truth_bytes >> 2, chunks = (cells + 63) >> 6;
  const size_t stride = (size_t)gridDim.x * nwave;
  for (size_t ch0 = (size_t)blockIdx.x * nwave + wave; ch0 < chunks; ch0 += 4 * stride) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // four independent loads in flight before the first ballot
      const size_t i = ((ch0 + u * stride) << 6) + lane;
      v[u] = i < cells ? f[i] : lo;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t ch = ch0 + u * stride;
      const uint64_t bits = __ballot(((ch << 6) + lane) < cells && (v[u] - lo) / span >= 0.5f);
      if (ch < chunks && lane < 2 && 2 * ch + lane < words) out[2 * ch + lane] = (uint32_t)(bits >> (32 * lane));
    }
  }
}

// the same for fields whose cell count is a multiple of 4: one 16-byte non-temporal load per lane, one trip per workgroup
// (the shape that streams best on this device, tools/probe/copy_probe.cpp); 8 lanes assemble one 32-bit word of truth bits
typedef float terrain_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_terrain_pack_keys4(const float* __restrict__ field, const uint32_t* __restrict__ range_keys,
                                                            uint8_t* __restrict__ truth, size_t cells, size_t truth_bytes) {
  const int e = blockIdx.y;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;   // 4-cell group of this env
  const float lo = key_value(range_keys[2 * e]), span = key_value(range_keys[2 * e + 1]) - lo;
  uint32_t nib = 0;
  if (4 * g < cells) {
    const terrain_f4 v = __builtin_nontemporal_load(reinterpret_cast<const terrain_f4*>(field + (size_t)e * cells) + g);
    nib = ((v.x - lo) / span >= 0.5f ? 1u : 0u) | ((v.y - lo) / span >= 0.5f ? 2u : 0u) | ((v.z - lo) / span >= 0.5f ? 4u : 0u) |
          ((v.w - lo) / span >= 0.5f ? 8u : 0u);
  }
  uint32_t w = nib << (4 * (threadIdx.x & 7));
  w |= __shfl_xor(w, 1, 64);
  w |= __shfl_xor(w, 2, 64);
  w |= __shfl_xor(w, 4, 64);
  const size_t word = g >> 3;
  if ((threadIdx.x & 7) == 0 && word < (truth_bytes >> 2)) reinterpret_cast<uint32_t*>(truth + (size_t)e * truth_bytes)[word] = w;
}

extern "C" int ippm_terrain_noise(ippm_ctx* ctx, const int64_t* episode, float* noise, int32_t n_envs, void* stream) {
  if (!ctx || !episode || !noise) { ippm_set_error("ippm_terrain_noise: null argument"); return -1; }
  if (n_envs <= 0) return 0;
  const size_t cells = (size_t)ctx->cfg.grid_x * ctx->cfg.grid_y;
  const int gx = (int)std::min<size_t>(64, (cells / 4 + 255) / 256);
  hipLaunchKernelGGL(k_terrain_noise, dim3(gx > 0 ? gx : 1, n_envs), dim3(256), 0, S_(stream), ctx->dcfg, episode, noise, cells);
  IPPM_LAUNCH_CHECK("terrain_noise");
  return 0;
}

extern "C" int ippm_terrain_pack(ippm_ctx* ctx, const float* field, const uint32_t* range_keys, uint8_t* truth, int32_t n_envs,
                                 void* stream) {
  if (!ctx || !field || !truth) { ippm_set_error("ippm_terrain_pack: null argument"); return -1; }
  if (n_envs <= 0) return 0;
  const size_t cells = (size_t)ctx->cfg.grid_x * ctx->cfg.grid_y;
  if (range_keys && cells % 4 == 0 && (reinterpret_cast<uintptr_t>(field) & 15) == 0) {
    IPPM_LAUNCH(ctx, IPPM_T_TERRAIN, k_terrain_pack_keys4, dim3((unsigned)((cells / 4 + 255) / 256), n_envs), dim3(256), S_(stream), field,
                       range_keys, truth, cells, ippm_truth_bytes(ctx->cfg.grid_x, ctx->cfg.grid_y));
    IPPM_LAUNCH_CHECK("terrain_pack_keys4");
    return 0;
  }
  if (range_keys) {
    const int gxb = (int)std::min<size_t>(64, ((cells + 63) / 64 + 15) / 16);
    hipLaunchKernelGGL(k_terrain_pack_keys, dim3(gxb > 0 ? gxb : 1, n_envs), dim3(256), 0, S_(stream), field, range_keys, truth, cells,
                       ippm_truth_bytes(ctx->cfg.grid_x, ctx->cfg.grid_y));
    IPPM_LAUNCH_CHECK("terrain_pack_keys");
    return 0;
  }
  hipLaunchKernelGGL(k_terrain_pack, dim3(IPPM_PACK_PARTS, n_envs), dim3(1024), 0, S_(stream), field, truth, cells,
                     ippm_truth_bytes(ctx->cfg.grid_x, ctx->cfg.grid_y));
  IPPM_LAUNCH_CHECK("terrain_pack");
  return 0;
}

static bool terrain_side_ok(int n) { return n == 128 || n == 256 || n == 512 || n == 1024; }

static int terrain_pow2_check(const ippm_ctx* ctx, const char* who) {
  if (!terrain_side_ok(ctx->cfg.grid_x) || !terrain_side_ok(ctx->cfg.grid_y)) {
    ippm_set_error(std::string(who) + ": grid sides must be 128, 256, 512 or 1024 (use ippm_terrain_noise + an FFT library otherwise)");
    return -1;
  }
  return 0;
}

// n = N1 * N2 split and sequences per workgroup for each supported side
#define TERRAIN_DISPATCH(n, CALL)                \
  switch (n) {                                   \
    case 128: { CALL(8, 16, 16); break; }        \
    case 256: { CALL(16, 16, 16); break; }       \
    case 512: { CALL(16, 32, 8); break; }        \
    default: { CALL(32, 32, 4); break; }         \
  }

extern "C" int ippm_terrain_spectrum(ippm_ctx* ctx, const int64_t* episode, const float* amp, float* spec, int32_t n_envs,
                                     void* stream) {
  if (!ctx || !episode || !amp || !spec) { ippm_set_error("ippm_terrain_spectrum: null argument"); return -1; }
  if (terrain_pow2_check(ctx, "ippm_terrain_spectrum")) return -1;
  if (n_envs <= 0) return 0;
  const int bins = ctx->cfg.grid_x * (ctx->cfg.grid_y / 2 + 1);
  hipLaunchKernelGGL(k_terrain_spectrum, dim3(std::min(64, (bins + 255) / 256), n_envs), dim3(256), 0, S_(stream), ctx->dcfg,
                     episode, amp, reinterpret_cast<float2*>(spec));
  IPPM_LAUNCH_CHECK("terrain_spectrum");
  return 0;
}

static int terrain_launch_x(ippm_ctx* ctx, const int64_t* episode, const float* amp, const float2* spec2, float2* work2,
                            uint32_t* range_keys, int n_envs, hipStream_t st, int key_stride = 2) {
  const int gx = ctx->cfg.grid_x, gy = ctx->cfg.grid_y;
  // generic columns 1 .. gy/2 - 1 in workgroups of Q, then one workgroup for the self-mirrored columns 0 and gy/2
#define LAUNCH_X(N1, N2, Q)                                                                                                  \
  {                                                                                                                          \
    const dim3 grid((gy / 2 - 1 + Q - 1) / Q + 1, n_envs), block(FourStep<N1, N2, Q>::THREADS);                              \
    if (spec2) IPPM_LAUNCH(ctx, IPPM_T_TERRAIN, (k_terrain_fft_x<N1, N2, Q, false>), grid, block, st, ctx->dcfg, episode, amp, spec2, work2, \
                           range_keys, ctx->d_roots, key_stride);                                                            \
    else IPPM_LAUNCH(ctx, IPPM_T_TERRAIN, (k_terrain_fft_x<N1, N2, Q, true>), grid, block, st, ctx->dcfg, episode, amp, spec2, work2,        \
                     range_keys, ctx->d_roots, key_stride);                                                                  \
  }
  TERRAIN_DISPATCH(gx, LAUNCH_X)
#undef LAUNCH_X
  IPPM_LAUNCH_CHECK("terrain_fft_x");
  return 0;
}

template <int MODE>
static int terrain_launch_y(ippm_ctx* ctx, const float2* work2, float* field, uint32_t* range_keys, uint8_t* truth, int n_envs,
                            hipStream_t st, int key_stride = 2) {
  const int gx = ctx->cfg.grid_x, gy = ctx->cfg.grid_y;
#define LAUNCH_Y(N1, N2, Q)                                                                                                  \
  IPPM_LAUNCH(ctx, IPPM_T_TERRAIN, (k_terrain_fft_y<N1, N2, Q, MODE>), dim3(gx / (2 * Q), n_envs), dim3(FourStep<N1, N2, Q>::THREADS), st, \
              ctx->dcfg, work2, field, range_keys, ctx->d_roots, reinterpret_cast<uint32_t*>(truth), (int)(ippm_truth_bytes(gx, gy) >> 2),       \
              MODE == 3 ? 4 : key_stride)
  TERRAIN_DISPATCH(gy, LAUNCH_Y)
#undef LAUNCH_Y
  IPPM_LAUNCH_CHECK("terrain_fft_y");
  return 0;
}

extern "C" int ippm_terrain_field(ippm_ctx* ctx, const int64_t* episode, const float* amp, const float* spec, float* work,
                                  float* field, uint32_t* range_keys, int32_t n_envs, void* stream) {
  if (!ctx || !work || !field) { ippm_set_error("ippm_terrain_field: null argument"); return -1; }
  if (!spec && (!episode || !amp)) { ippm_set_error("ippm_terrain_field: needs either spec or (episode, amp)"); return -1; }
  if (terrain_pow2_check(ctx, "ippm_terrain_field")) return -1;
  if (n_envs <= 0) return 0;
  if (int rc = terrain_launch_x(ctx, episode, amp, reinterpret_cast<const float2*>(spec), reinterpret_cast<float2*>(work), range_keys, n_envs,
                                S_(stream))) return rc;
  return terrain_launch_y<0>(ctx, reinterpret_cast<const float2*>(work), field, range_keys, nullptr, n_envs, S_(stream));
}

// The episode reset's form: spectrum drawn in pass X; the field itself is never stored.  Its threshold bits need its (min, max)
// first, so pass Y runs twice: once for the (min, max), once more (bit for bit the same transforms) for the bits.
// IPPM_TERRAIN_ONE_LAUNCH=1 (round 6, measured, NOT the default): every row computed ONCE -- the workgroups of an env put their
// (min, max) in, meet at a counter with their rows in registers and threshold them (MODE 3).  It saves the second read of the half
// spectrum (270 MB per 1024 fields of 256^2) and the second set of transforms, and gives most of that back to workgroups that sit
// on their CU waiting for seven peers: 124 us against 67 + 71 for the pass, 212 against 223 us for a whole reset's terrain, and no
// gain in the step (profiles/r06/terrain_one_launch_ab.txt).  Same truth bit for bit (tested at 3000 x 256^2, 700 x 512^2, 501 x 128^2).
extern "C" int ippm_terrain_truth(ippm_ctx* ctx, const int64_t* episode, const float* amp, float* work, uint32_t* range_keys,
                                  uint8_t* truth, int32_t n_envs, void* stream) {
  if (!ctx || !episode || !amp || !work || !range_keys || !truth) { ippm_set_error("ippm_terrain_truth: null argument"); return -1; }
  if (terrain_pow2_check(ctx, "ippm_terrain_truth")) return -1;
  if (n_envs <= 0) return 0;
  if (int rc = terrain_launch_x(ctx, episode, amp, nullptr, reinterpret_cast<float2*>(work), range_keys, n_envs, S_(stream), 4)) return rc;
  // (at most 32 workgroups per env, i.e. fields up to 512 cells a side: with the 128 of a 1024^2 field the waiters' compare-and-swaps on
  // one counter come close to what one address takes (~12 ns each) and a wait was seen to run into its bound once in a few runs)
  if (ctx->knob_terrain_one_launch && ctx->cfg.grid_x <= 512)
    return terrain_launch_y<3>(ctx, reinterpret_cast<const float2*>(work), nullptr, range_keys, truth, n_envs, S_(stream), 4);
  if (int rc = terrain_launch_y<1>(ctx, reinterpret_cast<const float2*>(work), nullptr, range_keys, nullptr, n_envs, S_(stream), 4)) return rc;
  return terrain_launch_y<2>(ctx, reinterpret_cast<const float2*>(work), nullptr, range_keys, truth, n_envs, S_(stream), 4);
}
