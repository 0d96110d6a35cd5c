// Issue cost of the instructions the Philox rounds are made of, relative to a plain 32-bit VALU op (gfx950):
// hipcc --offload-arch=gfx950 -O3 valu_rate_probe.cpp -o valu_rate_probe && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned seed, int iters) {
  unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, b0 = a0 ^ 0x55u, b1 = a1 ^ 0x77u, b2 = a2 + 9u, b3 = a3 + 11u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {   // 8 independent chains: issue-bound, not latency-bound
      if (KIND == 0) { a0 = (a0 ^ b0) + 0x9E3779B9u; a1 = (a1 ^ b1) + 0xBB67AE85u; a2 = (a2 ^ b2) + 1u; a3 = (a3 ^ b3) + 3u;
                       b0 = (b0 ^ a1) + 5u; b1 = (b1 ^ a2) + 7u; b2 = (b2 ^ a3) + 9u; b3 = (b3 ^ a0) + 11u; }          // 16 plain ops
      if (KIND == 1) { unsigned long long p0 = (unsigned long long)0xD2511F53u * a0, p1 = (unsigned long long)0xCD9E8D57u * a1,
                                          p2 = (unsigned long long)0xD2511F53u * a2, p3 = (unsigned long long)0xCD9E8D57u * a3;
                       a0 = (unsigned)(p0 >> 32) ^ b0; b0 = (unsigned)p0; a1 = (unsigned)(p1 >> 32) ^ b1; b1 = (unsigned)p1;
                       a2 = (unsigned)(p2 >> 32) ^ b2; b2 = (unsigned)p2; a3 = (unsigned)(p3 >> 32) ^ b3; b3 = (unsigned)p3; }  // 4 wide muls + 4 xor
      if (KIND == 2) { float f0 = __uint_as_float((a0 & 0x007FFFFFu) | 0x3F000000u), f1 = __uint_as_float((a1 & 0x007FFFFFu) | 0x3F000000u);
                       a0 += __float_as_uint(__builtin_amdgcn_exp2f(f0)); a1 += __float_as_uint(__builtin_amdgcn_logf(f1));
                       a2 += __float_as_uint(__builtin_amdgcn_rcpf(__uint_as_float((a2 & 0x007FFFFFu) | 0x3F000000u)));
                       a3 += __float_as_uint(__builtin_amdgcn_exp2f(__uint_as_float((a3 & 0x007FFFFFu) | 0x3F000000u))); }   // 4 transcendentals + ~10 plain
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
}
template <int KIND> float run(unsigned* d, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<KIND><<<256 * 8, 256>>>(d, 1, 4);
  hipEventRecord(a);
  k<KIND><<<256 * 8, 256>>>(d, 2, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 2000;
  const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters);
  // per wavefront-instruction: 2048 workgroups x 4 wavefronts / 1024 SIMDs = 8 wavefronts per SIMD
  const double per_simd_iters = 8.0 * iters * 16;
  printf("plain: %.3f ms = %.2f ns per 16 plain ops per wavefront\\n", t0, t0 * 1e6 / per_simd_iters);
  printf("mul64: %.3f ms = %.2f ns per (4 x 32x32->64 multiply + 4 xor) per wavefront -> one multiply ~ %.1f plain ops\\n", t1, t1 * 1e6 / per_simd_iters,
         ((t1 / t0) * 16.0 - 4.0) / 4.0);
  printf("trans: %.3f ms = %.2f ns per (4 transcendentals + ~10 plain) per wavefront -> one transcendental ~ %.1f plain ops\\n", t2, t2 * 1e6 / per_simd_iters,
         ((t2 / t0) * 16.0 - 10.0) / 4.0);
  return 0;
}
