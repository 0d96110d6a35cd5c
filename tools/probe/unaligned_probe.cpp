// Do 16-byte buffer / global accesses work at addresses that are only 4-byte aligned on gfx950?  (The maps of a grid that is not a
// multiple of 4 cells wide -- the reference's default 493 x 493 -- start every row at such an address.)
// hipcc --offload-arch=gfx950 -O3 unaligned_probe.cpp -o unaligned_probe && ./unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(float* buf, float* out, int n, int shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, n * 4, 0x00020000);
  const int off = (i * 4 + shift) * 4;   // byte offset: 16 B per lane, shifted by `shift` floats
  u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  float4 w = *reinterpret_cast<const float4*>(buf + i * 4 + shift);
  out[i * 8 + 0] = __uint_as_float(v.x); out[i * 8 + 1] = __uint_as_float(v.y); out[i * 8 + 2] = __uint_as_float(v.z); out[i * 8 + 3] = __uint_as_float(v.w);
  out[i * 8 + 4] = w.x; out[i * 8 + 5] = w.y; out[i * 8 + 6] = w.z; out[i * 8 + 7] = w.w;
  // unaligned 16-byte store back, shifted by one more float, and an unaligned 2-byte load
  u4 s; s.x = __float_as_uint(w.x + 1000.f); s.y = __float_as_uint(w.y + 1000.f); s.z = __float_as_uint(w.z + 1000.f); s.w = __float_as_uint(w.w + 1000.f);
  __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(out + 8 * 1024), 0, n * 4, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(s, r2, off, 0, 0);
}
__global__ void k16(const unsigned char* b, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, n, 0x00020000);
  out[i] = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, i, 0, 0);   // byte offset i: odd addresses too
}
int main() {
  const int lanes = 256, n = lanes * 4 + 16;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, (8 * 1024 + n) * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  int bad = 0;
  for (int shift = 0; shift < 4; ++shift) {
    hipMemset(o, 0, (8 * 1024 + n) * 4);
    k<<<lanes / 64, 64>>>(d, o, n, shift);
    std::vector<float> r(8 * 1024 + n);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < lanes; ++i)
      for (int j = 0; j < 4; ++j) {
        const float want = (float)(i * 4 + shift + j);
        if (r[i * 8 + j] != want || r[i * 8 + 4 + j] != want) { if (bad < 5) printf("shift %d lane %d j %d: buffer %g global %g want %g\n", shift, i, j, r[i * 8 + j], r[i * 8 + 4 + j], want); ++bad; }
        if (r[8 * 1024 + i * 4 + shift + j] != want + 1000.f) { if (bad < 5) printf("store shift %d lane %d j %d: %g want %g\n", shift, i, j, r[8 * 1024 + i * 4 + shift + j], want + 1000.f); ++bad; }
      }
  }
  std::vector<unsigned char> hb(300);
  for (int i = 0; i < 300; ++i) hb[i] = (unsigned char)(i * 7 + 3);
  unsigned char* db; unsigned* ob;
  hipMalloc(&db, 300); hipMalloc(&ob, 256 * 4);
  hipMemcpy(db, hb.data(), 300, hipMemcpyHostToDevice);
  k16<<<4, 64>>>(db, ob, 300);
  std::vector<unsigned> rb(256);
  hipMemcpy(rb.data(), ob, 256 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 256; ++i) {
    const unsigned want = hb[i] | (hb[i + 1] << 8);
    if (rb[i] != want) { if (bad < 10) printf("u16 at byte %d: %u want %u\n", i, rb[i], want); ++bad; }
  }
  printf("unaligned 16-byte buffer/global loads, stores and 2-byte loads: %s (%d mismatches)\n", bad ? "BROKEN" : "OK", bad);
  return bad != 0;
}
