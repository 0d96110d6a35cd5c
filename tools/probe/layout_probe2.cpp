// Probe 2: K3-like read-modify-write (float4 map + u32 truth read + u32 code write) over a realistic MIX of footprints
// (30/60/90 cells at random lattice positions), three lane geometries:
//  A row-major planes, pow2 lanes-per-row (the row-major K3)        B patch-tiled planes, pow2 patch columns per chunk
//  C patch-tiled planes, dense patch slots (a wavefront always holds 8 live patches)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int G = 256, NPC = G / 8, TPC = 14, TPR = 25;

template <int MODE>  // 0: maps only, 1: maps + byte planes (u32 per lane), 2: maps + bit-packed truth + 1 code byte per lane
__global__ void __launch_bounds__(256) kA(float* maps, const uint8_t* truth, uint8_t* code, const int* rect, int split) {
  const int m = blockIdx.x / split, part = blockIdx.x % split;
  const int yu = rect[m * 4], yd = rect[m * 4 + 1], xl = rect[m * 4 + 2], xr = rect[m * 4 + 3];
  const int y0 = yu & ~3, groups = (yd - y0 + 3) / 4, h = xr - xl, w = yd - yu;
  int shift = groups <= 1 ? 0 : 32 - __clz(groups - 1); if (shift > 6) shift = 6;
  const int lpr = 1 << shift, rpw = 64 >> shift;
  const int per = (h + split - 1) / split, r0 = part * per, r1 = min(h, r0 + per);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane >> shift, gl = lane & (lpr - 1);
  float* map = maps + (size_t)m * G * G;
  const uint8_t* tr = truth + (size_t)(m >> 2) * G * G;
  uint8_t* cd = code + (size_t)m * 96 * 96;
  for (int gi = gl; gi < groups; gi += lpr) {
    const int y = y0 + gi * 4;
    for (int row = r0 + wv * rpw + sub; row < r1; row += 4 * rpw) {
      float4* p = reinterpret_cast<float4*>(map + (size_t)(xl + row) * G + y);
      float4 v = *p;
      uint32_t t = 0x01000100u;
      const size_t lin = (size_t)(xl + row) * G + y;
      if (MODE == 1) t = *reinterpret_cast<const uint32_t*>(tr + lin);
      if (MODE == 2) { const uint32_t b = (tr[lin >> 3] >> (lin & 4)) & 0xF; t = (b & 1) | ((b & 2) << 7) | ((b & 4) << 14) | ((b & 8) << 21); }
      float* f = &v.x; uint32_t cw = 0;
      for (int q = 0; q < 4; ++q) { const bool in = (unsigned)(y + q - yu) < (unsigned)w; const uint32_t o = (t >> (8 * q)) & 1; f[q] = in ? f[q] + (o ? 0.5f : -0.5f) : f[q]; cw |= (in ? o : 0) << (8 * q); }
      *p = v;
      if (MODE == 1) *reinterpret_cast<uint32_t*>(cd + (size_t)row * 96 + (y - y0)) = cw;
      if (MODE == 2) cd[(size_t)row * 24 + gi] = (uint8_t)((cw & 1) | ((cw >> 7) & 2) | ((cw >> 14) & 4) | ((cw >> 21) & 8));
    }
  }
}

__device__ __forceinline__ size_t coff(int x, int y) { return ((size_t)(x >> 2) * NPC + (y >> 3)) * 32 + ((x & 3) << 3) + (y & 7); }
__device__ __forceinline__ size_t toff(int x, int y, int xl, int yu) { return ((size_t)((x >> 2) - (xl >> 2)) * TPC + ((y >> 3) - (yu >> 3))) * 32 + ((x & 3) << 3) + (y & 7); }

template <bool DENSE, bool PLANES = true>
__global__ void __launch_bounds__(256) kT(float* maps, const uint8_t* truth, uint8_t* code, const int* rect, int split) {
  const int m = blockIdx.x / split, part = blockIdx.x % split;
  const int yu = rect[m * 4], yd = rect[m * 4 + 1], xl = rect[m * 4 + 2], xr = rect[m * 4 + 3];
  const int pc0 = yu >> 3, npc = ((yd + 7) >> 3) - pc0, pr0 = xl >> 2, npr = ((xr + 3) >> 2) - pr0, h = xr - xl, w = yd - yu;
  const int per = (npr + split - 1) / split, a0 = part * per, a1 = min(npr, a0 + per);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r4 = (lane >> 1) & 3, half = lane & 1, idx8 = lane >> 3;
  float* map = maps + (size_t)m * G * G;
  const uint8_t* tr = truth + (size_t)(m >> 2) * G * G;
  uint8_t* cd = code + (size_t)m * TPR * TPC * 32;
  auto body = [&](int pr, int pc) {
    const int x = (pr0 + pr) * 4 + r4, y = (pc0 + pc) * 8 + half * 4;
    if ((unsigned)(x - xl) >= (unsigned)h) return;
    unsigned inm = 0;
    for (int q = 0; q < 4; ++q) inm |= ((unsigned)(y + q - yu) < (unsigned)w) ? (1u << q) : 0u;
    if (!inm) return;
    const size_t o = coff(x, y);
    float4* p = reinterpret_cast<float4*>(map + o);
    float4 v = *p;
    const uint32_t t = PLANES ? *reinterpret_cast<const uint32_t*>(tr + o) : 0x01000100u;
    float* f = &v.x; uint32_t cw = 0;
    for (int q = 0; q < 4; ++q) { const bool in = (inm >> q) & 1; const uint32_t ob = (t >> (8 * q)) & 1; f[q] = in ? f[q] + (ob ? 0.5f : -0.5f) : f[q]; cw |= (in ? ob : 0) << (8 * q); }
    *p = v;
    if (PLANES) *reinterpret_cast<uint32_t*>(cd + toff(x, y, xl, yu)) = cw;
  };
  if (DENSE) {
    const int slots = (a1 - a0) * npc;
    for (int s = wv * 8 + idx8; s < slots; s += 32) { const int pr = s / npc; body(a0 + pr, s - pr * npc); }
  } else {
    const int mm = min(npc, 8) - 1, shift = mm <= 0 ? 0 : 32 - __clz(mm), ppr = 1 << shift, spw = 8 >> shift;
    const int sub = idx8 >> shift, pcl = idx8 & (ppr - 1);
    for (int pcc = pcl; pcc < npc; pcc += ppr)
      for (int prr = a0 + wv * spw + sub; prr < a1; prr += 4 * spw) body(prr, pcc);
  }
}

int main() {
  const int M = 4096, SETS = 6;  // SETS disjoint map sets cycled so that no launch finds its data in the 256 MB Infinity Cache
  float* d; uint8_t *truth, *code; int* dr;
  CK(hipMalloc(&d, (size_t)SETS * M * G * G * 4)); CK(hipMemset(d, 0, (size_t)SETS * M * G * G * 4));
  CK(hipMalloc(&truth, (size_t)SETS * (M / 4) * G * G)); CK(hipMemset(truth, 1, (size_t)SETS * (M / 4) * G * G));
  CK(hipMalloc(&code, (size_t)SETS * M * TPR * TPC * 32));
  std::vector<int> r(M * 4);
  srand(1);
  const int cen[11] = {0, 25, 51, 76, 102, 128, 153, 179, 204, 230, 256};
  double cells = 0;
  for (int m = 0; m < M; ++m) {
    int cx = cen[rand() % 11], cy = cen[rand() % 11], rad = 15 * (1 + rand() % 3);
    r[m * 4] = std::max(cy - rad, 0); r[m * 4 + 1] = std::min(cy + rad, G - 1); r[m * 4 + 2] = std::max(cx - rad, 0); r[m * 4 + 3] = std::min(cx + rad, G - 1);
    cells += (double)(r[m*4+1]-r[m*4]) * (r[m*4+3]-r[m*4+2]);
  }
  CK(hipMalloc(&dr, M * 16)); CK(hipMemcpy(dr, r.data(), M * 16, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int split : {2}) for (int which = 0; which < 7; ++which) {
    int it = 0;
    auto launch = [&]() {
      const int set = (it++) % SETS;
      float* dm = d + (size_t)set * M * G * G; const uint8_t* tr = truth + (size_t)set * (M / 4) * G * G; uint8_t* cd = code + (size_t)set * M * TPR * TPC * 32;
      if (which == 0) kA<1><<<M * split, 256>>>(dm, tr, cd, dr, split);
      else if (which == 1) kT<false><<<M * split, 256>>>(dm, tr, cd, dr, split);
      else if (which == 2) kT<true><<<M * split, 256>>>(dm, tr, cd, dr, split);
      else if (which == 3) kA<0><<<M * split, 256>>>(dm, tr, cd, dr, split);
      else if (which == 4) kA<2><<<M * split, 256>>>(dm, tr, cd, dr, split);
      else if (which == 5) kT<false, false><<<M * split, 256>>>(dm, tr, cd, dr, split);
      else kT<true, false><<<M * split, 256>>>(dm, tr, cd, dr, split);
    };
    for (int rep = 0; rep < 3; ++rep) launch();
    CK(hipEventRecord(a));
    for (int rep = 0; rep < 12; ++rep) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const char* nm[7] = {"A row-major pow2", "B tiled pow2", "C tiled dense", "A maps only", "A packed planes", "B tiled maps only", "C dense maps only"};
    printf("%-18s split=%d: %.1f us/launch, %.0f GB/s algorithmic (10 B/cell, %.1f M cells)\n", nm[which], split, ms * 1000 / 12, cells * 10 / (ms * 1e-3 / 12) / 1e9, cells / 1e6);
  }
  return 0;
}
