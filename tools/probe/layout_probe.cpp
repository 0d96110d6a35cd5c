// Probe: achievable HBM rate of a read-modify-write over random 90x90-cell float tiles (one per 256x256 map, 4096 maps)
// in (a) row-major maps with float4 per lane, 2 rows x 32 lanes per wavefront (what K3 does), and
// (b) a patch-tiled layout (128-byte line = 4 rows x 8 cols), 8 consecutive patches (1 KiB) per wavefront instruction.
// hipcc -O3 --offload-arch=gfx950 layout_probe.cpp -o layout_probe && ./layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int G = 256;

__global__ void __launch_bounds__(256) rowmajor(float* maps, const int* rect, int split) {
  const int m = blockIdx.x / split, part = blockIdx.x % split;
  const int yu = rect[m * 4], yd = rect[m * 4 + 1], xl = rect[m * 4 + 2], xr = rect[m * 4 + 3];
  const int y0 = yu & ~3, groups = (yd - y0 + 3) / 4;
  const int h = xr - xl, rpw_rows = (h + split - 1) / split, r0 = part * rpw_rows, r1 = min(h, r0 + rpw_rows);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane >> 5, gl = lane & 31;
  float* map = maps + (size_t)m * G * G;
  if (gl >= groups) return;
  for (int row = r0 + wv * 2 + sub; row < r1; row += 8) {
    float4* p = reinterpret_cast<float4*>(map + (size_t)(xl + row) * G + y0 + gl * 4);
    float4 v = *p;
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    *p = v;
  }
}

__global__ void __launch_bounds__(256) tiled(float* maps, const int* rect, int split) {
  const int m = blockIdx.x / split, part = blockIdx.x % split;
  const int yu = rect[m * 4], yd = rect[m * 4 + 1], xl = rect[m * 4 + 2], xr = rect[m * 4 + 3];
  const int pc0 = yu >> 3, pc1 = (yd + 7) >> 3, pr0 = xl >> 2, pr1 = (xr + 3) >> 2;
  const int prs = pr1 - pr0, per = (prs + split - 1) / split, a0 = part * per, a1 = min(prs, a0 + per);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pc_l = lane >> 3, r4 = (lane >> 1) & 3, half = lane & 1;
  float* map = maps + (size_t)m * G * G;
  for (int pr = a0 + wv; pr < a1; pr += 4)
    for (int pc = pc0 + pc_l; pc < pc1; pc += 8) {
      float4* p = reinterpret_cast<float4*>(map + ((size_t)(pr0 + pr) * (G / 8) + pc) * 32 + r4 * 8 + half * 4);
      float4 v = *p;
      v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
      *p = v;
    }
}

int main() {
  const int M = 4096;
  float* d; int* dr;
  CK(hipMalloc(&d, (size_t)M * G * G * 4)); CK(hipMemset(d, 0, (size_t)M * G * G * 4));
  std::vector<int> r(M * 4);
  srand(1);
  for (int m = 0; m < M; ++m) {
    int cx = (rand() % 11) * 25, cy = (rand() % 11) * 25, rad = 45;
    r[m * 4] = std::max(cy - rad, 0); r[m * 4 + 1] = std::min(cy + rad, G - 1); r[m * 4 + 2] = std::max(cx - rad, 0); r[m * 4 + 3] = std::min(cx + rad, G - 1);
  }
  CK(hipMalloc(&dr, M * 16)); CK(hipMemcpy(dr, r.data(), M * 16, hipMemcpyHostToDevice));
  double cells = 0; for (int m = 0; m < M; ++m) cells += (double)(r[m*4+1]-r[m*4]) * (r[m*4+3]-r[m*4+2]);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int split : {1, 2, 4}) {
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 3; ++rep) { if (which) tiled<<<M * split, 256>>>(d, dr, split); else rowmajor<<<M * split, 256>>>(d, dr, split); }
      CK(hipEventRecord(a));
      for (int rep = 0; rep < 10; ++rep) { if (which) tiled<<<M * split, 256>>>(d, dr, split); else rowmajor<<<M * split, 256>>>(d, dr, split); }
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("%s split=%d: %.1f us/launch, %.0f GB/s useful (8 B/cell, %.1f M cells)\n", which ? "tiled   " : "rowmajor", split, ms * 100, cells * 8 / (ms * 1e-4) / 1e9, cells / 1e6);
    }
  }
  return 0;
}
