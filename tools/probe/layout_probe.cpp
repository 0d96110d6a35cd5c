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

__global__ void __launch_bounds__(256) rowmajor(float* maps, const int* rect, int split, const uint8_t* plane, uint8_t* outp, int mode) {
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
    uint32_t t = 0;
    const size_t bo = (size_t)(m >> 2) * G * G + (size_t)(xl + row) * G + y0 + gl * 4;
    if (mode) t = *reinterpret_cast<const uint32_t*>(plane + bo);
    v.x += 1.f + (t & 1); v.y += 1.f; v.z += 1.f; v.w += 1.f;
    *p = v;
    if (mode) *reinterpret_cast<uint32_t*>(outp + (size_t)m * 96 * 96 + (size_t)row * 96 + gl * 4) = t ^ 1;
  }
}

__global__ void __launch_bounds__(256) tiled(float* maps, const int* rect, int split, const uint8_t* plane, uint8_t* outp, int mode) {
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
      uint32_t t = 0;
      const int x = (pr0 + pr) * 4 + r4, y = pc * 8 + half * 4;
      if (mode == 1) {  // row-major byte planes, tiled lanes
        t = *reinterpret_cast<const uint32_t*>(plane + (size_t)(m >> 2) * G * G + (size_t)x * G + y);
      } else if (mode == 2) {  // tiled byte planes
        t = *reinterpret_cast<const uint32_t*>(plane + (size_t)(m >> 2) * G * G + ((size_t)(pr0 + pr) * (G / 8) + pc) * 32 + r4 * 8 + half * 4);
      }
      v.x += 1.f + (t & 1); v.y += 1.f; v.z += 1.f; v.w += 1.f;
      *p = v;
      if (mode == 1) *reinterpret_cast<uint32_t*>(outp + (size_t)m * 104 * 104 + (size_t)(x - (pr0 * 4)) * 104 + (y - pc0 * 8)) = t ^ 1;
      else if (mode == 2) *reinterpret_cast<uint32_t*>(outp + (size_t)m * 104 * 104 + ((size_t)pr * 13 + (pc - pc0)) * 32 + r4 * 8 + half * 4) = t ^ 1;
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
  uint8_t *plane, *outp;
  CK(hipMalloc(&plane, (size_t)(M / 4) * G * G)); CK(hipMemset(plane, 1, (size_t)(M / 4) * G * G));
  CK(hipMalloc(&outp, (size_t)M * 104 * 104));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int split : {2}) {
    for (int which = 0; which < 5; ++which) {
      // 0: rowmajor maps only; 1: rowmajor maps+planes; 2: tiled maps only; 3: tiled maps + row-major planes; 4: tiled maps + tiled planes
      auto launch = [&]() {
        if (which == 0) rowmajor<<<M * split, 256>>>(d, dr, split, plane, outp, 0);
        else if (which == 1) rowmajor<<<M * split, 256>>>(d, dr, split, plane, outp, 1);
        else tiled<<<M * split, 256>>>(d, dr, split, plane, outp, which - 2);
      };
      for (int rep = 0; rep < 3; ++rep) launch();
      CK(hipEventRecord(a));
      for (int rep = 0; rep < 10; ++rep) launch();
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const char* names[5] = {"rowmajor maps", "rowmajor maps+planes", "tiled maps", "tiled maps + rowmajor planes", "tiled maps + tiled planes"};
      printf("%-30s split=%d: %.1f us/launch (%.1f M cells)\n", names[which], split, ms * 100, cells / 1e6);
    }
  }
  return 0;
}
