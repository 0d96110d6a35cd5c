// Probe 4: DRAM-page locality.  Footprint read-modify-write on cold maps, row-major vs tiles whose rows are exactly one
// 128-byte line (32 cells) and whose TR rows are contiguous (TR x 128 B per tile): consecutive footprint rows then fall
// into the same DRAM page.  8 map sets of 1 GiB are cycled so that nothing is found in the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int G = 256;

template <int TR>  // 0: row-major; else tile = TR rows x 32 columns
__device__ __forceinline__ size_t idx(int x, int y) {
  if (TR == 0) return (size_t)x * G + y;
  return ((size_t)(x / TR) * (G / 32) + (y >> 5)) * (TR * 32) + (size_t)(x % TR) * 32 + (y & 31);
}

template <int TR>
__global__ void __launch_bounds__(256) kA(float* maps, const int* rect, int split) {
  const int m = blockIdx.x / split, part = blockIdx.x % split;
  const int yu = rect[m * 4], yd = rect[m * 4 + 1], xl = rect[m * 4 + 2], xr = rect[m * 4 + 3];
  const int y0 = yu & ~3, groups = (yd - y0 + 3) / 4, h = xr - xl, w = yd - yu;
  int shift = groups <= 1 ? 0 : 32 - __clz(groups - 1); if (shift > 6) shift = 6;
  const int lpr = 1 << shift, rpw = 64 >> shift;
  const int per = (h + split - 1) / split, r0 = part * per, r1 = min(h, r0 + per);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane >> shift, gl = lane & (lpr - 1);
  float* map = maps + (size_t)m * G * G;
  for (int gi = gl; gi < groups; gi += lpr) {
    const int y = y0 + gi * 4;
    for (int row = r0 + wv * rpw + sub; row < r1; row += 4 * rpw) {
      float4* p = reinterpret_cast<float4*>(map + idx<TR>(xl + row, y));
      float4 v = *p;
      float* f = &v.x;
      for (int q = 0; q < 4; ++q) { const bool in = (unsigned)(y + q - yu) < (unsigned)w; f[q] = in ? f[q] + 0.5f : f[q]; }
      *p = v;
    }
  }
}

int main() {
  const int M = 4096, SETS = 8;
  float* d; int* dr;
  CK(hipMalloc(&d, (size_t)SETS * M * G * G * 4)); CK(hipMemset(d, 0, (size_t)SETS * M * G * G * 4));
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
  for (int which = 0; which < 5; ++which) {
    int it = 0;
    const int split = 2;
    auto launch = [&]() {
      float* dm = d + (size_t)((it++) % SETS) * M * G * G;
      if (which == 0) kA<0><<<M * split, 256>>>(dm, dr, split);
      else if (which == 1) kA<8><<<M * split, 256>>>(dm, dr, split);
      else if (which == 2) kA<32><<<M * split, 256>>>(dm, dr, split);
      else if (which == 3) kA<64><<<M * split, 256>>>(dm, dr, split);
      else kA<256><<<M * split, 256>>>(dm, dr, split);
    };
    for (int rep = 0; rep < 8; ++rep) launch();
    CK(hipEventRecord(a));
    for (int rep = 0; rep < 16; ++rep) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const char* nm[5] = {"row-major", "tiles 8 x 32", "tiles 32 x 32", "tiles 64 x 32", "column strips 256 x 32"};
    printf("%-24s %.1f us (%.1f M cells, %.2f us/Mcell)\n", nm[which], ms * 1000 / 16, cells / 1e6, ms * 1000 / 16 / (cells / 1e6));
  }
  return 0;
}
