// Probe 3: is the footprint read-modify-write bound per VISIT (latency / request issue) or per BYTE?
// One kernel updates the same footprint rectangle in ONE map, or in TWO different maps (local + global, as a fused
// sense + global-fusion kernel would).  Cold data (map sets cycled past the 256 MB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int G = 256;

template <int NMAPS>
__global__ void __launch_bounds__(256) kA(float* maps, float* maps2, const int* rect, int split) {
  const int m = blockIdx.x / split, part = blockIdx.x % split;
  const int yu = rect[m * 4], yd = rect[m * 4 + 1], xl = rect[m * 4 + 2], xr = rect[m * 4 + 3];
  const int y0 = yu & ~3, groups = (yd - y0 + 3) / 4, h = xr - xl, w = yd - yu;
  int shift = groups <= 1 ? 0 : 32 - __clz(groups - 1); if (shift > 6) shift = 6;
  const int lpr = 1 << shift, rpw = 64 >> shift;
  const int per = (h + split - 1) / split, r0 = part * per, r1 = min(h, r0 + per);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane >> shift, gl = lane & (lpr - 1);
  float* map = maps + (size_t)m * G * G;
  float* map2 = maps2 + (size_t)m * G * G;
  for (int gi = gl; gi < groups; gi += lpr) {
    const int y = y0 + gi * 4;
    for (int row = r0 + wv * rpw + sub; row < r1; row += 4 * rpw) {
      const size_t o = (size_t)(xl + row) * G + y;
      float4 v = *reinterpret_cast<float4*>(map + o), v2;
      if (NMAPS == 2) v2 = *reinterpret_cast<float4*>(map2 + o);
      float* f = &v.x; float* f2 = &v2.x;
      for (int q = 0; q < 4; ++q) { const bool in = (unsigned)(y + q - yu) < (unsigned)w; f[q] = in ? f[q] + 0.5f : f[q]; if (NMAPS == 2) f2[q] = in ? f2[q] - 0.25f : f2[q]; }
      *reinterpret_cast<float4*>(map + o) = v;
      if (NMAPS == 2) *reinterpret_cast<float4*>(map2 + o) = v2;
    }
  }
}

int main() {
  const int M = 4096, SETS = 4;
  float *d, *d2; int* dr;
  CK(hipMalloc(&d, (size_t)SETS * M * G * G * 4)); CK(hipMemset(d, 0, (size_t)SETS * M * G * G * 4));
  CK(hipMalloc(&d2, (size_t)SETS * M * G * G * 4)); CK(hipMemset(d2, 0, (size_t)SETS * M * G * G * 4));
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
  for (int split : {1, 2}) for (int which = 0; which < 3; ++which) {
    int it = 0;
    auto launch = [&]() {
      const int set = (it++) % SETS;
      float* dm = d + (size_t)set * M * G * G; float* dm2 = d2 + (size_t)set * M * G * G;
      if (which == 0) kA<1><<<M * split, 256>>>(dm, dm2, dr, split);
      else if (which == 1) kA<2><<<M * split, 256>>>(dm, dm2, dr, split);
      else { kA<1><<<M * split, 256>>>(dm, dm2, dr, split); kA<1><<<M * split, 256>>>(dm2, dm, dr, split); }
    };
    for (int rep = 0; rep < 3; ++rep) launch();
    CK(hipEventRecord(a));
    for (int rep = 0; rep < 12; ++rep) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const char* nm[3] = {"one map", "two maps, one kernel", "two maps, two kernels"};
    printf("%-22s split=%d: %.1f us (%.1f M cells per map)\n", nm[which], split, ms * 1000 / 12, cells / 1e6);
  }
  return 0;
}
