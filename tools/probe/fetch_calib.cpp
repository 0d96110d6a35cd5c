// Calibration of rocprofv3's FETCH_SIZE (and the TCC -> fabric request counters behind it) on FOOTPRINT-SHAPED reads with a known
// line count (round 6; the microarch guide calibrates the counter on full-line streaming reads only and calls other shapes
// uncalibrated).  Read-only kernels over 8192 "maps" of 256 x 256 float32 (1 KiB rows, 2 GB in all: far beyond L2 + Infinity
// Cache), one workgroup per map:
//   k_stream                         every byte of the buffer, 16 B per lane                      (the guide's reference shape)
//   k_rows<START>                    per map a 90-row footprint whose rows are 23 lane-loads of 16 B (368 B) starting START bytes
//                                    into a 128-byte line -- K3's access shape (k_sense_tiles: runs of the row-major group sequence)
// The program prints, per kernel, the bytes requested and the bytes of the distinct 128-B lines / 64-B halves / 32-B sectors
// those requests touch; tools/gpu_fetch_calib.sh runs it under `rocprofv3 --pmc ...` and tools/fetch_calib_summary.py sets the
// counters against them.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/fetch_calib.cpp -o tools/probe/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int MAPS = 8192, G = 256, ROWS = 90, W = 23;
constexpr size_t MAP_BYTES = (size_t)G * G * 4;

__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ src, uint32_t* __restrict__ sink, size_t n16) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// one workgroup per map; lane-load t = (row t / W, group t % W) in row-major order, like K3's runs
template <int START>
__global__ __launch_bounds__(256) void k_rows(const char* __restrict__ src, uint32_t* __restrict__ sink) {
  const char* map = src + (size_t)blockIdx.x * MAP_BYTES;
  const int x0 = 17 + (blockIdx.x * 7) % 120;            // first row of the footprint
  const int col0 = 128 * (1 + blockIdx.x % 3) + START;   // byte offset of the footprint's first group in its row
  uint32_t acc = 0;
  for (int t = threadIdx.x; t < ROWS * W; t += 256) {
    const int row = t / W, g = t - row * W;
    const uint4 v = *reinterpret_cast<const uint4*>(map + (size_t)(x0 + row) * (G * 4) + col0 + g * 16);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int START>
static void rows(const char* buf, uint32_t* sink, const char* name) {
  hipLaunchKernelGGL((k_rows<START>), dim3(MAPS), dim3(256), 0, 0, buf, sink);
  CHECK(hipDeviceSynchronize());
  // what the launch touches, counted on the host (one map is enough: every map has the same shape up to its line phase)
  size_t lines = 0, halves = 0, sectors = 0;
  for (int m = 0; m < 3; ++m) {
    const int col0 = 128 * (1 + m % 3) + START;
    std::set<int> l, h, s;
    for (int g = 0; g < W; ++g)
      for (int b = 0; b < 16; b += 16) { const int a = col0 + g * 16 + b; l.insert(a / 128); h.insert(a / 64); s.insert(a / 32); }
    lines += l.size(); halves += h.size(); sectors += s.size();
  }
  const double per_row = 1.0 / 3.0;
  const double rows_total = (double)MAPS * ROWS;
  printf("{\"kernel\": \"%s\", \"start\": %d, \"requested_bytes\": %.0f, \"line128_bytes\": %.0f, \"half64_bytes\": %.0f, \"sector32_bytes\": %.0f}\n",
         name, START, rows_total * W * 16, rows_total * lines * per_row * 128, rows_total * halves * per_row * 64, rows_total * sectors * per_row * 32);
}

int main() {
  const size_t bytes = (size_t)MAPS * MAP_BYTES;
  char* buf = nullptr;
  uint32_t* sink = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_stream, dim3(MAPS), dim3(256), 0, 0, reinterpret_cast<const uint4*>(buf), sink, bytes / 16);
    CHECK(hipDeviceSynchronize());
  }
  printf("{\"kernel\": \"k_stream\", \"start\": 0, \"requested_bytes\": %.0f, \"line128_bytes\": %.0f, \"half64_bytes\": %.0f, \"sector32_bytes\": %.0f}\n",
         (double)bytes, (double)bytes, (double)bytes, (double)bytes);
  rows<0>(buf, sink, "k_rows<0>");
  rows<16>(buf, sink, "k_rows<16>");
  rows<32>(buf, sink, "k_rows<32>");
  rows<48>(buf, sink, "k_rows<48>");
  rows<64>(buf, sink, "k_rows<64>");
  rows<96>(buf, sink, "k_rows<96>");
  rows<112>(buf, sink, "k_rows<112>");
  CHECK(hipFree(buf));
  CHECK(hipFree(sink));
  return 0;
}
