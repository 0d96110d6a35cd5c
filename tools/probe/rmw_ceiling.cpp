// What does a BARE read-modify-write reach on the map kernels' access shapes?  (round 6)  K3 and the tile fusion sit at 0.55-0.59 of
// the 8 TB/s peak (0.68-0.72 of the streaming-copy rate) at every grid size, and at config 4's shape the fusion loses nothing without
// its reward arithmetic, 6 % without its whole op chain, 4 % without its code loads -- so what bounds them is the cells' own traffic.
// This probe times kernels that do NOTHING but load 16 bytes per lane, add one and store them back, over
//   dense     the whole buffer (every line read once and written once: the streaming-copy mix, in place)
//   rows<W>   per 1 KiB-row map (256 x 256 float32) or 2 KiB-row map (512 x 512) a footprint of ROWS rows x W lane-loads whose rows
//             start at a pseudo-random 16-byte phase of a 128-byte line, lane-loads dealt out in row-major runs like K3's -- i.e. K3's
//             and the fusion's map traffic with no truth / code / Philox / op chain / reward at all
// and prints GB/s of REQUESTED bytes (read + written) and of the 128-byte lines those requests touch.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/rmw_ceiling.cpp -o tools/probe/rmw_ceiling
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_dense(float4* __restrict__ buf, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) { float4 v = buf[i]; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; buf[i] = v; }
}

// grid = (parts, maps): a workgroup of 128 threads takes a run of 128 * 2 consecutive lane-loads of the footprint's row-major order
template <int G, int ROWS, int W>
__global__ __launch_bounds__(128) void k_rows(char* __restrict__ buf) {
  const int m = blockIdx.y;
  char* map = buf + (size_t)m * G * G * 4;
  const unsigned h = (unsigned)m * 2654435761u;
  const int x0 = (h >> 8) % (G - ROWS);
  const int col0 = ((h >> 20) % ((G * 4 - W * 16) / 16)) * 16;     // any 16-byte phase
  const int base = blockIdx.x * 256;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int t = base + q * 128 + threadIdx.x;
    if (t < ROWS * W) {
      const int row = t / W, g = t - row * W;
      float4* p = reinterpret_cast<float4*>(map + (size_t)(x0 + row) * (G * 4) + col0 + g * 16);
      float4 v = *p; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; *p = v;
    }
  }
}

// The same footprints as k_rows, but every row segment ROUNDED OUTWARDS to whole 128-byte lines: the lanes of the partial lines at a row's
// ends also load and store the cells next to the footprint (unchanged) -- more bytes requested, the same lines fetched, and every line written
// whole instead of in part.  WL = lane-loads per row after rounding at most (W + 14 for a 16-byte phase: up to 7 groups at either end).
template <int G, int ROWS, int W>
__global__ __launch_bounds__(128) void k_rows_rounded(char* __restrict__ buf) {
  constexpr int WL = ((W * 16 + 127 + 112) / 128) * 8;
  const int m = blockIdx.y;
  char* map = buf + (size_t)m * G * G * 4;
  const unsigned h = (unsigned)m * 2654435761u;
  const int x0 = (h >> 8) % (G - ROWS);
  const int col0 = ((h >> 20) % ((G * 4 - W * 16) / 16)) * 16;
  const int c0 = col0 & ~127, c1 = (col0 + W * 16 + 127) & ~127;       // the rounded byte range of a row
  const int wl = (c1 - c0) / 16;
  const int base = blockIdx.x * 256;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int t = base + q * 128 + threadIdx.x;
    if (t < ROWS * wl) {
      const int row = t / wl, g = t - row * wl;
      float4* p = reinterpret_cast<float4*>(map + (size_t)(x0 + row) * (G * 4) + c0 + g * 16);
      float4 v = *p;
      const bool in = c0 + g * 16 >= col0 && c0 + g * 16 < col0 + W * 16;
      if (in) { v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; }
      *p = v;
    }
  }
  (void)WL;
}

template <int G, int ROWS, int W>
static void rows_rounded(char* buf, int maps, const char* name) {
  constexpr int WL = ((W * 16 + 127 + 112) / 128) * 8;
  const dim3 grid((ROWS * WL + 255) / 256, maps);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_rows_rounded<G, ROWS, W>), grid, dim3(128), 0, 0, buf);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  for (int rep = 0; rep < 10; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rows_rounded<G, ROWS, W>), grid, dim3(128), 0, 0, buf);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best; sum += ms;
  }
  const double useful = 2.0 * maps * ROWS * W * 16;
  printf("{\"kernel\": \"%s\", \"maps\": %d, \"useful_MB\": %.1f, \"avg_us\": %.1f, \"min_us\": %.1f, \"useful_GBps\": %.0f}\n", name, maps, useful / 1e6,
         sum / 10 * 1e3, best * 1e3, useful / (sum / 10 * 1e-3) / 1e9);
}

template <int G, int ROWS, int W>
static void rows(char* buf, int maps, const char* name) {
  const dim3 grid((ROWS * W + 255) / 256, maps);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_rows<G, ROWS, W>), grid, dim3(128), 0, 0, buf);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  for (int rep = 0; rep < 10; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_rows<G, ROWS, W>), grid, dim3(128), 0, 0, buf);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best; sum += ms;
  }
  // lines touched: host count over the same pseudo-random phases
  double lines = 0;
  for (int m = 0; m < maps; ++m) {
    const unsigned h = (unsigned)m * 2654435761u;
    const int col0 = ((h >> 20) % ((G * 4 - W * 16) / 16)) * 16;
    lines += (double)ROWS * ((col0 + W * 16 - 1) / 128 - col0 / 128 + 1);
  }
  const double req = 2.0 * maps * ROWS * W * 16, touched = 2.0 * lines * 128;
  printf("{\"kernel\": \"%s\", \"maps\": %d, \"requested_MB\": %.1f, \"lines_MB\": %.1f, \"avg_us\": %.1f, \"min_us\": %.1f, \"requested_GBps\": %.0f, \"lines_GBps\": %.0f}\n",
         name, maps, req / 1e6, touched / 1e6, sum / 10 * 1e3, best * 1e3, req / (sum / 10 * 1e-3) / 1e9, touched / (sum / 10 * 1e-3) / 1e9);
}

// The same bare read-modify-write over RUNS of contiguous bytes: per 256 KiB map NRUNS runs of RUN bytes, PITCH bytes apart, the first one
// at a pseudo-random multiple of ALIGN bytes.  Rows of a row-major map are runs of 368 B at a pitch of 1 KiB; a map stored as 128-byte tiles of
// 8 rows x 4 cells (tile rows of 64 tiles = 8 KiB) gives a 90 x 90 footprint 12 runs of 23 tiles = 2944 B; as tiles of 4 rows x 8 cells (tile
// rows of 32 tiles = 4 KiB) 23 runs of 12 tiles = 1536 B -- what a blocked layout of the maps could reach at best.
template <int RUN, int PITCH, int NRUNS, int ALIGN>
__global__ __launch_bounds__(128) void k_runs(char* __restrict__ buf) {
  constexpr int W = RUN / 16;
  const int m = blockIdx.y;
  char* map = buf + (size_t)m * 262144;
  const unsigned h = (unsigned)m * 2654435761u;
  const int span = 262144 - (NRUNS - 1) * PITCH - RUN;
  const int start = (int)((h >> 8) % (unsigned)(span / ALIGN)) * ALIGN;
  const int base = blockIdx.x * 256;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int t = base + q * 128 + threadIdx.x;
    if (t < NRUNS * W) {
      const int r = t / W, g = t - r * W;
      float4* p = reinterpret_cast<float4*>(map + start + (size_t)r * PITCH + g * 16);
      float4 v = *p; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; *p = v;
    }
  }
}

template <int RUN, int PITCH, int NRUNS, int ALIGN>
static void runs(char* buf, int maps, const char* name) {
  constexpr int W = RUN / 16;
  const dim3 grid((NRUNS * W + 255) / 256, maps);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_runs<RUN, PITCH, NRUNS, ALIGN>), grid, dim3(128), 0, 0, buf);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  for (int rep = 0; rep < 10; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((k_runs<RUN, PITCH, NRUNS, ALIGN>), grid, dim3(128), 0, 0, buf);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best; sum += ms;
  }
  const double req = 2.0 * maps * NRUNS * RUN;
  printf("{\"kernel\": \"%s\", \"maps\": %d, \"requested_MB\": %.1f, \"avg_us\": %.1f, \"min_us\": %.1f, \"requested_GBps\": %.0f}\n", name, maps, req / 1e6,
         sum / 10 * 1e3, best * 1e3, req / (sum / 10 * 1e-3) / 1e9);
}

int main() {
  const size_t bytes = (size_t)6 << 30;      // 6 GiB: 24576 maps of 256^2 or 6144 of 512^2
  char* buf = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMemset(buf, 0, bytes));
  CHECK(hipDeviceSynchronize());
  {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const size_t n16 = bytes / 16 / 4;       // 1.5 GiB per launch
    float sum = 0.f, best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(k_dense, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<float4*>(buf) + (size_t)(rep % 4) * n16, n16);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      if (rep) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("{\"kernel\": \"dense read-modify-write in place\", \"requested_MB\": %.1f, \"avg_us\": %.1f, \"min_us\": %.1f, \"requested_GBps\": %.0f}\n",
           2.0 * n16 * 16 / 1e6, sum / 5 * 1e3, best * 1e3, 2.0 * n16 * 16 / (sum / 5 * 1e-3) / 1e9);
  }
  // config 2's shapes: 256^2 maps; footprints 30 / 60 / 90 cells a side (8 / 16 / 23 lane-loads per row), as many maps as K3 touches
  // per launch (4096) and as the fusion does (~3 x that)
  rows<256, 90, 23>(buf, 4096, "rows 256^2, 90 x 23 groups (15 m footprint), 4096 maps");
  rows<256, 90, 23>(buf, 12288, "rows 256^2, 90 x 23 groups, 12288 maps");
  rows<256, 60, 16>(buf, 12288, "rows 256^2, 60 x 16 groups (10 m), 12288 maps");
  rows<256, 30, 8>(buf, 24576, "rows 256^2, 30 x 8 groups (5 m), 24576 maps");
  // config 4's: 512^2 maps, footprints 180 cells a side (46 lane-loads per row)
  rows<512, 180, 46>(buf, 6144, "rows 512^2, 180 x 46 groups (15 m), 6144 maps");
  rows<512, 120, 31>(buf, 6144, "rows 512^2, 120 x 31 groups (10 m), 6144 maps");
  // the same footprints with every row segment rounded outwards to whole lines (compare avg_us with the rows above: same useful bytes)
  rows_rounded<256, 90, 23>(buf, 4096, "rows ROUNDED to whole lines 256^2, 90 x 23 groups, 4096 maps");
  rows_rounded<256, 90, 23>(buf, 12288, "rows ROUNDED to whole lines 256^2, 90 x 23 groups, 12288 maps");
  rows_rounded<256, 60, 16>(buf, 12288, "rows ROUNDED to whole lines 256^2, 60 x 16 groups, 12288 maps");
  rows_rounded<256, 30, 8>(buf, 24576, "rows ROUNDED to whole lines 256^2, 30 x 8 groups, 24576 maps");
  rows_rounded<512, 180, 46>(buf, 6144, "rows ROUNDED to whole lines 512^2, 180 x 46 groups, 6144 maps");
  // what a blocked layout of the 256^2 maps could reach: the 90 x 90 footprint as runs of whole 128-byte tiles
  runs<368, 1024, 90, 16>(buf, 12288, "runs: 90 x 368 B at 1 KiB pitch, 16-byte phase (= rows 256^2 above), 12288 maps");
  runs<384, 1024, 90, 128>(buf, 12288, "runs: 90 x 384 B at 1 KiB pitch, line-aligned (rows forced onto line boundaries), 12288 maps");
  runs<1536, 4096, 23, 128>(buf, 12288, "runs: 23 x 1536 B at 4 KiB pitch (tiles of 4 rows x 8 cells), 12288 maps");
  runs<2944, 8192, 12, 128>(buf, 12288, "runs: 12 x 2944 B at 8 KiB pitch (tiles of 8 rows x 4 cells), 12288 maps");
  runs<11776, 32768, 3, 128>(buf, 12288, "runs: 3 x 11776 B at 32 KiB pitch (tiles of 32 rows x 1 cell-group... i.e. column-blocked), 12288 maps");
  runs<35328, 65536, 1, 128>(buf, 12288, "runs: 1 x 35328 B (a footprint stored contiguously), 12288 maps");
  // (round 6, second session) the same comparison at K3's own launch size (4096 maps) and for the smaller footprints: rows against 4 x 8-cell tiles
  runs<368, 1024, 90, 16>(buf, 4096, "runs: 90 x 368 B at 1 KiB pitch, 16-byte phase, 4096 maps");
  runs<1536, 4096, 23, 128>(buf, 4096, "runs: 23 x 1536 B at 4 KiB pitch (4 x 8 tiles, 15 m), 4096 maps");
  runs<1664, 4096, 24, 128>(buf, 4096, "runs: 24 x 1664 B at 4 KiB pitch (4 x 8 tiles, 15 m, worst-case phase: 24 x 13 tiles), 4096 maps");
  runs<240, 1024, 60, 16>(buf, 12288, "runs: 60 x 240 B at 1 KiB pitch, 16-byte phase (10 m rows), 12288 maps");
  runs<1152, 4096, 16, 128>(buf, 12288, "runs: 16 x 1152 B at 4 KiB pitch (4 x 8 tiles, 10 m: 16 x 9 tiles), 12288 maps");
  runs<240, 1024, 60, 16>(buf, 4096, "runs: 60 x 240 B at 1 KiB pitch, 16-byte phase (10 m rows), 4096 maps");
  runs<1152, 4096, 16, 128>(buf, 4096, "runs: 16 x 1152 B at 4 KiB pitch (4 x 8 tiles, 10 m), 4096 maps");
  runs<128, 1024, 30, 16>(buf, 24576, "runs: 30 x 128 B at 1 KiB pitch, 16-byte phase (5 m rows), 24576 maps");
  runs<640, 4096, 9, 128>(buf, 24576, "runs: 9 x 640 B at 4 KiB pitch (4 x 8 tiles, 5 m: 9 x 5 tiles), 24576 maps");
  CHECK(hipFree(buf));
  return 0;
}
