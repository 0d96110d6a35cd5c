// Host side of libippmarl.so under AddressSanitizer (SURVEY section 5's suggestion), without Python in the process (an instrumented
// runtime preloaded under the interpreter + torch + the HIP runtime does not get past import).  Reads an ippm_config image written by
//   python tools/write_config_image.py small cfg.bin
// and drives the batched step through the C-ABI: context, reset (episode scalars, maps), 2 episodes of plan -> fuse -> sense with the
// dispatch-bound kernel timing on for part of them, counters, team sizes, the error paths of the entry points (null arguments, bad
// flags, out-of-range selections).  Device code is not instrumented (gfx950 without xnack); what is: argument checks, launch set-up,
// the event pools, counters, error strings.
//   hipcc -O1 -g -fsanitize=address -shared-libasan --offload-arch=gfx950 -Iinclude tools/probe/abi_asan_smoke.cpp \
//         -Lipp-marl_amd/lib -lippmarl_asan -Wl,-rpath,$PWD/ipp-marl_amd/lib -o tools/probe/abi_asan_smoke
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ippmarl.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define OK(x) do { int rc_ = (x); if (rc_ != 0) { std::fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, ippm_last_error()); return 3; } } while (0)
#define FAILS(x) do { int rc_ = (x); if (rc_ == 0) { std::fprintf(stderr, "%s was expected to fail\n", #x); return 4; } ++refused; } while (0)

template <typename T>
static T* dalloc(size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
  hipMemset(p, 0, n * sizeof(T));
  return static_cast<T*>(p);
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: abi_asan_smoke cfg.bin [envs]\n"); return 1; }
  ippm_config cfg;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(&cfg, 1, sizeof(cfg), f) != sizeof(cfg)) { std::fprintf(stderr, "cannot read %zu bytes of ippm_config from %s\n", sizeof(cfg), argv[1]); return 1; }
  std::fclose(f);
  if ((int)sizeof(cfg) != ippm_config_size()) { std::fprintf(stderr, "config layout mismatch\n"); return 1; }
  const int E = argc > 2 ? std::atoi(argv[2]) : 6, N = cfg.n_agents, A = cfg.n_actions, gx = cfg.grid_x, gy = cfg.grid_y;
  int refused = 0;
  ippm_ctx* ctx = nullptr;
  OK(ippm_ctx_create(&cfg, &ctx));
  int64_t words = 0;
  OK(ippm_work_words(ctx, E, &words));
  const size_t tile = (size_t)cfg.tile_stride * (cfg.tile_stride / 4), truth_b = (((size_t)gx * gy + 31) / 32) * 4;
  int64_t* episode = dalloc<int64_t>(E);
  int32_t *pos = dalloc<int32_t>(E * N * 3), *rect = dalloc<int32_t>(E * N * 4), *rect_next = dalloc<int32_t>((size_t)E * N * IPPM_SENSE_REC_WORDS);
  int32_t *ws = dalloc<int32_t>((size_t)E * (N + 1) * IPPM_WS_WORDS), *work = dalloc<int32_t>((size_t)words), *action = dalloc<int32_t>(E * N), *fault = dalloc<int32_t>(E);
  int32_t *split_pct = dalloc<int32_t>(E * 2), *team = dalloc<int32_t>(E);
  float *local = dalloc<float>((size_t)E * N * gx * gy), *global = dalloc<float>((size_t)E * gx * gy), *comm_range = dalloc<float>(E), *reward = dalloc<float>(E * 2);
  uint8_t *code = dalloc<uint8_t>((size_t)E * N * tile), *truth = dalloc<uint8_t>((size_t)E * truth_b), *comm = dalloc<uint8_t>(E * N * N), *mask = dalloc<uint8_t>(E * N * A);
  double* sums = dalloc<double>(E * 8);
  if (!episode || !local || !work || !sums) { std::fprintf(stderr, "hipMalloc failed\n"); return 2; }
  std::vector<int64_t> eps(E);
  std::vector<int32_t> teams(E);
  hipStream_t st;
  HIPCK(hipStreamCreate(&st));
  for (int wave = 0; wave < 2; ++wave) {
    for (int e = 0; e < E; ++e) { eps[e] = 1 + wave * E + e; teams[e] = 1 + (e + wave) % N; }
    HIPCK(hipMemcpy(episode, eps.data(), sizeof(int64_t) * E, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(team, teams.data(), sizeof(int32_t) * E, hipMemcpyHostToDevice));
    OK(ippm_set_team_sizes(ctx, wave ? team : nullptr));          // second episode: mixed team sizes
    OK(ippm_reset_episode(ctx, episode, pos, truth, nullptr, nullptr, split_pct, comm_range, ws, sums, nullptr, E, st));
    OK(ippm_reset_maps(ctx, episode, pos, truth, local, global, nullptr, code, rect, ws, wave == 0, E, st));
    OK(ippm_kernel_timing(ctx, wave));
    for (int t = 0; t <= cfg.budget; ++t) {
      OK(ippm_plan_step(ctx, episode, pos, comm_range, nullptr, comm, rect, ws, t, IPPM_STEP_COMM | IPPM_STEP_GLOBAL | IPPM_STEP_MOVE, nullptr, nullptr, 1,
                        mask, action, fault, rect_next, work, E, st));
      OK(ippm_fuse_step(ctx, local, global, code, ws, sums, nullptr, work, E, st));
      OK(ippm_sense_step(ctx, episode, pos, truth, local, nullptr, code, rect_next, rect, ws, nullptr, sums, reward, t + 1, -1, E, st));
    }
    OK(ippm_sync(ctx, st));
  }
  for (int cls = 0; cls < IPPM_TIMED_CLASSES; ++cls) {
    int64_t n = 0;
    double tot = 0, mn = 0;
    char name[128];
    OK(ippm_read_kernel_times(ctx, cls, 1, &n, &tot, &mn, name, (int32_t)sizeof(name), st));
    if (n) std::printf("class %d: %lld launches of %s, %.1f us on average\n", cls, (long long)n, name, tot / n);
  }
  ippm_counters cnt;
  OK(ippm_read_counters(ctx, &cnt, 1, st));
  std::vector<float> r(E * 2);
  HIPCK(hipMemcpy(r.data(), reward, sizeof(float) * E * 2, hipMemcpyDeviceToHost));
  std::printf("sensed cells %llu, work-list rejects %llu, last relative reward of env 0: %f\n", (unsigned long long)cnt.sense_cells,
              (unsigned long long)cnt.reserved[0], r[0]);
  // ---- error paths: every one must refuse and leave a message
  FAILS(ippm_ctx_create(nullptr, &ctx));
  FAILS(ippm_plan_step(ctx, episode, nullptr, comm_range, nullptr, comm, rect, ws, 0, IPPM_STEP_COMM, nullptr, nullptr, 1, mask, action, fault, rect_next, work, E, st));
  FAILS(ippm_plan_step(ctx, episode, pos, comm_range, nullptr, comm, rect, ws, 0, 0, nullptr, nullptr, 1, mask, action, fault, rect_next, work, E, st));
  FAILS(ippm_plan_step(ctx, episode, pos, comm_range, nullptr, comm, rect, ws, 0, IPPM_STEP_MOVE, nullptr, nullptr, 7, mask, action, fault, rect_next, work, E, st));
  FAILS(ippm_plan_step(ctx, episode, pos, comm_range, nullptr, comm, rect, ws, 0, IPPM_STEP_MOVE, nullptr, nullptr, 2, mask, action, fault, rect_next, work, E, st));
  FAILS(ippm_fuse_step(ctx, nullptr, global, code, ws, sums, nullptr, work, E, st));
  FAILS(ippm_sense_step(ctx, episode, pos, truth, local, nullptr, code, rect_next, rect, ws, nullptr, sums, nullptr, 1, -1, E, st));
  FAILS(ippm_sense_step(ctx, episode, pos, truth, local, nullptr, code, rect_next, rect, ws, nullptr, nullptr, nullptr, 1, N, E, st));
  FAILS(ippm_work_words(ctx, -1, &words));
  FAILS(ippm_set_team_sizes(nullptr, team));
  if (!ippm_last_error() || !*ippm_last_error()) { std::fprintf(stderr, "no error message\n"); return 4; }
  std::printf("%d bad calls refused; last message: %s\n", refused, ippm_last_error());
  OK(ippm_set_team_sizes(ctx, nullptr));
  OK(ippm_ctx_destroy(ctx));
  std::printf("abi_asan_smoke OK\n");
  return 0;
}
