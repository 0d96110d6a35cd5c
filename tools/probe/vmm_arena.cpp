// Experiment: an arena of device memory assembled from physical chunks of a chosen size, mapped into one virtual range in a
// chosen order (HIP virtual memory management API).  Used by tools/placement_vmm.py to see whether the good / bad kinds of
// allocation (VecEnv.tune_placement) can be produced on purpose.
//   hipcc -O2 -shared -fPIC --offload-arch=gfx950 tools/probe/vmm_arena.cpp -o tools/probe/libvmm_arena.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

struct Arena {
  void* base = nullptr;       // first mapped byte (aligned as asked)
  void* reserved = nullptr;   // the reservation it lies in
  size_t bytes = 0, chunk = 0, reserved_bytes = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "vmm_arena: %s -> %s\n", #x, hipGetErrorString(e_)); return nullptr; } } while (0)

// order: 0 = chunks mapped in the order they were created, 1 = reversed, 2 = shuffled (seed)
// spread > 1: spread x as many chunks are created as the arena needs and a random subset of them is kept (the rest is released
// again), so that the arena's physical pages are scattered over a range spread x its size
extern "C" void* vmm_arena_create_spread(size_t bytes, size_t chunk, int order, unsigned seed, size_t va_align, int spread, void** ptr_out);
extern "C" void* vmm_arena_create(size_t bytes, size_t chunk, int order, unsigned seed, size_t va_align, void** ptr_out) {
  return vmm_arena_create_spread(bytes, chunk, order, seed, va_align, 1, ptr_out);
}
extern "C" void* vmm_arena_create_spread(size_t bytes, size_t chunk, int order, unsigned seed, size_t va_align, int spread, void** ptr_out) {
  int dev = 0;
  hipGetDevice(&dev);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  chunk = std::max(chunk, gran);
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t n = (bytes + chunk - 1) / chunk;
  Arena* a = new Arena;
  a->bytes = n * chunk;
  a->chunk = chunk;
  // (the alignment argument of hipMemAddressReserve is not honoured beyond 2 MB here: reserve more and align by hand)
  va_align = std::max<size_t>(va_align, 2u << 20);
  a->reserved_bytes = a->bytes + va_align;
  CK(hipMemAddressReserve(&a->reserved, a->reserved_bytes, 2u << 20, nullptr, 0));
  a->base = (void*)(((uintptr_t)a->reserved + va_align - 1) / va_align * va_align);
  a->handles.resize(n);
  if (spread <= 1) {
    for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&a->handles[i], chunk, &prop, 0));
  } else {
    std::vector<hipMemGenericAllocationHandle_t> all(n * spread);
    for (size_t i = 0; i < all.size(); ++i) CK(hipMemCreate(&all[i], chunk, &prop, 0));
    std::vector<size_t> pick(all.size());
    for (size_t i = 0; i < pick.size(); ++i) pick[i] = i;
    std::mt19937 g(seed * 7919u + 13u);
    std::shuffle(pick.begin(), pick.end(), g);
    std::vector<char> keep(all.size(), 0);
    for (size_t i = 0; i < n; ++i) keep[pick[i]] = 1;
    size_t k = 0;
    for (size_t i = 0; i < all.size(); ++i) {
      if (keep[i]) a->handles[k++] = all[i];
      else CK(hipMemRelease(all[i]));
    }
  }
  std::vector<size_t> slot(n);
  for (size_t i = 0; i < n; ++i) slot[i] = i;
  if (order == 1) std::reverse(slot.begin(), slot.end());
  if (order == 2) { std::mt19937 g(seed); std::shuffle(slot.begin(), slot.end(), g); }
  for (size_t i = 0; i < n; ++i) CK(hipMemMap((char*)a->base + slot[i] * chunk, chunk, 0, a->handles[i], 0));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(a->base, a->bytes, &acc, 1));
  CK(hipMemset(a->base, 0, a->bytes));
  CK(hipDeviceSynchronize());
  *ptr_out = a->base;
  return a;
}

extern "C" void vmm_arena_destroy(void* h) {
  Arena* a = static_cast<Arena*>(h);
  if (!a) return;
  hipDeviceSynchronize();
  hipMemUnmap(a->base, a->bytes);
  for (auto& hd : a->handles) hipMemRelease(hd);
  hipMemAddressFree(a->reserved, a->reserved_bytes);
  delete a;
}

// physically contiguous device memory (hipExtMallocWithFlags, hipDeviceMallocContiguous); flags = 0: a plain allocation
extern "C" void* flagged_alloc(size_t bytes, unsigned flags) {
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, flags);
  if (e != hipSuccess) { std::fprintf(stderr, "flagged_alloc(%zu, %u): %s\n", bytes, flags, hipGetErrorString(e)); return nullptr; }
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
  return p;
}
extern "C" void flagged_free(void* p) { (void)hipFree(p); }
