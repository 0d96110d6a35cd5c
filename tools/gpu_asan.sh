#!/bin/bash
# The host side of libippmarl.so (argument checks, launch set-up, event pools, counters, error strings) under AddressSanitizer
# (SURVEY section 5's suggestion).  Built in the build container:
#   make -C ipp-marl_amd/csrc VARIANT=asan EXTRA="-fsanitize=address -shared-libasan -g -Wno-option-ignored"
#   hipcc -O1 -g -fsanitize=address -shared-libasan --offload-arch=gfx950 -Iinclude tools/probe/abi_asan_smoke.cpp \
#         -Lipp-marl_amd/lib -lippmarl_asan -Wl,-rpath,'$ORIGIN/../../ipp-marl_amd/lib' -o tools/probe/abi_asan_smoke
# (device code is not instrumented: gfx950 without xnack; the harness is C++ because an instrumented runtime preloaded under the Python
# interpreter + torch + the HIP runtime does not get past import).  $1 = tag
OUT=gpurun_out/${1:-asan}; mkdir -p $OUT
export LD_LIBRARY_PATH=$(dirname $(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)):$LD_LIBRARY_PATH
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:allocator_may_return_null=1:max_allocation_size_mb=4194304
python tools/write_config_image.py small $OUT/cfg_small.bin
( timeout 300 tools/probe/abi_asan_smoke $OUT/cfg_small.bin 6 2>&1 | tail -25 ) | tee $OUT/asan_host_shim.txt
python tools/write_config_image.py c2 $OUT/cfg_c2.bin
( timeout 300 tools/probe/abi_asan_smoke $OUT/cfg_c2.bin 8 2>&1 | tail -8 ) | tee -a $OUT/asan_host_shim.txt
