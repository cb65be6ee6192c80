#!/bin/bash
# The emulated kernels under AddressSanitizer and -fsanitize=bounds: builds the two checking variants of libprx_emu.so and runs the
# emulator suite plus the random-shape sweeps on them (about 20 minutes on 8 cores).  From the repo root:  bash tools/hipemu/check.sh
set -u
cd "$(dirname "$0")"
make -s -j"$(nproc)" B=_build_san OUT=libprx_emu_san.so SAN="-fsanitize=bounds -fsanitize-trap=bounds" || exit 1
make -s -j"$(nproc)" B=_build_asan OUT=libprx_emu_asan.so SAN="-fsanitize=address -shared-libasan -fno-omit-frame-pointer" \
     SANLINK="-fsanitize=address -shared-libasan" || exit 1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd ../..
echo "== -fsanitize=bounds"
HIPEMU_LIB=$PWD/tools/hipemu/libprx_emu_san.so HIPEMU_TRACE=1 python -m pytest tests/test_emu_cpu.py -x -q -p no:cacheprovider 2>&1 | tail -2
asan() { LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:alloc_dealloc_mismatch=0:new_delete_type_mismatch=0 \
         HIPEMU_LIB=$PWD/tools/hipemu/libprx_emu_asan.so "$@" 2>&1 | grep -v "^    #[1-9][0-9]" | tail -12; }
echo "== AddressSanitizer: emulator suite"
asan python -m pytest tests/test_emu_cpu.py -x -q -p no:cacheprovider
for k in "gemm 3 400" "conv 3 400" "gn 3 300" "vqgan 3 12" "runners 3 12" "vit 3 5"; do
    echo "== AddressSanitizer: sweep $k"
    asan python tests/_emu_fuzz.py $k
done
