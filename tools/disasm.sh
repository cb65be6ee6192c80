#!/bin/bash
# Disassemble the gfx950 code object inside a built .o of pixray_amd/csrc:   tools/disasm.sh gemmfit [outdir]
# Writes <outdir>/<name>.s (llvm-objdump -d, demangled) and <name>.notes (per-kernel VGPR / SGPR / scratch / LDS metadata).
set -e
name=$1; out=${2:-/tmp/isa}
R=$(cd "$(dirname "$0")/.." && pwd)
L=/opt/rocm/lib/llvm/bin
mkdir -p $out
$L/llvm-objcopy --dump-section=.hip_fatbin=$out/$name.fatbin $R/pixray_amd/csrc/$name.o
t=$($L/clang-offload-bundler --list --type=o --input=$out/$name.fatbin | grep gfx950)
$L/clang-offload-bundler --unbundle --type=o --input=$out/$name.fatbin --targets=$t --output=$out/$name.co
$L/llvm-objdump -d -C $out/$name.co > $out/$name.s
$L/llvm-readelf --notes $out/$name.co > $out/$name.notes
echo "$out/$name.s: $(wc -l < $out/$name.s) lines"
