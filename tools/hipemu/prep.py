"""Source preparation for the host build of pixray_amd/csrc (tools/hipemu/Makefile): copies every .hip / .h / .inc into
_build/src with the handful of textual changes the host compiler needs.  The product sources are never modified.

  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EXPR) : "memory")   ->  hipemu::waitcnt_vm(EXPR)       counted wait for this wave's DMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory")               ->  hipemu::waitcnt_vm(0)
  asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")          ->  hipemu::waitcnt_vm(n)          (macro argument, gemm8p.h)
  any other asm volatile( ... )                                 ->  HIPEMU_ASM( ... )              (lgkmcnt waits, register pins: no-ops)
  /*hipemu:wave_sync*/                                          ->  hipemu::wave_sync();           lockstep exchange inside one wave
  __attribute__((amdgpu_waves_per_eu(a, b)))                    ->  (dropped)
  "../../include/prx.h"                                         ->  the path from _build/src
"""
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
rules = [
    (re.compile(r'asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\((.*)\) : "memory"\)'), r'hipemu::waitcnt_vm(\1)'),
    (re.compile(r'asm volatile\("s_waitcnt vmcnt\(0\)" ::: "memory"\)'), 'hipemu::waitcnt_vm(0)'),
    (re.compile(r'asm volatile\("s_waitcnt vmcnt\(" #(\w+) "\)" ::: "memory"\)'), r'hipemu::waitcnt_vm(\1)'),
    (re.compile(r'asm volatile\('), 'HIPEMU_ASM('),
    (re.compile(r'/\*hipemu:wave_sync\*/'), 'hipemu::wave_sync();'),
    (re.compile(r'__attribute__\(\(amdgpu_waves_per_eu\([0-9, ]*\)\)\)'), ''),
    (re.compile(r'"\.\./\.\./include/prx\.h"'), '"../../../../include/prx.h"'),
]
for name in sorted(os.listdir(src)):
    if not name.endswith((".hip", ".h", ".inc")):
        continue
    text = open(os.path.join(src, name)).read()
    for rx, rep in rules:
        text = rx.sub(rep, text)
    assert "asm volatile" not in text, name
    out = os.path.join(dst, name)
    if not os.path.exists(out) or open(out).read() != text:
        open(out, "w").write(text)
