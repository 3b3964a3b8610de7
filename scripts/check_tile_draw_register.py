#!/usr/bin/env python3
"""Build-time guard for the tile draw of gemm_wide_kernel (swinvox_amd/csrc/igemm.hip).

Producer wave 0 draws the next tile with an atomic hipcc must not count (a counted one would drain the LDS-DMA queue); its return
lands in a FIXED register, v167, that inline asm writes and reads across separate statements.  Nothing in the language reserves the
register in between - the kernel is only correct while the COMPILER never touches v167 inside that kernel (a duplicated tile once
corrupted the BatchNorm statistics, commit cadc99b).  This script makes that a checked property of the shipped binary:

  * disassembles every gfx950 code object of libswinvox_hip.so (llvm-objdump),
  * in every `gemm_wide_kernel` instantiation requires that each instruction naming v167 - directly or inside a register range - is one
    of the three hand-written forms (sentinel `v_mov_b32 v167, -1`, the draw `global_atomic_add v167, ...`, the read
    `v_mov_b32 vN, v167`), that all three occur, and that the kernel's VGPR budget still ends at v167,
  * exits non-zero otherwise (the Makefile runs it after linking; tests/test_cpu_oracle_and_abi.py runs it on the built library).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = 167

_single = re.compile(r"\bv(\d+)\b")
_range = re.compile(r"\bv\[(\d+):(\d+)\]")


def names_reg(text: str) -> bool:
    if any(int(m.group(1)) == REG for m in _single.finditer(text)):
        return True
    return any(int(m.group(1)) <= REG <= int(m.group(2)) for m in _range.finditer(text))


def allowed(ins: str) -> str:
    ins = ins.strip()
    if re.fullmatch(r"v_mov_b32(_e32)? v167, -1", ins):
        return "sentinel"
    if re.fullmatch(r"global_atomic_add v167, v\[\d+:\d+\], v\d+, off sc0", ins):
        return "draw"
    if re.fullmatch(r"v_mov_b32(_e32)? v\d+, v167", ins):
        return "read"
    return ""


def check(lib: str) -> list:
    problems, seen_kernels = [], 0
    tmp = tempfile.mkdtemp(prefix="sv_draw_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(f for f in os.listdir(tmp) if "gfx950" in f)
        for f in objs:
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur, kinds = None, None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    if cur is not None and kinds != {"sentinel", "draw", "read"}:
                        problems.append(f"{cur}: expected the three hand-written v167 forms, found {sorted(kinds)}")
                    cur = m.group(1) if "gemm_wide_kernel" in m.group(1) else None
                    kinds = set()
                    seen_kernels += cur is not None
                    continue
                if cur is None:
                    continue
                ins = line.split("//")[0].strip()
                if not ins or not names_reg(ins):
                    continue
                kind = allowed(ins)
                if kind:
                    kinds.add(kind)
                else:
                    problems.append(f"{cur}: v{REG} used outside the tile-draw asm: `{ins}`")
            if cur is not None and kinds != {"sentinel", "draw", "read"}:
                problems.append(f"{cur}: expected the three hand-written v167 forms, found {sorted(kinds)}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if seen_kernels == 0:
        problems.append("no gemm_wide_kernel instantiation found in the library")
    return problems


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "swinvox_amd", "libswinvox_hip.so")
    problems = check(lib)
    for p in problems:
        print("check_tile_draw_register:", p, file=sys.stderr)
    if problems:
        sys.exit(1)
    print("check_tile_draw_register: ok (v167 only in the tile-draw asm of every gemm_wide_kernel)")


if __name__ == "__main__":
    main()
