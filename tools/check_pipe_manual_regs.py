#!/usr/bin/env python3
"""Build gate for csrc/sann_pipe.hip: the registers that hold loads in flight belong to the hand-written asm alone.

unit_pipe_kernel is compiled with amdgpu_num_vgpr(96): the register allocator owns v0 .. v95, and v96 .. v102 (stage D's
descriptor data) and v104 .. v127 (stage P's postings) are written by `global_load_*` statements that name them literally
and read only by the `v_mov_b32` that follow the matching hand-written `s_waitcnt`.  Anything else the compiler did with
one of them -- a copy, a spill slot, an operand -- would read or overwrite a value that has not arrived yet.  (v103 is left
out of the layout on purpose: hipcc parks spilled SGPRs in the last register of the allocation granule above its limit.)

usage: check_pipe_manual_regs.py <device assembly of sann_pipe.hip (hipcc -S --cuda-device-only)>
"""
import re
import sys

MANUAL = set(range(96, 103)) | set(range(104, 128))
reg_single = re.compile(r"\bv(\d+)\b")
reg_tuple = re.compile(r"\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in reg_tuple.finditer(text):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in reg_single.finditer(text):
        out.add(int(m.group(1)))
    return out


kernel, bad, n_kernels, n_loads, n_moves = None, [], 0, 0, 0
for ln, line in enumerate(open(sys.argv[1], errors="replace"), 1):
    m = re.match(r"^(_ZN4sann16unit_pipe_kernel\w+):", line)
    if m:
        kernel = m.group(1)
        n_kernels += 1
        continue
    if kernel is None:
        continue
    code = line.split(";")[0].strip()
    if code.startswith("s_endpgm"):
        kernel = None
        continue
    if not code or code.startswith(".") or code.endswith(":"):
        continue
    used = regs_of(code) & MANUAL
    if not used:
        continue
    op, _, rest = code.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if op.startswith("global_load_dword") and regs_of(ops[0]) <= MANUAL and not (regs_of(",".join(ops[1:])) & MANUAL):
        n_loads += 1  # a hand-written load INTO manual registers, addressed from the compiler's
        continue
    if op.startswith("v_mov_b32") and len(ops) == 2 and not (regs_of(ops[0]) & MANUAL) and regs_of(ops[1]) <= MANUAL:
        n_moves += 1  # the copy behind a hand-written wait
        continue
    bad.append(f"{kernel} line {ln}: {code}")
if n_kernels == 0:
    sys.exit("check_pipe_manual_regs: no unit_pipe_kernel in the assembly")
if bad:
    sys.stderr.write("registers reserved for loads in flight are used by compiler-generated code:\n  " + "\n  ".join(bad[:20]) + "\n")
    sys.exit(1)
print(f"pipe kernel manual-register check: {n_kernels} kernels, {n_loads} loads into and {n_moves} copies out of v96-v102 / v104-v127, nothing else touches them")
