#!/usr/bin/env python3
"""Build gate for every HIP source of csrc/: no kernel may spill a vector register.

ROCm 7.2's hipcc was seen (round 2, unit_fast_kernel at 64 registers; the listing was not kept) placing VGPR spill stores in FRONT of
the exec-mask restore of a join block -- `scratch_store_dword` of a per-lane value, then `s_or_b64 exec, exec, ...` -- so lanes that
were masked off inside the branch reload garbage later -- a kernel that spills VGPRs inside divergent control flow is not merely
slower, it can be wrong.  Scratch that comes from a per-thread ARRAY (the synthetic-corpus generators keep small arrays) is not a
spill and is allowed; `VGPRs Spill` above zero is not.  SGPR spills go to VGPR lanes and are reported, not refused.

usage: check_kernel_spills.py <stderr of hipcc -Rpass-analysis=kernel-resource-usage> [source name for the message]
"""
import re
import sys

name, bad, seen, sgpr = None, [], 0, []
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        seen += 1
        continue
    m = re.search(r"VGPRs Spill: (\d+)", line)
    if m and name and int(m.group(1)) > 0:
        bad.append((name, int(m.group(1))))
    m = re.search(r"SGPRs Spill: (\d+)", line)
    if m and name and int(m.group(1)) > 0:
        sgpr.append((name, int(m.group(1))))
    if " error: " in line or "warning:" in line:
        sys.stderr.write(line)
src = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
if bad:
    for n, b in bad:
        sys.stderr.write(f"{src}: kernel {n} spills {b} VGPRs: see tools/check_kernel_spills.py\n")
    sys.exit(1)
print(f"spill check {src}: {seen} functions, no VGPR spills" + (f"; SGPR spills (to VGPR lanes): {sgpr}" if sgpr else ""))
