#!/usr/bin/env python3
"""Build gate for csrc/sann_fast.hip: no instantiation of unit_fast_kernel may use scratch memory.

ROCm 7.2's hipcc places VGPR spill stores in FRONT of the exec-mask restore of a join block (seen in the ISA of a
64-register build of unit_fast_kernel<256, 6>: `scratch_store_dword` of a per-lane value, then `s_or_b64 exec, exec, ...`),
so lanes that were masked off inside the branch reload garbage later.  The kernel is full of divergent control flow;
a build that spills is therefore not merely slower, it is wrong (and was: nondeterministic duplicate handling).
The launch bounds and the kernel's register diet are chosen so that nothing spills; this script keeps it that way.

usage: check_unit_kernel_resources.py <stderr of hipcc -Rpass-analysis=kernel-resource-usage> [kernel name, default unit_fast_kernel]
"""
import re
import sys

name = None
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "unit_fast_kernel"
bad, seen = [], 0
regs = {}
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        continue
    m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
    mv = re.search(r"VGPRs: (\d+)", line)
    if mv and name and KERNEL in name and "AGPR" not in line:
        regs[name] = int(mv.group(1))
    if m and name and KERNEL in name:
        seen += 1
        if int(m.group(1)) != 0:
            bad.append((name, int(m.group(1))))
    if "error:" in line:
        sys.stderr.write(line)
if seen == 0:
    sys.exit(f"check_unit_kernel_resources: no {KERNEL} instantiation found in the log")
if bad:
    for n, b in bad:
        sys.stderr.write(f"unit kernel {n} spills ({b} bytes of scratch per lane): see tools/check_unit_kernel_resources.py\n")
    sys.exit(1)
print(f"unit kernel resource check ({KERNEL}): {seen} instantiations, no scratch; VGPRs {sorted(set(regs.values()))}")
