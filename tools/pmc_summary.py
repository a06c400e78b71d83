"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel (and the launch count)."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", name)  # mangled name of a kernel in an anonymous namespace
    if m:
        n = int(m.group(1))
        rest = name[m.end():]
        return rest[:n]
    depth, out = 0, []
    for ch in name:  # cut at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)[-80:]


for d in sys.argv[1:]:
    for path in sorted(glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")):
        print("#", path)
        per = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(path)):
            per[(short(r["Kernel_Name"]), r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        acc = defaultdict(lambda: defaultdict(list))
        for (k, _), cs in per.items():
            for c, v in cs.items():
                acc[k][c].append(v)
        for k, cs in acc.items():
            if "amd_rocclr" in k:
                continue
            print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
