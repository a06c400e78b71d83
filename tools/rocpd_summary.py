"""Summarise rocprofv3 rocpd (.db) output: per-kernel stats (the --stats table) and per-kernel mean
PMC counter values.  usage: rocpd_summary.py stats|pmc <results.db> ..."""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][-70:]


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, (end - start) from kernels").fetchall()
    acc = defaultdict(list)
    for n, d in rows:
        acc[short(n)].append(d)
    total = sum(sum(v) for v in acc.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(f'"{n}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / total:.2f},{min(v)},{max(v)}')


def pmc(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    rows = db.execute("select * from counters_collection").fetchall()
    ik, ic, iv = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
    idisp = cols.index("dispatch_id") if "dispatch_id" in cols else None
    per = defaultdict(lambda: defaultdict(float))
    for r in rows:  # a counter can come as one row per dimension instance: sum them per dispatch
        per[(short(r[ik]), r[idisp] if idisp is not None else 0)][r[ic]] += float(r[iv])
    acc = defaultdict(lambda: defaultdict(list))
    for (k, _), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
    for k, cs in acc.items():
        if "rocclr" in k:
            continue
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))


if __name__ == "__main__":
    mode = sys.argv[1]
    for p in sys.argv[2:]:
        print("#", p)
        (stats if mode == "stats" else pmc)(p)
