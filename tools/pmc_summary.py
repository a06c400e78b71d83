"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for path in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "amd_rocclr" in k:
                continue
            print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
