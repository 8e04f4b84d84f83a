"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name (and grid size)."""
import csv
import collections
import glob
import sys

for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0][-60:]
            if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
                continue
            acc[(name, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path)
    for key, cs in acc.items():
        print(key[0], "grid", key[1], " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(cs.items())))
