"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel name."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel".ljust(42) + "".join(n[-18:].rjust(20) for n in names))
for k in sorted(acc):
    print(k.ljust(42) + "".join(("%.6g" % (sum(acc[k][n]) / len(acc[k][n])) if acc[k][n] else "-").rjust(20) for n in names))
