#!/usr/bin/env python3
"""Parse rocprofv3 --pmc CSV output (counter_collection.csv) -> per-kernel average counter value."""
import csv
import glob
import sys
from collections import defaultdict


def load(dirname):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            out[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


if __name__ == "__main__":
    for d in sys.argv[1:]:
        for k, cs in load(d).items():
            for c, v in cs.items():
                print("%s | %s | n=%d avg=%.1f min=%.1f max=%.1f" % (k[:70], c, len(v), sum(v) / len(v), min(v), max(v)))
