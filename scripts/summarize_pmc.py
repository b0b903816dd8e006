"""Average rocprofv3 --pmc counters of one kernel over its last dispatches.

usage: summarize_pmc.py OUT.csv KERNEL DIR [DIR ...]
Each DIR is the -d directory of one `rocprofv3 --pmc ... --kernel-trace` pass; counters from all passes are merged
into one two-column csv (counter, mean per dispatch over the last 10 dispatches of KERNEL).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    out, kernel, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    rows = []
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(dict)          # counter -> dispatch -> value (summed over XCD/SE instances)
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if not r["Kernel_Name"].startswith(kernel):
                        continue
                    c, i = r["Counter_Name"], int(r["Dispatch_Id"])
                    per[c][i] = per[c].get(i, 0.0) + float(r["Counter_Value"])
            for c, dv in per.items():
                last = [dv[i] for i in sorted(dv)[-(12 if kernel == 'g1_env_kernel' else 10):]]   # two steps of the split G1 pipeline: 12 env launches
                rows.append((c, sum(last) / len(last)))
    with open(out, "w") as fh:
        fh.write("counter,mean_per_dispatch_over_last_10_dispatches\n")
        for c, v in rows:
            fh.write("%s,%s\n" % (c, v))
    print(open(out).read())


if __name__ == "__main__":
    main()
