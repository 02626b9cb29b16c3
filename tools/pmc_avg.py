"""Average PMC counter values per kernel from rocprofv3 --pmc output directories:
    python tools/pmc_avg.py <kernel substring> dir1 [dir2 ...]"""
import glob
import os
import sys

import pandas as pd


def main():
    key = sys.argv[1]
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            t = pd.read_csv(f)
            t = t[t.Kernel_Name.str.contains(key, regex=False)]
            g = t.groupby("Counter_Name").Counter_Value.mean()
            for k, v in g.items():
                print("%-32s %16.1f  (n=%d)" % (k, v, (t.Counter_Name == k).sum()))


if __name__ == "__main__":
    main()
