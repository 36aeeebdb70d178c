"""FETCH_SIZE / WRITE_SIZE (KB, averaged per launch) of the kernels whose name contains argv[3], from two rocprofv3 --pmc output
directories (argv[1]: FETCH_SIZE pass, argv[2]: WRITE_SIZE pass) -> JSON with the gfx950 correction applied (FETCH_SIZE x 2)."""
import csv
import glob
import json
import sys


def avg(d, counter, pat):
    v = [float(r["Counter_Value"]) for f in glob.glob(d + "/*/*counter_collection.csv") for r in csv.DictReader(open(f))
         if r["Counter_Name"] == counter and pat in r["Kernel_Name"]]
    return (sum(v) / len(v), len(v)) if v else (None, 0)


f, nf = avg(sys.argv[1], "FETCH_SIZE", sys.argv[3])
w, nw = avg(sys.argv[2], "WRITE_SIZE", sys.argv[3])
print(json.dumps(dict(kernel_pattern=sys.argv[3], FETCH_SIZE_KB=f, WRITE_SIZE_KB=w, launches=[nf, nw],
                      hbm_bytes_per_launch=None if f is None or w is None else 1024.0 * (2.0 * f + w))))
