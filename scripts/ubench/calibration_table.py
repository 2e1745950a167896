#!/usr/bin/env python3
"""Counter calibration table: the rocprofv3 --pmc passes over scripts/ubench/counter_calibration (one csv per pass) against the
request / byte counts the program printed.  usage: calibration_table.py <program_stdout.csv> <pmc csv> [<pmc csv> ...]"""
import collections
import csv
import re
import sys

known = {}
for row in csv.DictReader(open(sys.argv[1])):
    known[row["kernel"]] = {k: int(v) for k, v in row.items() if k != "kernel"}


def short(kernel_name):
    m = re.search(r"rows_kernel<(\d+), (\d+)>", kernel_name)
    if m:
        return "row_%s_mask%s" % (("read", "write", "rmw")[int(m.group(2))], m.group(1))
    m = re.search(r"rmw_private<(\d+)>", kernel_name)
    if m:
        return "rmw_private_%s" % m.group(1)
    for k in ("stream_read16", "stream_read1", "stream_write16"):
        if re.search(r"\b%s\b" % k, kernel_name):
            return k
    return None


vals = collections.defaultdict(dict)
for path in sys.argv[2:]:
    for row in csv.DictReader(open(path)):
        k = short(row.get("Kernel_Name", ""))
        if k:
            vals[k][row["Counter_Name"]] = vals[k].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])

names = sorted({c for v in vals.values() for c in v})
print("per kernel: known counts, raw counters, and the ratios that calibrate them")
for k, kn in known.items():
    v = vals.get(k, {})
    print("== %s: streamed %d B, %d lines of 128 B, %d halves of 64 B, %d rows of 32 B" % (k, kn["bytes_streamed"], kn["lines_128B"], kn["halves_64B"], kn["rows_32B"]))
    for c in names:
        if c not in v:
            continue
        x = v[c]
        line = "   %-24s %.6g" % (c, x)
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            b = x * 1024.0
            line += "  KiB = %.4g B" % b
            if kn["bytes_streamed"]:
                line += "  = %.3f x bytes streamed" % (b / kn["bytes_streamed"])
            line += "  = %.1f B per 128-byte line touched, %.1f B per 32-byte row touched" % (b / kn["lines_128B"], b / kn["rows_32B"])
        else:
            line += "  = %.3f per line, %.3f per 64-byte half, %.3f per row" % (x / kn["lines_128B"], x / kn["halves_64B"], x / kn["rows_32B"])
        print(line)
