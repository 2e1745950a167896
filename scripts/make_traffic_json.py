#!/usr/bin/env python3
"""profiles/<tag>_summary*.txt (scripts/profile_gpu.sh) -> profiles/traffic_latest.json, which bench.py reads for roofline.traffic.
usage: make_traffic_json.py <summary.txt> <cmd.txt> <streams> <block_bytes> <config>"""
import json, re, sys
summary, cmd, streams, block, config = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
vals = {}
for line in open(summary):
    m = re.match(r"(.{60}) (FETCH_SIZE|WRITE_SIZE)\s+avg=(\S+) n=(\d+)", line)
    if not m:
        continue
    name = m.group(1).strip()
    short = re.sub(r"^void ", "", name).split("<")[0].split("(")[0].replace("divans_hip::", "")
    if "divans" not in name:
        continue
    vals.setdefault(short, {})[m.group(2)] = float(m.group(3)) * 1024.0      # counters are in KiB
out = {"command": open(cmd).read().strip(), "streams": streams, "block_bytes": block, "config": config,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB), separate passes, per-dispatch average; narrow (2-byte/lane) "
               "accesses: the guide's x2 FETCH_SIZE correction is calibrated for 16 B/lane streams only, so the raw counter is reported",
       "kernels": {k: {"fetch_bytes": v.get("FETCH_SIZE"), "write_bytes": v.get("WRITE_SIZE"),
                       "hbm_bytes_per_launch": (v.get("FETCH_SIZE") or 0) + (v.get("WRITE_SIZE") or 0)} for k, v in vals.items()}}
json.dump(out, open("profiles/traffic_latest.json", "w"), indent=1)
print(json.dumps(out, indent=1))
