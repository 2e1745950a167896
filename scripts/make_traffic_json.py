#!/usr/bin/env python3
"""profiles/<tag>_<config>_summary.txt (scripts/profile_round2.sh) -> profiles/traffic_latest.json, which bench.py reads for roofline.traffic.
usage: make_traffic_json.py <tag> [<config>:<summary.txt>:<cmd.txt>:<streams>:<block_bytes> ...]"""
import json, re, sys
tag = sys.argv[1]
out = {"tag": tag, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB), separate passes, per-dispatch average of the divans kernels; the row "
       "accesses are 2 bytes per lane, for which the guide's x2 FETCH_SIZE correction (calibrated on 16 B/lane streaming reads) does not apply: "
       "raw counters are reported.  Keys: kernel names as rocprofv3 reports them, without namespace and template arguments.", "configs": {}}
for spec in sys.argv[2:]:
    config, summary, cmd, streams, block = spec.split(":")
    vals = {}
    for line in open(summary):
        m = re.match(r"(.+?)\s+(FETCH_SIZE|WRITE_SIZE)\s+avg=(\S+) n=(\d+)", line)
        if not m or "divans" not in m.group(1):
            continue
        short = re.sub(r"^void ", "", m.group(1).strip()).split("<")[0].split("(")[0].replace("divans_hip::", "")
        # bench.py names the decode kernel as rocprofv3 does and falls back to this base name (without template arguments) for the lookup
        vals.setdefault(short, {})[m.group(2)] = float(m.group(3)) * 1024.0      # counters are in KiB
    out["configs"][config] = {"command": open(cmd).read().strip(), "streams": int(streams), "block_bytes": int(block), "config": config,
                              "kernels": {k: {"fetch_bytes": v.get("FETCH_SIZE"), "write_bytes": v.get("WRITE_SIZE"),
                                              "hbm_bytes_per_launch": (v.get("FETCH_SIZE") or 0) + (v.get("WRITE_SIZE") or 0)} for k, v in vals.items()}}
json.dump(out, open("profiles/traffic_latest.json", "w"), indent=1)
print(json.dumps({c: {k: v["hbm_bytes_per_launch"] for k, v in e["kernels"].items()} for c, e in out["configs"].items()}, indent=1))
