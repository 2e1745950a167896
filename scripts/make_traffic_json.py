#!/usr/bin/env python3
"""profiles/<tag>_<config>_summary.txt (scripts/profile_round2.sh) -> profiles/traffic_latest.json, which bench.py reads for roofline.traffic.
usage: make_traffic_json.py <tag> [<config>:<summary.txt>:<cmd.txt>:<streams>:<block_bytes> ...]"""
import json, re, sys
tag = sys.argv[1]
# Calibration (profiles/r05_counter_calibration.txt, scripts/ubench/counter_calibration.hip on this image's rocprofv3 / gfx950): an L2 read miss
# is ONE TCC_EA0_RDREQ that fills the whole 128-byte line whatever the access width -- 16 B/lane streams, 1 B/lane streams and the decoders'
# 16 lanes x 2 B row reads alike (touching 1, 2 or all 4 rows of a line costs 1.00 request) -- and FETCH_SIZE tallies it at 64 bytes, so the bytes
# read are 2 x FETCH_SIZE.  WRITE_SIZE is exact: a 32-byte row store is one 32-byte TCC_EA0_WRREQ, two rows of one 64-byte half one 64-byte request.
FETCH_BYTES_PER_COUNTED_BYTE = 2.0
WRITE_BYTES_PER_COUNTED_BYTE = 1.0
out = {"tag": tag, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB), separate passes, per-dispatch average of the divans kernels.  "
       "hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE: every L2 read miss is one 128-byte line fill that FETCH_SIZE counts as 64 bytes "
       "(calibrated on known request counts for streaming reads AND for the decoders' 2-byte-per-lane row reads, profiles/r05_counter_calibration.txt); "
       "WRITE_SIZE is exact (32- and 64-byte requests).  fetch_bytes / write_bytes are the corrected bytes, *_counter the raw counters.  "
       "Keys: kernel names as rocprofv3 reports them, without namespace and template arguments.",
       "calibration": {"bytes_per_TCC_EA0_RDREQ": 128, "FETCH_SIZE_bytes_counted_per_RDREQ": 64, "bytes_per_TCC_EA0_WRREQ": "32 (64 for _64B requests)",
                       "source": "profiles/r05_counter_calibration.txt"}, "configs": {}}
for spec in sys.argv[2:]:
    config, summary, cmd, streams, block = spec.split(":")
    vals = {}
    for line in open(summary):
        m = re.match(r"(.+?)\s+(FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ_sum|TCC_EA0_WRREQ_sum|TCC_EA0_WRREQ_64B_sum)\s+avg=(\S+) n=(\d+)", line)
        if not m or "divans" not in m.group(1):
            continue
        short = re.sub(r"^void ", "", m.group(1).strip()).split("<")[0].split("(")[0].replace("divans_hip::", "")
        # bench.py names the decode kernel as rocprofv3 does and falls back to this base name (without template arguments) for the lookup
        vals.setdefault(short, {})[m.group(2)] = float(m.group(3)) * (1024.0 if m.group(2).endswith("_SIZE") else 1.0)      # the size counters are in KiB
    out["configs"][config] = {"command": open(cmd).read().strip(), "streams": int(streams), "block_bytes": int(block), "config": config,
                              "kernels": {k: {"fetch_counter": v.get("FETCH_SIZE"), "write_counter": v.get("WRITE_SIZE"),
                                              "fetch_bytes": FETCH_BYTES_PER_COUNTED_BYTE * (v.get("FETCH_SIZE") or 0),
                                              "write_bytes": WRITE_BYTES_PER_COUNTED_BYTE * (v.get("WRITE_SIZE") or 0),
                                              "hbm_bytes_per_launch": FETCH_BYTES_PER_COUNTED_BYTE * (v.get("FETCH_SIZE") or 0) +
                                                                      WRITE_BYTES_PER_COUNTED_BYTE * (v.get("WRITE_SIZE") or 0),
                                              # L2 -> fabric requests (the "ea" pass, where it ran): 128-byte fills, 32- / 64-byte write requests
                                              "rdreq_per_launch": v.get("TCC_EA0_RDREQ_sum"), "wrreq_per_launch": v.get("TCC_EA0_WRREQ_sum"),
                                              "wrreq_64B_per_launch": v.get("TCC_EA0_WRREQ_64B_sum")} for k, v in vals.items()}}
json.dump(out, open("profiles/traffic_latest.json", "w"), indent=1)
print(json.dumps({c: {k: v["hbm_bytes_per_launch"] for k, v in e["kernels"].items()} for c, e in out["configs"].items()}, indent=1))
