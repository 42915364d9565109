#!/usr/bin/env python3
"""profiles/traffic_latest.json from the traffic.json of three tools/profile.sh runs (noise, lowsnr, frames):
    python tools/merge_traffic.py <noise dir> [<lowsnr dir> [<frames dir>]]
Top level = the noise workload (what bench.py's headline roofline.committed_traffic reads), "legs" = the record-bearing workloads
(bench.py's per-leg roofline.traffic).  Every part carries the hash of the kernel sources it was measured on."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = json.load(open(os.path.join(sys.argv[1], "traffic.json")))
out["legs"] = {}
for d in sys.argv[2:]:
    t = json.load(open(os.path.join(d, "traffic.json")))
    out["legs"][t.get("workload", os.path.basename(d))] = t
json.dump(out, open(os.path.join(root, "profiles", "traffic_latest.json"), "w"), indent=1)
print("profiles/traffic_latest.json: noise + %s" % ", ".join(out["legs"]))
