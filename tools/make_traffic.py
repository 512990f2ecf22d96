#!/usr/bin/env python
"""profiles/traffic.json (what bench.py's roofline.traffic reads) from a committed tools/summarize_profile.py summary:
    python tools/make_traffic.py profiles/r02d/bench_b64_608_bf16_summary.json
Kernels are listed by total time, so that the first name matching a prefix is the variant most of the time goes to."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
d = json.load(open(os.path.join(ROOT, src)))
ks = sorted(d["kernels"].items(), key=lambda kv: -kv[1].get("total_ms", 0.0))
out = {"source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, of `bench.py --steps 3 --warmup 1`; "
                 "FETCH_SIZE x2 gfx950 correction, KiB->bytes)" % src,
       "hbm_bytes_per_launch": {k: int(round(v["hbm_bytes_per_launch"])) for k, v in ks if v.get("hbm_bytes_per_launch")}}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
