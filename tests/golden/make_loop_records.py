#!/usr/bin/env python3
"""Generates tests/golden/loop_records.json ON A GPU BOX: the 48-byte csm_result
records that csm_bnb_match_batch (the HIP library, not the oracle) returned
for a small seeded loop-detection batch, hex-encoded, plus the decoded fields.
The CPU (gloo) test of the sharded detector replays these real records through
the exchange so that the record layout that crosses the all-gather is the
library's own. Usage (GPU box): python tests/golden/make_loop_records.py OUT.json
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "my-lidar-graph-slam-v2_amd"))


def main(out_path):
    import __graft_entry__ as ge
    ge.build()
    from csm_hip import api, parallel, synth
    n = 24
    rng = np.random.RandomState(11)
    ctx = api.Context(0)
    queries = []
    for i in range(n):
        c = synth.csm_case(7000 + i, n_beams=1080, fov=1.5 * math.pi)
        ctx.upload_grid(i, c["grid"])
        init = tuple(np.asarray(c["truth"]) + rng.uniform(-0.6, 0.6, 3) * (1, 1, 0.15))
        queries.append(dict(map_id=i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                            rel_pose=(0.0, 0.0, 0.0), init_pose=init))
    outs = ctx.bnb_match_batch(queries, 2.5, 2.5, 0.5, 2, 0.55, 0.6, as_records=True)
    raw = outs.record_bytes().reshape(n, parallel.RECORD_BYTES)
    doc = {"generator": "tests/golden/make_loop_records.py", "library": ctx.lib.csm_version().decode(),
           "params": [2.5, 2.5, 0.5, 2, 0.55, 0.6], "first_seed": 7000, "n": n,
           "records_hex": [bytes(r).hex() for r in raw],
           "decoded": parallel.bytes_to_records(raw)}
    for d in doc["decoded"]:
        d["score"] = float(d["score"]).hex()
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    ctx.close()
    print("wrote", out_path, "found", sum(d["found"] for d in doc["decoded"]))


if __name__ == "__main__":
    main(sys.argv[1])
