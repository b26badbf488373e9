"""CPU suite: the N > 1 path (contiguous sharding + all-gather of 48-byte best
records) on world_size 2 and 3 with the gloo backend. The scorer is a stand-in
(the exchange logic is what is under test; the HIP scorer itself is covered by
the -m gpu tests): a synthetic one, and one that replays the records the HIP
library itself returned for a seeded 24-query batch on an MI355X
(tests/golden/loop_records.json, written by tests/golden/make_loop_records.py)."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from csm_hip import parallel


def _fake_scorer(queries):
    out = []
    for q in queries:
        i = q["map_id"]
        out.append(dict(found=int(i % 3 != 0), best_x=i - 5, best_y=2 * i, best_theta=-i,
                        key=(1 << 33) + i, sum_values=7 * i, known=i, tie_count=1, flags=0,
                        score=0.25 + i / 1024.0))
    return out


def _worker(rank, world, port, n_queries, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        queries = [dict(map_id=i) for i in range(n_queries)]
        det = parallel.LoopDetectorBranchBoundHIP("ld", None, 2.5, 2.5, 0.5, 2, 0.55, 0.6,
                                                  scorer=_fake_scorer)
        records, found = det.detect(queries)
        ret[rank] = (records, found)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_queries", [(2, 8), (2, 7), (3, 8), (2, 1)])
def test_sharded_detect_gathers_in_query_order(world, n_queries):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_queries, ret), nprocs=world, join=True)
    want = _fake_scorer([dict(map_id=i) for i in range(n_queries)])
    for rank in range(world):
        records, found = ret[rank]
        assert records == want
        assert found == [i for i in range(n_queries) if i % 3 != 0]


def test_shard_bounds_cover_exactly_once():
    for n in range(0, 40):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = parallel.shard_bounds(n, r, world)
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def test_record_layout_round_trip():
    recs = _fake_scorer([dict(map_id=i) for i in range(5)])
    b = parallel.records_to_bytes(recs)
    assert b.shape == (5, 48)
    assert parallel.bytes_to_records(b) == recs


# ---- the library's own records through the exchange ----

def _golden():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loop_records.json")
    with open(path) as f:
        return json.load(f)


def _replay_scorer(queries):
    """Returns, as the HIP scorer does, uint8 [m, 48]: the recorded csm_result bytes
    of the queries asked for (map_id = query number)."""
    doc = _golden()
    rows = [np.frombuffer(bytes.fromhex(doc["records_hex"][q["map_id"]]), np.uint8) for q in queries]
    return np.stack(rows) if rows else np.zeros((0, parallel.RECORD_BYTES), np.uint8)


def _replay_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = _golden()["n"]
        det = parallel.LoopDetectorBranchBoundHIP("ld", None, 2.5, 2.5, 0.5, 2, 0.55, 0.6,
                                                  scorer=_replay_scorer)
        ret[rank] = det.detect([dict(map_id=i) for i in range(n)])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_recorded_hip_records_survive_the_exchange(world):
    doc = _golden()
    assert doc["n"] == len(doc["records_hex"]) == len(doc["decoded"])
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_replay_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    want = []
    for d in doc["decoded"]:
        d = dict(d)
        d["score"] = float.fromhex(d["score"])
        want.append(d)
    assert any(d["found"] for d in want) and any(not d["found"] for d in want)
    for rank in range(world):
        records, found = ret[rank]
        assert records == want
        assert found == [i for i, d in enumerate(want) if d["found"]]
        # what crossed the wire is the library's byte layout, bit for bit
        raw = parallel.records_to_bytes(records)
        assert [bytes(r).hex() for r in raw] == doc["records_hex"]
