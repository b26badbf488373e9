"""Several GPUs behind the C ABI in one process (csm_group): contiguous blocks,
one host thread per member, one exchange of the records. On this one-GPU box
the members are: one on GPU 0; two on GPU 0 (threads + host-staged exchange);
one on GPU 0 with the exchange forced through RCCL (a one-rank communicator:
dlopen, ncclCommInitAll, grouped ncclAllGather). Each must return exactly what a
plain context returns, and every member's gathered device buffer must hold all
records in query order."""
import math
import os

import numpy as np
import pytest

from csm_hip import api, parallel, synth

pytestmark = pytest.mark.gpu
PARAMS = (2.5, 2.5, 0.5, 2, 0.45, 0.55)


def _batch(n):
    rng = np.random.RandomState(21)
    queries, grids = [], {}
    for i in range(n):
        c = synth.csm_case(9000 + i, n_beams=720, fov=1.5 * math.pi)
        init = tuple(np.asarray(c["truth"]) + rng.uniform(-0.5, 0.5, 3) * (1, 1, 0.12))
        grids[i] = c["grid"]
        queries.append(dict(map_id=i, geom=c["geom"], angles=c["angles"], ranges=c["ranges"],
                            rel_pose=(0.0, 0.0, 0.0), init_pose=init))
    return queries, grids


@pytest.fixture(scope="module")
def reference():
    queries, grids = _batch(13)
    ctx = api.Context(0)
    for k, g in grids.items():
        ctx.upload_grid(k, g)
    outs = ctx.bnb_match_batch(queries, *PARAMS)
    ctx.close()
    assert any(o["pose_found"] for o in outs)
    return queries, grids, outs


@pytest.mark.parametrize("devices,force_rccl", [([0], False), ([0, 0], False), ([0, 0, 0], False), ([0], True)])
def test_group_equals_single_context(reference, devices, force_rccl):
    queries, grids, want = reference
    grp = api.Group(devices, force_rccl=force_rccl)
    assert len(grp.members) == len(devices)
    used_rccl, _ = grp.exchange_info()
    assert used_rccl == force_rccl
    grp.upload_grids(queries, grids)
    for _ in range(2):                       # a second call reuses buffers and communicators
        got = grp.bnb_match_batch(queries, *PARAMS)
    drop = ("input_setup_us", "optimization_us")
    for g, w in zip(got, want):
        assert {k: v for k, v in g.items() if k not in drop} == {k: v for k, v in w.items() if k not in drop}
    # every member's device buffer: all records, block by block, in query order
    want_rec = parallel.records_to_bytes([w["raw"] for w in want])
    n, m = len(queries), len(devices)
    for k in range(m):
        buf = grp.gathered_records(k)
        assert buf.shape[0] == m
        for r in range(m):
            lo, hi = api.host_shard_bounds(n, r, m)
            assert np.array_equal(buf[r, :hi - lo], want_rec[lo:hi]), (k, r)
            assert not buf[r, hi - lo:].any()            # padding of the shorter blocks
    used_rccl, gather_us = grp.exchange_info()
    assert gather_us > 0
    grp.close()


def test_group_reports_a_missing_map(reference):
    queries, grids, _ = reference
    grp = api.Group([0, 0])
    # maps uploaded to member 0 only: member 1's block cannot run
    for q in queries:
        grp.members[0].upload_grid(q["map_id"], grids[q["map_id"]])
    with pytest.raises(api.CsmError) as e:
        grp.bnb_match_batch(queries, *PARAMS)
    assert e.value.code == -2 and "member 1" in str(e.value)
    grp.close()
