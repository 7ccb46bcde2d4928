"""GPU: loamx_dist_* (RCCL) — the two exchanges of the batched mode: map broadcast feeding the double-buffered
index build through its event, and the all-gather of the results.  One rank on a one-GPU box; two ranks (one process per GPU,
started by loam_velodyne_amd.launch) wherever two GPUs are visible."""
import os

import numpy as np
import pytest

from conftest import ROOT
from loam_velodyne_amd import launch, loamx

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _run(nproc, tmp_path, batch=4):
    os.environ["LOAMX_TEST_BATCH"] = str(batch)
    try:
        assert launch.main(["--nproc", str(nproc), WORKER, str(tmp_path), "gpu"]) == 0
    finally:
        os.environ.pop("LOAMX_TEST_BATCH", None)
    return [np.load(tmp_path / f"rank{k}.npz") for k in range(nproc)]


def test_one_rank(tmp_path):
    (r,) = _run(1, tmp_path)
    assert r["poses"].shape == (4, 6) and r["shard"].tolist() == [0, 4]
    assert np.all(r["flags"][:, 0] >= 1) and np.all(r["flags"][:, 1] >= 50)          # iterations, rows selected
    assert np.isfinite(r["poses"]).all()


@pytest.mark.skipif(loamx.device_count() < 2, reason="needs two GPUs (one process per GPU)")
def test_two_ranks_equal_one_rank(tmp_path):
    (tmp_path / "one").mkdir()
    (tmp_path / "two").mkdir()
    one = _run(1, tmp_path / "one")[0]
    two = _run(2, tmp_path / "two")
    assert two[0]["shard"].tolist() == [0, 2] and two[1]["shard"].tolist() == [2, 4]
    assert two[0]["map_sum"] == two[1]["map_sum"] == one["map_sum"]                  # the broadcast reached rank 1
    for r in two:                                                                   # every rank holds all results, in batch order,
        assert np.array_equal(r["poses"], one["poses"]) and np.array_equal(r["flags"], one["flags"])   # identical to the unsharded run


@pytest.mark.skipif(loamx.device_count() < 2, reason="needs two GPUs (one process per GPU)")
@pytest.mark.parametrize("batch,shards", [(5, [[0, 2], [2, 5]]), (1, [[0, 0], [0, 1]])])
def test_two_ranks_unequal_shards(tmp_path, batch, shards):
    """a batch that does not divide by the world size (and one smaller than it: an empty shard): the result exchange pads to the
    longest shard, every rank takes part, and every rank ends up with the unsharded run's results in batch order"""
    (tmp_path / "one").mkdir()
    (tmp_path / "two").mkdir()
    one = _run(1, tmp_path / "one", batch)[0]
    two = _run(2, tmp_path / "two", batch)
    assert [r["shard"].tolist() for r in two] == shards
    for r in two:
        assert r["poses"].shape == (batch, 6)
        assert np.array_equal(r["poses"], one["poses"]) and np.array_equal(r["flags"], one["flags"])
