"""CPU: the torchrun-free launcher and its file rendezvous (loam_velodyne_amd/launch.py) with two ranks — the RCCL communicator
itself is replaced by a recorder (tests/dist_worker.py, mode "fake"); the collectives run on a GPU box in tests/test_gpu_dist.py."""
import json
import os

from conftest import ROOT
from loam_velodyne_amd import launch


def test_two_ranks_rendezvous_through_a_file(tmp_path):
    rc = launch.main(["--nproc", "2", os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "fake"])
    assert rc == 0
    r = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert [x["rank"] for x in r] == [0, 1] and all(x["world"] == 2 for x in r)
    assert r[0]["uid"] == r[1]["uid"] and len(bytes.fromhex(r[0]["uid"])) == 128      # rank 1 read exactly what rank 0 wrote
    assert [x["device"] for x in r] == [0, 1]                                          # one GPU per process


def test_shard_rule_matches_the_survey():
    """GPU g of G takes sweeps [g*B/G, (g+1)*B/G): the same rule the C-ABI implements (api_dist.hip)"""
    for B, G in ((32, 4), (64, 8), (10, 4), (3, 8)):
        cuts = [(g * B // G, (g + 1) * B // G) for g in range(G)]
        assert cuts[0][0] == 0 and cuts[-1][1] == B and all(cuts[i][1] == cuts[i + 1][0] for i in range(G - 1))
