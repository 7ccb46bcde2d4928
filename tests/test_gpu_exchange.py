"""GPU: the exchange between the workgroups of one odometry stream (k_odom_lm, BasicLaserOdometry.cpp:484-622 is ONE loop there) rests on
a property of the memory system that no document promises: an aligned 16-byte agent-scope store is observed whole, or split at 8 bytes —
never finer — so a tagged record {tag, value lo, value hi, tag} read with both tags equal to k holds both halves of value k
(csrc/dev_math.hpp: xrec_store / xrec_load).  The poses would drift silently if a ROCm or firmware update broke that; this probe would
not be silent: producer / consumer workgroup pairs on different XCDs hammer shared records and every ACCEPTED read is checked against the
value its tag names."""
import pytest

from loam_velodyne_amd import loamx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pairs,rounds", [(64, 20000), (256, 4000), (1, 50000)])
def test_tagged_records_are_never_accepted_inconsistent(pairs, rounds):
    h = loamx.Batch(1)
    r = h.xrec_stress(pairs, rounds)
    h.close()
    print(f"xrec stress {pairs} pairs x 64 records x {rounds} versions: {r}")
    assert r["timed_out"] == 0, r
    assert r["accepted"] >= 64 * pairs, r          # every consumer thread saw at least the last version
    assert r["inconsistent"] == 0, r               # the property: a record with two matching tags carries the value of that tag
