"""CPU: executable specification of the bucketed voxel grid (loam_velodyne_amd/csrc/voxbucket.hip) in NumPy, held against the oracle's
pcl::VoxelGrid restatement bit for bit.  What the kernels' correctness argument rests on is checked here without a GPU:

  * buckets are RANGES of the voxel order (iz, iy, ix), so the buckets of a segment one after the other are PCL's output order —
    for the sampled splitters k_vb_plan takes and for ANY other non-decreasing splitters (the sample only balances the load);
  * inside a bucket the voxel index linearised over the bucket's OWN bounding box orders the voxels as PCL's index over the segment's box
    does, and the sort word `index << pos_bits | input position` puts the points of a voxel in input order (the summation order);
  * the order in which points ARRIVE in a bucket's slot array (atomics: arbitrary on the device) does not show in the result;
  * every give-up condition is detected: a coordinate beyond +-2^20 voxels, more than VB_CAP points in a bucket, a bucket box that needs
    more than 63 sort bits, PCL's pass-through case (segment box beyond INT_MAX voxels).

The device itself is compared with the oracle in tests/test_gpu_voxbucket.py; this file pins the ALGORITHM."""
import numpy as np
import pytest

VB_CAP, VB_T, VB_SAMPLE, VB_OFF = 4096, 2048, 512, 1 << 20


class GiveUp(Exception):
    def __init__(self, reason):
        super().__init__(f"give-up reason {reason}")
        self.reason = reason


def _bits(v):
    return int(v).bit_length()


def _voxels(pts, leaf):
    """floor(v * inverse leaf) in float arithmetic, as pcl::VoxelGrid and vb_voxel() form it; reason 0 beyond +-2^20 or not finite"""
    inv = np.float32(1.0) / np.float32(leaf)
    f = np.floor(pts[:, :3] * inv)
    if not np.all(np.abs(f) < np.float32(VB_OFF)):   # (also catches NaN / inf)
        raise GiveUp(0)
    return f.astype(np.int64)


def _key(v):
    """vb_key(): (iz, iy, ix) lexicographically, 21 bits each"""
    return [((int(z) + VB_OFF) << 42) | ((int(y) + VB_OFF) << 21) | (int(x) + VB_OFF) for x, y, z in v]


def plan_splitters(pts, leaf):
    """k_vb_plan: ceil(n / VB_T) buckets; VB_SAMPLE evenly spaced points ranked by voxel key, every (m / buckets)-th one a splitter"""
    n = len(pts)
    nb = max(1, -(-n // VB_T))
    if nb > VB_SAMPLE:
        raise GiveUp(1)
    if nb == 1:
        return [0]
    m = min(n, VB_SAMPLE)
    sample = pts[[(t * n) // m for t in range(m)]]
    keys = sorted(_key(_voxels(sample, leaf)))
    return [0] + [keys[(k * m) // nb] for k in range(1, nb)]


def bucketed_voxel_grid(pts, leaf, splitters=None, rng=None):
    """one segment through k_vb_plan / k_vb_stack / k_vb_reduce; rng: shuffles the arrival order inside every bucket"""
    pts = np.ascontiguousarray(pts, np.float32)
    n = len(pts)
    if n == 0:
        return np.zeros((0, 4), np.float32)
    lo = plan_splitters(pts, leaf) if splitters is None else list(splitters)
    assert lo[0] == 0 and all(a <= b for a, b in zip(lo, lo[1:]))
    v = _voxels(pts, leaf)                                   # k_vb_stack: exact voxel of every point (reason 0)
    keys = _key(v)
    lo_arr = np.array(lo, dtype=object)
    # bucket = the last splitter <= key (the binary search of k_vb_stack: "if (s[mid] <= key) lo = mid; else hi = mid")
    bucket = np.array([int(np.searchsorted(lo_arr, k, side="right")) - 1 for k in keys])
    pos_bits = max(1, _bits(n - 1))
    # PCL's own pass-through test on the segment's box (the last bucket does it on the device, reason 2)
    dims = v.max(0) - v.min(0) + 1
    if int(dims[0]) * int(dims[1]) * int(dims[2]) > 2147483647:
        raise GiveUp(2)
    out = []
    for b in range(len(lo)):
        el = np.flatnonzero(bucket == b)                     # input positions, in arrival order (any)
        if len(el) > VB_CAP:
            raise GiveUp(5)
        if rng is not None:
            el = rng.permutation(el)
        if not len(el):
            continue
        vb = v[el]
        b0 = vb.min(0)
        dx, dy, dz = (int(d) for d in (vb.max(0) - b0 + 1))
        key_bits = _bits(dx * dy * dz - 1)
        if max(key_bits, 1) + pos_bits > 63:
            raise GiveUp(3)
        lin = [(int(x) - int(b0[0])) + dx * ((int(y) - int(b0[1])) + dy * (int(z) - int(b0[2]))) for x, y, z in vb]
        words = sorted((l << pos_bits) | int(p) for l, p in zip(lin, el))   # the LSD radix sort's result: ascending words, all distinct
        s = 0
        while s < len(words):                                # run heads -> one mean per voxel, summed in sorted = input order
            e, acc = s, np.zeros(4, np.float32)
            while e < len(words) and words[e] >> pos_bits == words[s] >> pos_bits:
                acc = (acc + pts[words[e] & ((1 << pos_bits) - 1)]).astype(np.float32)
                e += 1
            out.append(acc / np.float32(e - s))
            s = e
    return np.array(out, np.float32).reshape(-1, 4)


def _sweep_like(rng, n, extent):
    """surfaces, not a volume: planes and a wall, as a registration's stack cloud is"""
    kind = rng.integers(0, 3, n)
    p = rng.uniform(-extent, extent, (n, 3))
    p[kind == 0, 1] = -1.8 + 0.01 * rng.normal(size=(kind == 0).sum())          # ground
    p[kind == 1, 0] = extent * 0.7 + 0.005 * rng.normal(size=(kind == 1).sum())  # an axis-aligned wall (dense in one voxel column)
    out = np.zeros((n, 4), np.float32)
    out[:, :3] = p
    out[:, 3] = rng.integers(0, 64, n) + rng.uniform(0, 0.1, n)
    return out


@pytest.mark.parametrize("n,extent,leaf", [(7000, 20.0, 0.2), (30000, 40.0, 0.4), (30000, 6.0, 0.4), (2048, 10.0, 0.2), (2049, 10.0, 0.2), (300, 3.0, 0.4), (1, 1.0, 0.2)])
def test_sampled_splitters_give_pcls_result(orc, n, extent, leaf):
    rng = np.random.default_rng(n)
    pts = _sweep_like(rng, n, extent)
    want = orc.voxel_grid(pts, leaf)
    got = bucketed_voxel_grid(pts, leaf)
    assert got.shape == want.shape and np.array_equal(got, want)
    # arrival order inside the buckets is the atomics' business: it must not show
    assert np.array_equal(bucketed_voxel_grid(pts, leaf, rng=rng), want)


def test_any_splitters_give_the_same_result(orc):
    """the splitters only balance the load: arbitrary ones (duplicates, none at all, all at one end) change nothing but the bucket sizes"""
    rng = np.random.default_rng(3)
    pts = _sweep_like(rng, 6000, 12.0)     # (small enough that no choice overflows a bucket)
    want = orc.voxel_grid(pts, 0.4)
    keys = sorted(_key(_voxels(pts, 0.4)))
    for trial in range(8):
        k = int(rng.integers(0, 7))
        sp = sorted(keys[i] for i in rng.integers(0, len(keys), k))
        if trial == 0:
            sp = [keys[-1]] * 3            # everything but the last voxel in bucket 0
        if trial == 1:
            sp = [keys[0], keys[0]]        # empty leading buckets
        if trial == 2:
            sp = [keys[len(keys) // 2] + 1]   # a splitter that is no point's key
        try:
            got = bucketed_voxel_grid(pts, 0.4, splitters=[0] + sp, rng=rng)
        except GiveUp as g:
            assert g.reason == 5           # an unbalanced choice may overflow a bucket — it must say so, never answer wrongly
            continue
        assert np.array_equal(got, want), (trial, sp)


def test_duplicates_and_voxel_boundaries(orc):
    """points exactly on voxel faces, repeated points, negative coordinates: floor() and the input-order means"""
    rng = np.random.default_rng(5)
    base = rng.integers(-40, 40, (3000, 3)).astype(np.float32) * np.float32(0.2)     # lattice points = voxel faces for leaf 0.2 / 0.4
    pts = np.zeros((9000, 4), np.float32)
    pts[:, :3] = np.concatenate([base, base, base + np.float32(1e-4) * rng.normal(size=base.shape).astype(np.float32)])
    pts[:, 3] = rng.integers(0, 16, len(pts))
    pts = pts[rng.permutation(len(pts))]
    for leaf in (0.2, 0.4):
        assert np.array_equal(bucketed_voxel_grid(pts, leaf, rng=rng), orc.voxel_grid(pts, leaf))


def test_give_up_conditions():
    rng = np.random.default_rng(7)
    pts = _sweep_like(rng, 5000, 10.0)
    far = pts.copy()
    far[17, 0] = 0.2 * (1 << 20) + 1.0
    with pytest.raises(GiveUp) as g:
        bucketed_voxel_grid(far, 0.2)
    assert g.value.reason == 0                                  # a coordinate beyond +-2^20 voxels
    nan = pts.copy()
    nan[3, 2] = np.nan
    with pytest.raises(GiveUp) as g:
        bucketed_voxel_grid(nan, 0.2)
    assert g.value.reason == 0
    dense = pts.copy()
    dense[:VB_CAP + 1, :3] = np.float32(0.05)                   # more points in ONE voxel than a bucket holds
    with pytest.raises(GiveUp) as g:
        bucketed_voxel_grid(dense, 0.4)
    assert g.value.reason == 5
    wide = pts.copy()
    wide[0, :3] = -0.19 * (1 << 20)
    wide[1, :3] = 0.19 * (1 << 20)
    with pytest.raises(GiveUp) as g:
        bucketed_voxel_grid(wide, 0.2)
    assert g.value.reason == 2                                  # PCL's pass-through case: the general kernel handles it


def test_segments_with_more_buckets_than_the_lds_copy_holds(orc):
    """> 64 buckets per segment (the binary search leaves the workgroup's LDS copy of the splitters): 150 k points"""
    rng = np.random.default_rng(11)
    pts = _sweep_like(rng, 150000, 60.0)
    assert len(plan_splitters(pts, 0.4)) == 74
    assert np.array_equal(bucketed_voxel_grid(pts, 0.4), orc.voxel_grid(pts, 0.4))
