"""CPU: the synthetic generator is deterministic and produces sweeps of the shapes BASELINE.md names."""
import numpy as np

from loam_velodyne_amd import synth


def test_sweep_shapes_and_determinism(small_world):
    for sensor, (R, A, lo, hi) in synth.SENSORS.items():
        sw = synth.make_sweep(small_world, sensor, np.zeros(6), np.zeros(6), seed=1, az_steps=256)
        assert sw.points.shape == (R * 256, 4) and sw.ring_sizes.tolist() == [256] * R
        ring = np.floor(sw.points[:, 3]).astype(int)
        assert np.array_equal(ring, np.repeat(np.arange(R), 256))
        rel = sw.points[:, 3] - ring
        assert rel.min() >= 0 and rel.max() < 0.1
        r = np.linalg.norm(sw.points[:, :3], axis=1)
        assert r.min() > 0.5 and r.max() < 400
    a = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=2, az_steps=128)
    b = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=2, az_steps=128)
    assert np.array_equal(a.points, b.points)
    assert synth.SENSORS["VLP-16"][0] * synth.SENSORS["VLP-16"][1] == 28800
    assert synth.SENSORS["HDL-64E"][0] * synth.SENSORS["HDL-64E"][1] == 131072


def test_map_is_exact_size(small_world):
    c, s = small_world.make_map(20000)
    assert len(c) == 2000 and len(s) == 18000 and c.dtype == np.float32
    assert np.abs(c[:, [0, 2]]).max() <= 45.5
