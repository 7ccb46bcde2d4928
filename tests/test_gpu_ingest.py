"""GPU: raw-sweep ingestion (MultiScanRegistration::process) vs the oracle.
Ring assignment, order inside a ring and xyz are exact (integer / copy work); intensity = ring + relTime is float work whose
atan / atan2 go through a different libm on the device: tolerance 2 ulp of the stored value (ring + relTime <= 64.1)."""
import numpy as np
import pytest

import oracle_py as op
from loam_velodyne_amd import loamx, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sensor,az,bad", [("VLP-16", 1800, 0), ("VLP-16", 900, 7), ("HDL-32", 512, 16), ("HDL-64E", 2048, 64)])
def test_binning_matches_oracle(orc, small_world, sensor, az, bad):
    sw = synth.make_sweep(small_world, sensor, np.zeros(6), np.array([0.002, 0.01, 0.0, 0.3, 0.0, 0.9]), seed=az, az_steps=az)
    raw = synth.to_raw(sw, bad_every=bad)
    if bad:   # a revolution that does not start at the azimuth seam
        R = synth.SENSORS[sensor][0]
        raw = np.roll(raw.reshape(az, R, 3), az // 3, axis=0).reshape(-1, 3)
    o_pts, o_rs = op.multiscan_bin(orc, raw, sensor)
    g = loamx.ScanRegistration().process_raw(raw, sensor)
    assert np.array_equal(g["ring_sizes"], o_rs)
    assert g["full"].shape == o_pts.shape
    assert np.array_equal(g["full"][:, :3], o_pts[:, :3])
    assert np.array_equal(np.floor(g["full"][:, 3] + 1e-4), np.floor(o_pts[:, 3] + 1e-4))
    assert np.all(np.abs(g["full"][:, 3] - o_pts[:, 3]) <= 2 * np.spacing(np.maximum(np.abs(o_pts[:, 3]), np.float32(1.0))))


def test_features_from_raw_equal_features_from_rings(orc, small_world):
    """raw -> (GPU binning) -> features  ==  oracle binning -> GPU features on the binned rings, cloud for cloud."""
    sw = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.zeros(6), seed=11, az_steps=1200)
    raw = synth.to_raw(sw, bad_every=50)
    g = loamx.ScanRegistration().process_raw(raw, "VLP-16")
    ref = loamx.ScanRegistration().process(g["full"], g["ring_sizes"])
    for k in ("sharp", "less_sharp", "flat", "less_flat"):
        assert np.array_equal(g[k], ref[k]), k
    o_pts, o_rs = op.multiscan_bin(orc, raw, "VLP-16")
    of = op.ScanRegistration(orc).process(o_pts, o_rs)
    for k in ("sharp", "less_sharp", "flat", "less_flat"):
        assert g[k].shape == of[k].shape, k
        assert np.array_equal(g[k][:, :3], of[k][:, :3]), k          # same picks (relTime differences of 1e-7 do not move them)


def test_custom_mapper_and_errors():
    rng = np.random.default_rng(1)
    raw = rng.normal(0, 10, (5000, 3)).astype(np.float32)
    g = loamx.ScanRegistration().process_raw(raw, mapper=(-20.0, 20.0, 8))
    assert len(g["ring_sizes"]) == 8 and g["ring_sizes"].sum() == len(g["full"]) <= 5000
    with pytest.raises(loamx.LoamxError):
        loamx.ScanRegistration().process_raw(raw, mapper=(10.0, -10.0, 8))     # upper <= lower (MultiScanRegistration.cpp:114-117)
    with pytest.raises(loamx.LoamxError):
        loamx.ScanRegistration().process_raw(raw, mapper=(-10.0, 10.0, 1))      # n < 2 (:118-121)
    with pytest.raises(loamx.LoamxError):
        loamx.ScanRegistration().process_raw(raw, sensor="VLP-128")
    e = loamx.ScanRegistration().process_raw(np.zeros((0, 3), np.float32), "VLP-16")
    assert len(e["full"]) == 0 and e["ring_sizes"].sum() == 0


def test_imu_deskew_matches_oracle(orc, small_world):
    """updateIMUData + projectPointToStartOfSweep + updateIMUTransform, sweep after sweep, with IMU messages
    arriving between the sweeps (the history is a 200-deep ring buffer, so indices shift once it is full)."""
    rng = np.random.default_rng(21)
    osr, gsr = op.ScanRegistration(orc), loamx.ScanRegistration()
    t_imu, k_imu = 0.0, 0
    for k in range(4):
        t_scan = 0.1 * (k + 1)
        while t_imu < t_scan + 0.12:                      # ~130 messages per sweep: the buffer wraps during the second sweep
            roll, pitch, yaw = 0.02 * np.sin(3 * t_imu), 0.015 * np.cos(2 * t_imu), 0.4 * t_imu + (6.2 if k_imu % 97 == 50 else 0.0)
            acc = (0.8 * np.sin(5 * t_imu), 0.1, -0.5 * np.cos(4 * t_imu))
            osr.update_imu(t_imu, roll, pitch, yaw, acc)
            gsr.update_imu(t_imu, roll, pitch, yaw, acc)
            t_imu += 0.00077
            k_imu += 1
        sw = synth.make_sweep(small_world, "VLP-16", np.zeros(6), np.array([0.0, 0.04, 0.0, 0.1, 0.0, 0.5]), seed=30 + k, az_steps=700)
        raw = synth.to_raw(sw, bad_every=33)
        o = osr.process_raw(raw, t_scan, "VLP-16")
        gsr.set_time(t_scan)
        g = gsr.process_raw(raw, "VLP-16")
        assert np.array_equal(g["ring_sizes"], o["ring_sizes"])
        # de-skewed coordinates (range <= 90 m): relTime differs by an ulp between the two libm's, the interpolation ratio
        # (IMU samples 0.77 ms apart) amplifies that by 1e2, and the planted 6.2 rad yaw wrap makes one interval 0.08 rad wide
        assert np.abs(g["full"][:, :3] - o["full"][:, :3]).max() < 1e-4, k
        assert np.all(np.abs(g["full"][:, 3] - o["full"][:, 3]) <= 2 * np.spacing(np.maximum(np.abs(o["full"][:, 3]), np.float32(1.0))))
        assert np.abs(gsr.imu_trans() - o["imu_trans"]).max() < 2e-5, (k, gsr.imu_trans(), o["imu_trans"])
        if k >= 1:
            assert np.abs(o["imu_trans"]).max() > 1e-3                              # the IMU path is really active
        for name in ("sharp", "less_sharp", "flat", "less_flat"):
            assert abs(len(g[name]) - len(o[name])) <= max(2, len(o[name]) // 100), (k, name)
