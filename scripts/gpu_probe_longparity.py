"""GPU probe (round 6): where does the long window's mapped pose leave the oracle chain's, and what is the reference's own envelope there?

The bench line's `value_long` block compares stream 0 of the batched pipeline with the oracle chain over ~405 HDL-64E sweeps against the
1 M-point frozen map.  This probe runs that stream alone and, beside the GPU chain:
  A  the oracle's parity build (liboracle.so, -O2 -ffp-contract=off)            — the chain of record
  B  the oracle's timed build (liboracle_fast.so, -O3 -march=native: FMA contraction) — the same source under other legal flags
  C  the reference's own translation units (oracle/_ref/libref_*.so)
  D  the same with libref_mapping_alt.so (the forwarded Eigen operations done the other plausible way)
  C2 / D2  the same with libref_odometry_alt.so and the plain / the alternative mapping
  E  the oracle's registration fed with the GPU's odometry (its re-projected clouds and transformSum): isolates the registration
  F  per step from identical state: the GPU's batched registration (loamx_batch_*) on chain A's inputs of every sweep
and writes every chain's per-sweep poses and counts to gpurun_out/longparity.npz.

usage: gpu_probe_longparity.py [N=400] [W=5]
"""
import os
import sys
import time
import json

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

SENSOR, M = "HDL-64E", 1_000_000
SHM = "/dev/shm/lp_sweeps.npy"


def cpu_chain(kind):
    """one CPU chain over the sweeps in SHM -> rows (t, ts[6], pose[6], odom it, odom sel, map it, map sel, corner_ds, surf_ds)"""
    import oracle_py as op
    from loam_velodyne_amd import synth, dist as lxdist
    sw = np.load(SHM, mmap_mode="r")
    rs = np.full(synth.SENSORS[SENSOR][0], synth.SENSORS[SENSOR][1], np.uint32)
    m = np.load("/dev/shm/lp_map.npy")
    n_corner = int(round(len(m) * 0.1))
    g = lxdist.stream_start(0)
    start = np.array([0, 0, 0, g[0], g[1], g[2]], np.float32)
    if kind in ("A", "B"):
        orc = op.Oracle(fast=(kind == "B"))
        sr, od, mp = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc)
    else:   # C plain | D mapping alt | C2 odometry alt | D2 both alt
        sr = op.RefScanRegistration()
        od = op.RefLaserOdometryAlt() if kind in ("C2", "D2") else op.RefLaserOdometry()
        mp = op.RefLaserMappingAlt() if kind in ("D", "D2") else op.RefLaserMapping()
    mp.set_frozen(m[:n_corner], m[n_corner:])
    mp.set_transform("aft", start)
    rows, inputs = [], []
    t0 = time.time()
    for t in range(len(sw)):
        od.set_features(sr.process(np.asarray(sw[t]), rs))
        od.process()
        if t > 0:
            mp.set_transform("sum", od.transform_sum)
            lc, ls, guess = od.last_corner(), od.last_surf(), mp.associate()
            pose = mp.register_frozen(lc, ls, guess)
            mp.set_transform("bef", od.transform_sum)
            mp.set_transform("aft", pose)
            if kind in ("A", "B"):
                so, sm_ = od.stats(), mp.stats()
                cnt = [so["iterations"], so["sel"], sm_["iterations"], sm_["sel"], sm_["corner_ds"], sm_["surf_ds"]]
            else:
                cnt = [0] * 6
            rows.append(np.concatenate([[t], np.array(od.transform_sum, np.float64), np.array(pose, np.float64), cnt]))
            if kind == "A":
                inputs.append((lc.copy(), ls.copy(), np.array(guess, np.float32)))
    np.save(f"/dev/shm/lp_chain_{kind}.npy", np.array(rows))
    if kind == "A":
        np.savez("/dev/shm/lp_inputs_A.npz", **{f"c{t}": x[0] for t, x in enumerate(inputs)}, **{f"s{t}": x[1] for t, x in enumerate(inputs)},
                 g=np.array([x[2] for x in inputs]))
    return kind, time.time() - t0


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    N = int(args[0]) if len(args) > 0 else 400
    W = int(args[1]) if len(args) > 1 else 5
    LOOK = 6
    from loam_velodyne_amd import loamx, synth, dist as lxdist
    import multiprocessing as mp_
    from concurrent.futures import ProcessPoolExecutor
    T = 1 + W + N
    T_all = T + LOOK
    world = synth.World(half_extent=125.0)
    cm, sm = world.make_map(M)
    m = np.concatenate([cm, sm], axis=0)
    np.save("/dev/shm/lp_map.npy", m)
    g = lxdist.stream_start(0)
    start = np.array([0, 0, 0, g[0], g[1], g[2]], np.float32)
    poses = synth.trajectory(T_all, yaw_step_deg=1.43, start=g)
    jobs = [(125.0, SENSOR, poses[t], poses[t + 1], 1000 * 0 + t) for t in range(T_all)]
    nw = max(1, min(64, len(os.sched_getaffinity(0)), len(jobs)))
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=nw, mp_context=mp_.get_context("spawn")) as ex:
        made = list(ex.map(synth.make_sweep_job, jobs, chunksize=max(1, len(jobs) // (4 * nw))))
    print(f"sweeps: {len(made)} in {time.time() - t0:.1f} s on {nw} workers", flush=True)
    rs = made[0][1]
    assert all(np.array_equal(r, rs) for _, r in made)
    np.save(SHM, np.stack([p for p, _ in made[:T]]))

    # ---- CPU chains in worker processes, beside the GPU chain
    ex = ProcessPoolExecutor(max_workers=6, mp_context=mp_.get_context("spawn"))
    kinds = ["A", "B"]
    import oracle_py as op
    if op.RefScanRegistration.available() and op.RefLaserMapping.available():
        kinds += ["C", "D"]
        if op.RefLaserOdometryAlt.available():
            kinds += ["C2", "D2"]
    futs = [ex.submit(cpu_chain, k) for k in kinds]

    if "--cpu-only" in sys.argv:   # (no device: the CPU chains against the G / E / F rows of an earlier run's gpurun_out/longparity.npz)
        for f in futs:
            k, dt = f.result()
            print(f"chain {k}: {dt:.1f} s", flush=True)
        ex.shutdown()
        ch = {k: np.load(f"/dev/shm/lp_chain_{k}.npy") for k in kinds}
        old = np.load(os.path.join(ROOT, "gpurun_out", "longparity.npz"))
        np.savez(os.path.join(ROOT, "gpurun_out", "longparity_cpu.npz"), G=old["G"], E=old["E"], F=old["F"], **{f"ch{k}": v for k, v in ch.items()})
        for k in kinds:
            for k2 in ("A", "C"):
                if k != k2 and k2 in ch:
                    dd = np.abs(ch[k][:, 10:13] - ch[k2][:, 10:13]).max(1)
                    print(f"{k} vs {k2}: max {dd.max():.3e} at sweep {int(ch[k][int(np.argmax(dd)), 0])}, rmse {np.sqrt((dd ** 2).mean()):.3e}, n > 1e-4: {(dd > 1e-4).sum()}, p99 {np.percentile(dd, 99):.3e}")
        return
    # ---- the GPU chain: stream 0 alone, as in the bench (staged batch, look-ahead on)
    p = loamx.Pipeline(1)
    p.set_frozen(cm, sm)
    p.set_state(0, aft=start)
    p.upload([[made[t]] for t in range(T_all)])
    n_points = len(made[0][0])
    rows_g, gpu_in = [], []
    for t in range(T):
        p.step(t)
        _, ts, aft, st = p.get(0)
        if t > 0:
            rows_g.append(np.concatenate([[t], ts.astype(np.float64), aft.astype(np.float64),
                                          [st["odom_iterations"], st["odom_sel"], st["map_iterations"], st["map_sel"], st["corner_ds"], st["surf_ds"]]]))
            gpu_in.append((p.last_clouds(0, n_points), ts.copy()))
    p.close()
    G = np.array(rows_g)
    print("GPU chain done", flush=True)

    for f in futs:
        k, dt = f.result()
        print(f"chain {k}: {dt:.1f} s", flush=True)
    ex.shutdown()
    ch = {k: np.load(f"/dev/shm/lp_chain_{k}.npy") for k in kinds}
    A = ch["A"]

    # ---- E: the oracle's registration fed with the GPU's odometry
    orc = op.Oracle(fast=False)
    omp = op.LaserMapping(orc)
    omp.set_frozen(cm, sm)
    omp.set_transform("aft", start)
    rows_e = []
    for k, ((lc, ls), ts) in enumerate(gpu_in):
        omp.set_transform("sum", ts)
        pose = omp.register_frozen(lc, ls, omp.associate())
        omp.set_transform("bef", ts)
        omp.set_transform("aft", pose)
        s_ = omp.stats()
        rows_e.append(np.concatenate([[k + 1], ts.astype(np.float64), np.array(pose, np.float64), [0, 0, s_["iterations"], s_["sel"], s_["corner_ds"], s_["surf_ds"]]]))
    E = np.array(rows_e)

    # ---- F: per step from identical state — the GPU's batched registration on chain A's inputs
    inp = np.load("/dev/shm/lp_inputs_A.npz")
    guesses = inp["g"]
    nA = len(guesses)
    F = np.zeros((nA, 6)); Fst = []
    CH = 32
    b = loamx.Batch(CH)
    b.set_frozen(cm, sm)
    for a in range(0, nA, CH):
        e = min(nA, a + CH)
        b.upload([inp[f"c{t}"] for t in range(a, e)], [inp[f"s{t}"] for t in range(a, e)], guesses[a:e])
        b.run()
        gp, gs = b.download()
        F[a:e] = gp[:e - a]
        Fst += list(gs[:e - a])
    b.close()

    def d(X, Y):   # per-sweep max |difference| of the mapped pose (m, rad) and of the accumulated odometry
        n = min(len(X), len(Y))
        return (np.abs(X[:n, 10:13] - Y[:n, 10:13]).max(1), np.abs(X[:n, 7:10] - Y[:n, 7:10]).max(1),
                np.abs(X[:n, 4:7] - Y[:n, 4:7]).max(1), np.abs(X[:n, 1:4] - Y[:n, 1:4]).max(1))

    rep = {}
    for name, X, Y in [("gpu_vs_A", G, A), ("E_vs_gpu", E, G), ("E_vs_A", E, A)] + [(f"{k}_vs_A", ch[k], A) for k in kinds if k != "A"] + \
                      ([("D_vs_C", ch["D"], ch["C"])] if "D" in ch else []):
        pm, pr, om, orad = d(X, Y)
        k = int(np.argmax(pm))
        rep[name] = dict(mapped_max_m=float(pm.max()), mapped_max_rad=float(pr.max()), mapped_rmse_m=float(np.sqrt((pm ** 2).mean())), at_sweep=int(X[k, 0]),
                         n_above_1e4=int((pm > 1e-4).sum()), n_above_5e5=int((pm > 5e-5).sum()), odom_sum_max_m=float(om.max()), odom_sum_max_rad=float(orad.max()),
                         counts_equal=[int((X[:len(Y), 13 + j] == Y[:len(X), 13 + j]).sum()) for j in range(6)], n=int(min(len(X), len(Y))))
    dF = np.abs(F[:, 3:] - A[:nA, 10:13]).max(1)
    dFr = np.abs(F[:, :3] - A[:nA, 7:10]).max(1)
    kF = int(np.argmax(dF))
    rep["F_per_step_vs_A"] = dict(mapped_max_m=float(dF.max()), mapped_max_rad=float(dFr.max()), mapped_rmse_m=float(np.sqrt((dF ** 2).mean())), at_sweep=int(A[kF, 0]),
                                  n_above_1e5=int((dF > 1e-5).sum()), n_above_1e6=int((dF > 1e-6).sum()),
                                  stats_at_max={k_: (v_.tolist() if hasattr(v_, "tolist") else v_) for k_, v_ in dict(Fst[kF]).items()} if isinstance(Fst[kF], dict) else [int(x) for x in np.ravel(Fst[kF])], oracle_counts_at_max=[int(x) for x in A[kF, 13:]])
    # the sweeps where the GPU chain is furthest from A: what do the counts say there?
    pm = d(G, A)[0]
    worst = np.argsort(-pm)[:8]
    rep["gpu_vs_A_worst"] = [dict(sweep=int(G[k, 0]), d_m=float(pm[k]), comp=int(np.argmax(np.abs(G[k, 10:13] - A[k, 10:13]))), gpu_counts=[int(x) for x in G[k, 13:]], A_counts=[int(x) for x in A[k, 13:]],
                                  E_d_gpu=float(np.abs(E[k, 10:13] - G[k, 10:13]).max()), F_d_A=float(dF[k]) if k < nA else None,
                                  B_d_A=float(np.abs(ch["B"][k, 10:13] - A[k, 10:13]).max()),
                                  D_d_C=(float(np.abs(ch["D"][k, 10:13] - ch["C"][k, 10:13]).max()) if "D" in ch else None)) for k in worst]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "longparity.npz"), G=G, E=E, F=F, **{f"ch{k}": v for k, v in ch.items()})
    with open(os.path.join(ROOT, "gpurun_out", "longparity.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))
    for fn in (SHM, "/dev/shm/lp_map.npy", "/dev/shm/lp_inputs_A.npz") + tuple(f"/dev/shm/lp_chain_{k}.npy" for k in kinds):
        try:
            os.remove(fn)
        except OSError:
            pass


if __name__ == "__main__":
    main()
