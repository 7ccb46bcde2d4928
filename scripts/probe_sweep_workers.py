"""how many sweep-generating worker processes pay on this host? (bench.py LOAMX_BENCH_WORKERS)"""
import multiprocessing as mp, os, sys, time
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loam_velodyne_amd import synth
import bench
def main():
  poses = synth.trajectory(8, yaw_step_deg=1.43)
  print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
  for bound in (False, True):
      if bound:
          os.sched_setaffinity(0, range(0, (os.cpu_count() or 2) // 2))   # (roughly what the NUMA binding of bench.py does)
      for nw in (32, 64, 96):
          jobs = [(125.0, "HDL-64E", poses[k % 8], poses[k % 8 + 1], k) for k in range(768)]
          t0 = time.time()
          with ProcessPoolExecutor(max_workers=nw, mp_context=mp.get_context("spawn"), initializer=bench.unbind_worker) as ex:
              list(ex.map(synth.make_sweep_job, jobs, chunksize=max(1, len(jobs) // (4 * nw))))
          print("parent bound" if bound else "parent unbound", "workers", nw, "768 sweeps in %.1f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
  main()
