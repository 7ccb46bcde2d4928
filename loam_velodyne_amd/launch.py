"""torchrun-free launcher for the multi-GPU batched mode: one process per GPU of ONE node, RCCL rendezvous through a file.

    python -m loam_velodyne_amd.launch --nproc 8 [--id-file PATH] script.py [args...]

Every child gets LOAMX_RANK, LOAMX_WORLD_SIZE, LOAMX_LOCAL_RANK (= its GPU) and LOAMX_ID_FILE; `rendezvous()` below turns that
into a loamx.Dist: rank 0 writes the 128-byte ncclUniqueId to the file (atomically), the others wait for it.  No torch, no
network rendezvous: the ranks of one node share a file system.  HSA_ENABLE_IPC_MODE_LEGACY=0 is exported for the children (the
host driver only supports dmabuf IPC)."""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import tempfile
import time


def rendezvous(timeout_s: float = 120.0):
    """Called inside a launched process: returns (loamx.Dist, rank, world_size, local_rank)."""
    from . import loamx
    rank = int(os.environ.get("LOAMX_RANK", "0"))
    world = int(os.environ.get("LOAMX_WORLD_SIZE", "1"))
    local = int(os.environ.get("LOAMX_LOCAL_RANK", str(rank)))
    path = os.environ.get("LOAMX_ID_FILE")
    if world == 1 and not path:
        return loamx.Dist(loamx.Dist.unique_id(), 0, 1, local), 0, 1, local
    assert path, "LOAMX_ID_FILE is not set (start the ranks with python -m loam_velodyne_amd.launch)"
    if rank == 0:
        uid = loamx.Dist.unique_id()
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)          # atomic: a reader never sees a partial id
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"rank {rank}: no unique id at {path} after {timeout_s} s")
            time.sleep(0.01)
        with open(path, "rb") as f:
            uid = f.read()
    return loamx.Dist(uid, rank, world, local), rank, world, local


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--nproc", type=int, required=True, help="ranks = GPUs of this node to use")
    ap.add_argument("--id-file", default=None, help="rendezvous file (default: a fresh temporary file)")
    ap.add_argument("script")
    ap.add_argument("args", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    tmpdir = None
    id_file = a.id_file
    if id_file is None:
        tmpdir = tempfile.mkdtemp(prefix="loamx_rdv_")
        id_file = os.path.join(tmpdir, "nccl_unique_id")
    elif os.path.exists(id_file):
        os.remove(id_file)
    procs = []
    for r in range(a.nproc):
        env = dict(os.environ, LOAMX_RANK=str(r), LOAMX_WORLD_SIZE=str(a.nproc), LOAMX_LOCAL_RANK=str(r), LOAMX_ID_FILE=id_file)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, a.script, *a.args], env=env))
    # any rank that ends badly (non-zero exit, or killed by a signal: negative return code) fails the job and takes the others
    # down with it — a rank lost before or inside a collective would leave its peers, and this launcher, waiting forever
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 128 - r   # (shell convention for signals)
                for q in live:
                    q.terminate()
        if live:
            time.sleep(0.02)
    if tmpdir:
        try:
            if os.path.exists(id_file):
                os.remove(id_file)
            os.rmdir(tmpdir)
        except OSError:
            pass
    return rc


if __name__ == "__main__":
    sys.exit(main())
