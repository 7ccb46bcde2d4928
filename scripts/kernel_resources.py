#!/usr/bin/env python3
"""Static resource table of every device kernel in loam_velodyne_amd/csrc (no GPU needed): each .hip file is compiled
device-only for gfx950 with the product's flags and the amdhsa metadata note is read back — VGPRs (arch + accumulation),
SGPRs, LDS bytes (static), scratch bytes (spills), workgroup size bound, and the wave occupancy per SIMD those imply on
CDNA4 (512 VGPRs per SIMD lane-slice, 8 waves max).  Writes a markdown table to stdout.
Usage: python scripts/kernel_resources.py > profiles/rNN_kernel_resources.md"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "loam_velodyne_amd", "csrc")
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I/opt/rocm/include --cuda-device-only --no-gpu-bundle-output".split()
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"


def demangle(n):
    out = subprocess.run([FILT, n], capture_output=True, text=True).stdout.strip()
    out = out.replace("(anonymous namespace)::", "")
    out = re.sub(r"\(.*$", "", out)            # drop the parameter list
    return out.replace("loamx::", "")


def kernels_of(path, tmp):
    co = os.path.join(tmp, os.path.basename(path) + ".co")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c", path, "-o", co], check=True, capture_output=True)
    notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True).stdout
    rows = []
    for blk in re.split(r"\n\s+- \.agpr_count:", "\n" + notes)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k, d="0": (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, d])[1]
        name = (re.search(r"\n\s+\.name:\s+(\S+)", blk) or [None, "?"])[1]     # (args carry .name entries too: take the kernel-level one)
        rows.append(dict(name=demangle(name), vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")),
                         lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size")),
                         wg=int(g("max_flat_workgroup_size")), spill_v=int(g("vgpr_spill_count")), spill_s=int(g("sgpr_spill_count"))))
    return rows


def occupancy(vgpr, agpr):
    total = max(vgpr + agpr, 1)
    total = (total + 7) // 8 * 8            # allocation granule
    return max(1, min(8, 512 // total))


def main():
    with tempfile.TemporaryDirectory() as tmp:
        print("| file | kernel | VGPR | AGPR | SGPR | LDS B (static) | scratch B | VGPR spills | max WG | waves / SIMD |")
        print("|---|---|---|---|---|---|---|---|---|---|")
        for path in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            try:
                rows = kernels_of(path, tmp)
            except subprocess.CalledProcessError as e:
                print(f"| {os.path.basename(path)} | (device compile failed: {e.stderr.decode()[:80]}) |", file=sys.stderr)
                continue
            lib = [r for r in rows if "rocprim" in r["name"] and r["vgpr"] > 0]
            rows = [r for r in rows if "rocprim" not in r["name"]]
            if lib:   # rocPRIM's merge sort behind the voxel grid (one line: the widest instantiation actually emitted)
                w = max(lib, key=lambda r: r["vgpr"])
                w = dict(w, name=f"rocprim merge sort <u64 key, u32 value> ({len(lib)} kernels, widest shown)")
                rows.append(w)
            for r in sorted(rows, key=lambda r: r["name"]):
                print(f"| {os.path.basename(path)} | `{r['name']}` | {r['vgpr']} | {r['agpr']} | {r['sgpr']} | {r['lds']} | {r['scratch']} | "
                      f"{r['spill_v']} | {r['wg']} | {occupancy(r['vgpr'], r['agpr'])} |")


if __name__ == "__main__":
    main()
