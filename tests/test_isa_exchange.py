"""Static check of the exchange loads (no GPU): `xchg_load_nowait` (dev_math.hpp) issues an agent-scope load WITHOUT its wait so that a
batch of them is in flight at once; nothing may touch a destination register before the batch's `s_waitcnt vmcnt(0)` — the hardware
does not interlock on a pending load, and the compiler does not know the register is pending.  The call site's
discipline is checked here in the ISA of the product build's flags, so a compiler that starts moving or spilling those registers
fails the CPU suite instead of corrupting a normal-equation sum on the device."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "loam_velodyne_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I/opt/rocm/include --cuda-device-only --no-gpu-bundle-output".split()


def _regs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return out


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(OBJDUMP)), reason="needs hipcc and llvm-objdump")
def test_no_instruction_touches_a_pending_exchange_load(tmp_path):
    co = str(tmp_path / "registration.co")
    subprocess.run([HIPCC, *FLAGS, "-c", os.path.join(CSRC, "registration.hip"), "-o", co], check=True, capture_output=True)
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    batches, worst = 0, 0
    pending, in_batch, func = set(), 0, "?"
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            func, pending, in_batch = m.group(1), set(), 0
            continue
        ins = line.strip().split("//")[0].strip()
        if not ins:
            continue
        op = ins.split()[0]
        if op == "global_load_dwordx2" and ins.rstrip().endswith("sc1"):      # the no-wait agent-scope load (64-bit partial sums)
            dst, rest = ins[len(op):].split(",", 1)
            assert not (_regs(rest) & pending), f"{func}: address of an exchange load depends on a pending one: {ins}"
            pending |= _regs(dst)
            in_batch += 1
            continue
        if op == "s_waitcnt" and "vmcnt(0)" in ins:
            if in_batch:
                batches += 1
                worst = max(worst, in_batch)
            pending, in_batch = set(), 0
            continue
        if pending:
            assert not (_regs(ins) & pending), f"{func}: '{ins}' touches a register of an exchange load that is still in flight"
            assert not op.startswith("s_cbranch") and op not in ("s_branch", "s_endpgm", "s_setpc_b64"), f"{func}: control flow between exchange loads and their wait: {ins}"
    assert batches >= 1 and worst >= 8, (batches, worst)   # the batch exists and is a batch (k_gn_iter's solve_sweep: 18 loads in flight)


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(OBJDUMP)), reason="needs hipcc and llvm-objdump")
def test_inline_16_byte_exchange_store_is_followed_by_its_wait_states(tmp_path):
    """xrec_store (dev_math.hpp) is a 16-byte agent-scope store issued through inline asm: the instruction behind it must be the
    `s_nop` of the same statement (a store of more than 64 bits followed at once by a write to its data registers is a hazard the
    compiler only pads for stores it emitted itself)"""
    co = str(tmp_path / "odometry.co")
    subprocess.run([HIPCC, *FLAGS, "-c", os.path.join(CSRC, "odometry.hip"), "-o", co], check=True, capture_output=True)
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    lines = [ln.strip().split("//")[0].strip() for ln in dis.splitlines()]
    lines = [ln for ln in lines if ln and not re.match(r"^[0-9a-f]+ <", ln)]
    stores = [i for i, ln in enumerate(lines) if ln.startswith("global_store_dwordx4") and ln.rstrip().endswith("sc1")]
    assert stores, "no agent-scope 16-byte store found in odometry.hip's kernels"
    for i in stores:
        assert lines[i + 1].startswith("s_nop"), (lines[i], lines[i + 1])
