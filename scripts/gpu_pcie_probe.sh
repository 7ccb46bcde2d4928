#!/bin/bash
# Which engine carries the PCIe window's copies, and what do variants of the download do?  usage: scripts/gpu_pcie_probe.sh <tag>
set -u
tag=$1; out=gpurun_out/$tag; mkdir -p $out
AMD_LOG_LEVEL=4 AMD_LOG_MASK=0x300 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-configs --repeat 1 > $out/logged.json 2> $out/logged.err
grep -c "" $out/logged.err; grep -i "falling\|failed" $out/logged.err | sort | uniq -c | head; grep "HSA Copy" $out/logged.err | grep -E "size=1677" | sed 's/.*HSA Copy/HSA Copy/' | cut -c1-200 | sort | uniq -c | sort -rn | head -20
grep -i "blit\|shader" $out/logged.err | sed 's/^[^]]*\]//' | cut -c1-160 | sort | uniq -c | sort -rn | head -10
tail -c 200000 $out/logged.err > $out/logged_tail.err; rm $out/logged.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --repeat 1 --ab-pcie ";LOAMX_D2H_NOWAIT=1;LOAMX_D2H_CHUNK_KB=2048;LOAMX_D2H_CHUNK_KB=256;LOAMX_D2H_ON_CSTREAM=1;" > $out/ab.json 2> $out/ab.err
grep "ab-pcie" $out/ab.err; tail -c 900 $out/ab.json | head -c 700
