root=$(pwd); out=$root/gpurun_out/r05_lmts; mkdir -p $out
LOAMX_LIB=build/prof/libloamx_lm.so LOAMX_BENCH_TIMING_PERIOD=1000 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-pcie --repeat 1 --long-steps 0 > $out/bench.json 2> $out/err.txt
grep "lm ts" $out/err.txt | tail -40
