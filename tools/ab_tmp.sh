run() { timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>&1 | grep -o '"value[^,]*\|kernel_ms[^,]*\|"ms_per_step[^,]*' | tr '\n' ' '; echo " :: $* SPLITMAX=$APGP_SWEEP_SPLIT_MAX S2=$APGP_SWEEP2"; }
for M in 4096 16384 100000; do for s2 in 0 1; do APGP_SWEEP2=$s2 run --candidates $M; done; done
for M in 1000000; do for N in 512 1152 8192; do for s2 in 0 1; do APGP_SWEEP2=$s2 run --candidates $M --n-train $N; done; done; done
for s2 in 0 1; do APGP_SWEEP2=$s2 run --ndim 16; APGP_SWEEP2=$s2 run --ndim 2; done
# split threshold: rest blocks = 64,128,200 on top of 2 full rounds
for rest in 64 128 200 240; do M=$(( (512 + rest) * 64 )); for sm in 0 216; do APGP_SWEEP_SPLIT_MAX=$sm run --candidates $M; done; done
