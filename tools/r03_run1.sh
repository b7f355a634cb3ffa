set -e
mkdir -p gpurun_out/r03
for n in 65536 32768 16384 8192; do
  python bench.py --mode dynamic --randomize --gravity 9.81 --envs $n --no-cpu-baseline --ppo-iters 0 --steps 2000 --warmup 200 > gpurun_out/r03/dyn_emul_$n.json 2> gpurun_out/r03/dyn_emul_$n.err
  echo done $n
done
for n in 65536 8192; do
  python bench.py --envs $n --no-cpu-baseline --ppo-iters 0 --steps 2000 --warmup 200 --dynamic-leg 0 --large-envs 0 --fused-leg 0 > gpurun_out/r03/kin_emul_$n.json 2> gpurun_out/r03/kin_emul_$n.err
  echo done kin $n
done
