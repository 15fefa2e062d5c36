#!/bin/bash
# North-star slide (synthetic 40 000^2, 31 684 tiles of 256 / 32) through bench.py: one GPU for real, and the driver's
# N = 2 / 4 / 8 launch lines as gloo dry runs (all ranks share the box's one GPU: sharding, barrier / max-over-ranks
# timing and the record all-gather run end to end; the rates are one GPU's, divided).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py --slide 40000 --steps 200 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_40k_1gpu.json 2> gpurun_out/r06_bench_40k_1gpu.err
for n in 2 4 8; do
  CPX_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) \
    bench.py --gpus $n --slide 40000 --steps 6 --warmup 2 --no-stages --no-cpu-baseline --no-side-lines \
    > gpurun_out/r06_bench_40k_${n}rank_gloo_dryrun.json 2> gpurun_out/r06_bench_40k_${n}rank_gloo_dryrun.err
done
tail -c 400 gpurun_out/r06_bench_40k_1gpu.json; for n in 2 4 8; do head -c 300 gpurun_out/r06_bench_40k_${n}rank_gloo_dryrun.json; echo; done
