set -x
bash tools/profile_round.sh r03o > gpurun_out/r03o_profile.log 2>&1; tail -4 gpurun_out/r03o_profile.log
GTS_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --contigs 2000000 > gpurun_out/r03o_2rank.json 2> gpurun_out/r03o_2rank.err
tail -2 gpurun_out/r03o_2rank.err
python bench.py --inversions 0.1 --duplicate-pairs --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03o_inv.json 2>> gpurun_out/r03o.err
