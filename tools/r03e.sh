set -x
python -m pytest tests -m gpu -x -q -k "team or plan_kernels or sharded or alter or pool_wait or c_driver or walk_queue or reference_search" > gpurun_out/r03e_tests.log 2>&1; tail -3 gpurun_out/r03e_tests.log
python bench.py --workload 50M --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03e_50M.json 2> gpurun_out/r03e.err
python bench.py --steps 10 --warmup 3 --mode partition --no-cpu-baseline > gpurun_out/r03e_partition.json 2>> gpurun_out/r03e.err
python bench.py --steps 10 --warmup 3 > gpurun_out/r03e_head.json 2>> gpurun_out/r03e.err
tail -3 gpurun_out/r03e.err
