set -x
python -m pytest tests -m gpu -x -q -k "plan_kernels or team" > gpurun_out/r03f_tests.log 2>&1; tail -3 gpurun_out/r03f_tests.log
python bench.py --workload 50M --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03f_50M.json 2> gpurun_out/r03f.err
python bench.py --steps 10 --warmup 3 --mode partition --no-cpu-baseline > gpurun_out/r03f_partition.json 2>> gpurun_out/r03f.err
tail -3 gpurun_out/r03f.err
