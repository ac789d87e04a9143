set -x
python -m pytest tests/test_gpu_parity.py -x -q -k "not full_size and not 50M" > gpurun_out/r03c_tests.log 2>&1; tail -3 gpurun_out/r03c_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
$B > gpurun_out/r03c_head.json 2> gpurun_out/r03c.err &&
$B --inversions 0.1 --duplicate-pairs > gpurun_out/r03c_inv.json 2>> gpurun_out/r03c.err &&
python bench.py --workload 50M --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03c_50M.json 2>> gpurun_out/r03c.err
