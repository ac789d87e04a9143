set -x
B="python bench.py --inversions 0.1 --duplicate-pairs --steps 5 --warmup 2 --no-cpu-baseline"
$B > gpurun_out/r03y_inv_s1.json 2> gpurun_out/r03y.err
$B --opt suspects_first=0 > gpurun_out/r03y_inv_s0.json 2>> gpurun_out/r03y.err
$B > gpurun_out/r03y_inv_s1b.json 2>> gpurun_out/r03y.err
$B --opt suspects_first=0 > gpurun_out/r03y_inv_s0b.json 2>> gpurun_out/r03y.err
python bench.py --workload 50M --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03y_50M.json 2>> gpurun_out/r03y.err
python bench.py --steps 10 --warmup 3 --mode partition --no-cpu-baseline > gpurun_out/r03y_partition.json 2>> gpurun_out/r03y.err
tail -2 gpurun_out/r03y.err
