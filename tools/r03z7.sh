set -x
python bench.py --inversions 0.1 --duplicate-pairs --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03z7_inv.json 2> gpurun_out/r03z7.err
python bench.py --workload 50M --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r03z7_50M.json 2>> gpurun_out/r03z7.err
python bench.py --steps 10 --warmup 3 --mode partition --no-cpu-baseline --no-secondary > gpurun_out/r03z7_partition.json 2>> gpurun_out/r03z7.err
tail -2 gpurun_out/r03z7.err
