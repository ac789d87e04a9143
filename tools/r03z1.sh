set -x
python bench.py --workload 50M --steps 2 --warmup 1 --no-cpu-baseline --opt defer_global_components=1 > gpurun_out/r03z1_50M_defer.json 2> gpurun_out/r03z1.err
tail -2 gpurun_out/r03z1.err
