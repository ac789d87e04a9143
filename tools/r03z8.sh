set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r03z8_tests.log 2>&1; tail -3 gpurun_out/r03z8_tests.log
python bench.py --inversions 0.1 --duplicate-pairs --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03z8_inv.json 2> gpurun_out/r03z8.err
python bench.py --workload 50M --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r03z8_50M.json 2>> gpurun_out/r03z8.err
tail -2 gpurun_out/r03z8.err
