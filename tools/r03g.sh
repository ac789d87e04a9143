set -x
python -m pytest tests -m gpu -x -q -k "plan_kernels or team or route_kernels or sharded_pipeline" > gpurun_out/r03g_tests.log 2>&1; tail -3 gpurun_out/r03g_tests.log
python bench.py --workload 50M --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03g_50M.json 2> gpurun_out/r03g.err
python bench.py --steps 10 --warmup 3 --mode partition --no-cpu-baseline > gpurun_out/r03g_partition.json 2>> gpurun_out/r03g.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03g_head.json 2>> gpurun_out/r03g.err
GTS_ENGINE_LIB=$PWD/gt-scaffold_amd/csrc/libgtscaffold_hip_l16.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03g_head_l16.json 2>> gpurun_out/r03g.err
tail -3 gpurun_out/r03g.err
