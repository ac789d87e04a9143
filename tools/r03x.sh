set -x
python -m pytest tests -m gpu -x -q -k "small_graphs or noisy or poison or component_pool or self_loop or medium" > gpurun_out/r03x_tests.log 2>&1; tail -3 gpurun_out/r03x_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
$B > gpurun_out/r03x_s1.json 2> gpurun_out/r03x.err
$B --opt suspects_first=0 > gpurun_out/r03x_s0.json 2>> gpurun_out/r03x.err
$B > gpurun_out/r03x_s1b.json 2>> gpurun_out/r03x.err
$B --opt suspects_first=0 > gpurun_out/r03x_s0b.json 2>> gpurun_out/r03x.err
tail -2 gpurun_out/r03x.err
