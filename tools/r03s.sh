set -x
python -m pytest tests -m gpu -x -q -k "small_graphs or noisy or cutoff or tie or hub or handmade or poison or component_pool or medium or misjoin" > gpurun_out/r03s_tests.log 2>&1; tail -3 gpurun_out/r03s_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
$B --opt small_masks=0 > gpurun_out/r03s_m0.json 2> gpurun_out/r03s.err
$B --opt small_masks=1 > gpurun_out/r03s_m1.json 2>> gpurun_out/r03s.err
$B --opt small_masks=0 > gpurun_out/r03s_m0b.json 2>> gpurun_out/r03s.err
$B --opt small_masks=1 > gpurun_out/r03s_m1b.json 2>> gpurun_out/r03s.err
tail -2 gpurun_out/r03s.err
