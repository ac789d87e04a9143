set -x
python -m pytest tests -m gpu -x -q -k "distance_widths or packed_lds or small_graphs or noisy or tie or handmade or poison or component_pool or deferred or reference_search" > gpurun_out/r03t_tests.log 2>&1; tail -3 gpurun_out/r03t_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
$B --opt lds_int16_distances=0 > gpurun_out/r03t_d0.json 2> gpurun_out/r03t.err
$B --opt lds_int16_distances=1 > gpurun_out/r03t_d1.json 2>> gpurun_out/r03t.err
$B --opt lds_int16_distances=0 > gpurun_out/r03t_d0b.json 2>> gpurun_out/r03t.err
$B --opt lds_int16_distances=1 > gpurun_out/r03t_d1b.json 2>> gpurun_out/r03t.err
tail -2 gpurun_out/r03t.err
