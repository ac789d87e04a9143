set -x
python -m pytest tests -m gpu -x -q -k "small_graphs or noisy or poison or component_pool or pool_wait or medium or deferred" > gpurun_out/r03v_tests.log 2>&1; tail -3 gpurun_out/r03v_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
$B > gpurun_out/r03v_64s4.json 2> gpurun_out/r03v.err
$B --opt batch_big_contigs=32 > gpurun_out/r03v_32s4.json 2>> gpurun_out/r03v.err
$B --opt batch_big_slots=6 > gpurun_out/r03v_64s6.json 2>> gpurun_out/r03v.err
$B --opt batch_big_contigs=96 > gpurun_out/r03v_96s4.json 2>> gpurun_out/r03v.err
$B --opt batch_big_slots=3 > gpurun_out/r03v_64s3.json 2>> gpurun_out/r03v.err
tail -2 gpurun_out/r03v.err
