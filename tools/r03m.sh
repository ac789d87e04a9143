set -x
python -m pytest tests -m gpu -x -q -k "deparse or component_pool or poison or pool_wait or small_graphs" > gpurun_out/r03m_tests.log 2>&1; tail -3 gpurun_out/r03m_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
$B > gpurun_out/r03m_fill4.json 2> gpurun_out/r03m.err
$B --opt pool_fill_kb=0 > gpurun_out/r03m_fill0.json 2>> gpurun_out/r03m.err
$B --opt pool_fill_kb=8 > gpurun_out/r03m_fill8.json 2>> gpurun_out/r03m.err
$B --opt pool_fill_kb=4 > gpurun_out/r03m_fill4b.json 2>> gpurun_out/r03m.err
tail -3 gpurun_out/r03m.err
