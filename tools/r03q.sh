set -x
python -m pytest tests -m gpu -x -q -k "reference_search or walk_queue or poison or handmade or deferred or small_graphs or noisy or tie" > gpurun_out/r03q_tests.log 2>&1; tail -3 gpurun_out/r03q_tests.log
python bench.py --inversions 0.1 --duplicate-pairs --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03q_inv.json 2> gpurun_out/r03q.err
tail -2 gpurun_out/r03q.err
