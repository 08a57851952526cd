mkdir -p gpurun_out/r05b
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -15 > gpurun_out/r05b/tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05b/bench.json 2> gpurun_out/r05b/bench.err
tail -5 gpurun_out/r05b/tests.log; tail -c 300 gpurun_out/r05b/bench.err
