mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/r3/pytest5.log 2>&1 ; echo rc=$? >> gpurun_out/r3/pytest5.log
tail -n 3 gpurun_out/r3/pytest5.log
timeout -k 10 600 python bench.py > gpurun_out/r3/bench1.json 2> gpurun_out/r3/bench1.err; echo bench rc=$?
tail -c 600 gpurun_out/r3/bench1.err
