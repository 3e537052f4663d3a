mkdir -p gpurun_out/r3
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/r3/pytest3.log 2>&1 ; echo rc=$? >> gpurun_out/r3/pytest3.log
tail -n 30 gpurun_out/r3/pytest3.log
