export TMPDIR=/tmp
timeout 1300 python -u -m pytest tests -m gpu -q -x --timeout=500 --durations=8 > gpurun_out/r06f_pytest_gpu.log 2>&1
tail -14 gpurun_out/r06f_pytest_gpu.log
timeout 300 python tools/dev/lib_ab.py - > gpurun_out/r06f_ab.jsonl 2>&1
cut -c1-700 gpurun_out/r06f_ab.jsonl
