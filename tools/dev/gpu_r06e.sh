export TMPDIR=/tmp
timeout 1300 python -u -m pytest tests -m gpu -q -x --timeout=500 --durations=8 > gpurun_out/r06e_pytest_gpu.log 2>&1
tail -14 gpurun_out/r06e_pytest_gpu.log
ORZ_FAST_SCHED=0 timeout 300 python tools/dev/lib_ab.py - > gpurun_out/r06e_ab_sched0.jsonl 2>&1
timeout 300 python tools/dev/lib_ab.py - > gpurun_out/r06e_ab_sched1.jsonl 2>&1
ORZ_FAST_SCHED=0 timeout 300 python tools/dev/lib_ab.py - >> gpurun_out/r06e_ab_sched0.jsonl 2>&1
timeout 300 python tools/dev/lib_ab.py - >> gpurun_out/r06e_ab_sched1.jsonl 2>&1
echo "sched off:"; cut -c1-420 gpurun_out/r06e_ab_sched0.jsonl
echo "sched on:"; cut -c1-420 gpurun_out/r06e_ab_sched1.jsonl
