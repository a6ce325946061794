export TMPDIR=/tmp
timeout 1100 python -u -m pytest tests -m gpu -q -x --timeout=400 --durations=8 > gpurun_out/r06b_pytest_gpu.log 2>&1
tail -15 gpurun_out/r06b_pytest_gpu.log
timeout 300 python bench.py --steps 3 --warmup 1 2>gpurun_out/r06b_bench.err | tail -1 > gpurun_out/r06b_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r06b_bench.json'))
print(d['value'], d['ms_per_step'], d['compressed_bytes'], d['size_delta_pct'], d['roundtrip_ok'], d.get('host_syncs_per_block'), d['kernel_table']['sum_ms_per_block_without_symbol_ranking'])
for k in ['members','members_l2_text','members_l2_zeros']:
    m=d.get(k,{}); print(k, {x: m.get(x) for x in ['value','size_delta_pct','roundtrip_ok','error','gpu_over_cpu_members']})
print(d.get('cpu_baseline_members',{}).get('value'))
"
tail -3 gpurun_out/r06b_bench.err
timeout 200 python tools/dev/members_scale.py 8 8 2>&1 | tail -3
timeout 600 python tools/dev/tile_sweep.py > gpurun_out/r06b_tile_sweep.jsonl 2>gpurun_out/r06b_tile_sweep.err; cat gpurun_out/r06b_tile_sweep.jsonl; tail -3 gpurun_out/r06b_tile_sweep.err
