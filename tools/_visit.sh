export TC_GIT_SHA=0bcc385
timeout 200 python -m pytest tests/test_gpu_route.py tests/test_gpu_sharding.py -x -q 2>&1 | tail -3
timeout 200 python tools/route_bound.py 1 100 2>&1 | tail -4; timeout 200 python tools/route_bound.py 8 60 2>&1 | tail -4
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
TC_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 10 > gpurun_out/dist1.json 2> gpurun_out/dist1.err; tail -2 gpurun_out/dist1.err
python - <<'PY'
import json
for l in open('gpurun_out/dist1.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d.get('zipf_stream',{}).get('value'))
PY
