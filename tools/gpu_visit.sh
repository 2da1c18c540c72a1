#!/bin/bash
# One GPU-box visit of round 3 (run from the repo root): probe, parity tests, knob A/B, the bench line, profiles.
# usage: bash tools/gpu_visit.sh <tag> [steps...]   steps: tests quick ab bench prof timeline
set -u
TAG=${1:-r03_v1}; shift
STEPS=${@:-tests ab bench prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
export TC_GIT_SHA=$(cat $R/.git_sha 2>/dev/null || echo unknown)
for S in $STEPS; do
  case $S in
    tests) timeout 400 python -m pytest tests -m gpu -x -q --timeout 120 > $O/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -5 $O/pytest_gpu.log ;;
    quick) timeout 600 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_fixed.py tests/test_gpu_slots.py -m gpu -x -q > $O/pytest_quick.log 2>&1; echo "quick rc=$?"; tail -15 $O/pytest_quick.log ;;
    general) timeout 600 python -m pytest tests/test_gpu_slots.py tests/test_gpu_fixed.py tests/test_gpu_fuzz.py tests/test_gpu_robustness.py -m gpu -x -q > $O/pytest_general.log 2>&1; echo "general tests rc=$?"; tail -4 $O/pytest_general.log
             for WL in general general_zipf; do timeout 300 python bench.py --workload $WL --no-also --no-cpu --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['stream'], round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s', 'kernel', d.get('roofline',{}).get('avg_ms'))"; done ;;
    generalq) for WL in general general_zipf; do timeout 300 python bench.py --workload $WL --no-also --no-cpu --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['stream'], round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s', 'kernel', d.get('roofline',{}).get('avg_ms'))"; done ;;
    dist) timeout 300 python -m pytest tests/test_gpu_route.py tests/test_gpu_sharding.py -m gpu -x -q > $O/pytest_route.log 2>&1; echo "route tests rc=$?"; tail -4 $O/pytest_route.log
          for RT in exchange replicate; do
            TC_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --route $RT --workload zipf > $O/dist2_${RT}.txt 2> $O/dist2_${RT}.err; echo "2 ranks on one GPU, $RT zipf rc=$?"; tail -1 $O/dist2_${RT}.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['route'], round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s  imbalance', d['imbalance_max_over_mean'], 'router ms', d.get('router_ms_per_step'))"; grep -i "falling back\|Traceback" $O/dist2_${RT}.err | head -3
            for WL in uniform zipf; do
              TC_BENCH_FORCE_DIST=1 timeout 90 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 100 --warmup 10 --route $RT --workload $WL > $O/dist1_${RT}_$WL.txt 2> $O/dist1_${RT}_$WL.err; echo "1 rank, multi-GPU code path, $RT $WL rc=$?"; tail -1 $O/dist1_${RT}_$WL.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['route'], round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s router ms', d.get('router_ms_per_step'), 'eval', (d.get('roofline') or {}).get('avg_ms'))"; grep -i "falling back\|Traceback" $O/dist1_${RT}_$WL.err | head -3
            done
          done ;;
    dist1) timeout 300 python -m pytest tests/test_gpu_route.py tests/test_gpu_sharding.py -m gpu -x -q > $O/pytest_route.log 2>&1; echo "route tests rc=$?"; tail -2 $O/pytest_route.log
           for RT in exchange replicate; do for WL in uniform zipf; do
              TC_BENCH_FORCE_DIST=1 timeout 90 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 100 --warmup 10 --route $RT --workload $WL > $O/dist1_${RT}_$WL.txt 2> $O/dist1_${RT}_$WL.err; echo "1 rank $RT $WL rc=$?"; tail -1 $O/dist1_${RT}_$WL.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['route'], round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s router ms', d.get('router_ms_per_step'), 'eval', (d.get('roofline') or {}).get('avg_ms'))"; grep "host us" $O/dist1_${RT}_$WL.err
           done; done ;;
    dist1prof) cd /tmp; export TMPDIR=/tmp
           RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 TC_BENCH_FORCE_DIST=1 timeout 120 rocprofv3 --kernel-trace --stats -d $O/x_stats -o s -- python $R/bench.py --gpus 1 --steps 40 --warmup 10 --route exchange > $O/x_stats.log 2>&1; echo "rc=$?"
           cd $R; python tools/trace_seq.py $O/x_stats -520 160 > $O/trace_seq.txt 2>&1; python tools/summarize_prof.py ${TAG}_exchange_1rank $O/x_stats > /dev/null 2>&1; head -24 profiles/${TAG}_exchange_1rank.txt; mkdir -p $O/profiles; cp profiles/${TAG}_exchange_1rank* $O/profiles/; rm -rf $O/x_stats ;;
    keysprof) bash tools/gpu_round.sh $TAG string_keys string_keys_long ;;
    keysq) timeout 200 python tools/keys_only.py 2>/dev/null | tail -1 ;;
    keys) timeout 200 python tools/keys_only.py 2>/dev/null | tail -1
          cd /tmp; export TMPDIR=/tmp
          timeout 200 rocprofv3 --kernel-trace --stats -d $O/k_stats -o s -- python $R/tools/profile_keys.py short 16 > $O/k_stats.log 2>&1; echo "rc=$?"
          cd $R; python tools/trace_seq.py $O/k_stats -420 420 > $O/keys_trace_seq.txt 2>&1; python tools/summarize_prof.py ${TAG}_string_keys_trace $O/k_stats > /dev/null 2>&1; mkdir -p $O/profiles; cp profiles/${TAG}_string_keys_trace* $O/profiles/; rm -rf $O/k_stats ;;
    keytests) timeout 600 python -m pytest tests/test_gpu_keys.py tests/test_gpu_keys_spec.py tests/test_gpu_metrics.py tests/test_gpu_snapshot.py -m gpu -x -q > $O/pytest_keys.log 2>&1; echo "key tests rc=$?"; tail -4 $O/pytest_keys.log ;;
    routeflaky) for i in $(seq 1 40); do timeout 120 python -m pytest tests/test_gpu_route.py -m gpu -x -q --timeout 60 > $O/rf_$i.log 2>&1; if grep -q failed $O/rf_$i.log; then echo "run $i FAILED"; grep -E "^E  |FAILED|assert" $O/rf_$i.log | head -12; else rm $O/rf_$i.log; fi; done; echo loop done ;;
    ab) timeout 600 python tools/ab_step.py 100 > $O/ab_step.txt 2>&1; echo "ab rc=$?"; cat $O/ab_step.txt ;;
    bench) TC_BENCH_VERBOSE=1 timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"
           tail -c 4200 $O/bench_stdout.txt; echo; wc -c $O/bench_stdout.txt; cp gpurun_out/bench_detail.json $O/ 2>/dev/null ;;
    prof) bash tools/gpu_round.sh $TAG uniform_fixed zipf_fixed general_uniform_fixed general_zipf_fixed ;;
    timeline) cd /tmp; export TMPDIR=/tmp
           for WL in uniform zipf; do
             timeout 240 rocprofv3 --kernel-trace -d $O/tl_$WL -o t -- python $R/bench.py --profile-run --steps 40 --warmup 5 --layout fixed --workload $WL > $O/tl_$WL.log 2>&1
             python $R/tools/timeline.py $O/tl_$WL > $O/timeline_$WL.txt 2>&1; tail -30 $O/timeline_$WL.txt; rm -rf $O/tl_$WL
           done; cd $R ;;
    *) echo "unknown step $S" ;;
  esac
done
