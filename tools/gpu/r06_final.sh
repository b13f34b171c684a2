# END OF ROUND 6: everything the records quote, on the final libraries (run on the GPU box):  bash tools/gpu/r06_final.sh <tag>
set -u
T=${1:-r06_final}; O=gpurun_out/$T; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2> $O/bench_wall.txt
timeout 500 bash tools/profile_round.sh ${T}_prof > $O/profile_round.log 2>&1
timeout 700 bash tools/profile_batch.sh ${T}_batch C4 > $O/profile_batch_c4.log 2>&1
timeout 900 bash tools/profile_batch.sh ${T}_batch C5 > $O/profile_batch_c5.log 2>&1
timeout 200 python tools/phase_profile.py > $O/phase_c2.log 2>&1
timeout 200 python tools/split_trace.py > $O/split_trace_c2.log 2>&1
mkdir -p $O/profiles_new; cp profiles/${T}_prof* profiles/${T}_batch* profiles/latest_pmc.json $O/profiles_new/ 2>/dev/null
env -u WORLD_SIZE -u RANK -u LOCAL_RANK timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-batch-configs --no-cpu-baseline > $O/bench_2ranks.json 2> $O/bench_2ranks.err
find gpurun_out -name "*.db" -delete
tail -n 4 $O/pytest_gpu.log; cut -c1-400 $O/bench.json; cat $O/bench_wall.txt; head -n 10 $O/profiles_new/${T}_prof_rocprof.md; cut -c1-300 $O/bench_2ranks.json
