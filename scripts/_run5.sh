cd $GRAFT_REPO_ROOT
python scripts/host_cycle_time.py 3 6 8 0.03 2>&1 | grep -v "^\[tile_plan\] \(patches\|tile classes\|greedy\)"
timeout -k 10 500 python bench.py --workload c5a --steps 100 --warmup 5 --no-cpu-baseline 2> gpurun_out/r3_c5a.err | tail -1 > gpurun_out/r3_c5a.json; cat gpurun_out/r3_c5a.json
timeout -k 10 600 python -m pytest tests/test_gpu_amr.py -x -q 2>&1 | tail -2
