cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_patch.py tests/test_gpu_graph.py -x -q 2>&1 | tail -2
bash scripts/ab_env.sh T8GPU_PATCH3_FORK "1 0" --workload c5 --steps 50 --reps 3
bash scripts/ab_env.sh T8GPU_PATCH3_FORK "1 0" --workload c5u --steps 50 --reps 3
