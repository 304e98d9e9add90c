cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_patch.py -x -q 2>&1 | tail -2
bash scripts/ab_env.sh T8GPU_PATCH_WGS "3 2" --workload c4 --steps 50 --reps 3
