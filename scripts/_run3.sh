cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_halo.py -x -q > gpurun_out/r3_tests_c.log 2>&1; rc=$?
tail -4 gpurun_out/r3_tests_c.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/halo_overhead.py 8 3 100 > gpurun_out/r3_halo_overhead_8_2class.log 2>&1; tail -6 gpurun_out/r3_halo_overhead_8_2class.log
T8GPU_STEPPER_CLASSES=3 timeout -k 10 300 python scripts/halo_overhead.py 8 3 100 > gpurun_out/r3_halo_overhead_8_3class.log 2>&1; tail -6 gpurun_out/r3_halo_overhead_8_3class.log
