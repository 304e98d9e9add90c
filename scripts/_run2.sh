cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_graph.py tests/test_gpu_fused.py tests/test_gpu_patch.py -x -q > gpurun_out/r3_tests_b.log 2>&1; rc=$?
tail -5 gpurun_out/r3_tests_b.log
[ $rc -eq 0 ] || exit $rc
T8GPU_TEST_RCCL_CAPTURE=1 timeout -k 10 600 python -m pytest tests/test_gpu_graph.py -k opt_in -q -rx > gpurun_out/r3_rccl_capture.log 2>&1
tail -5 gpurun_out/r3_rccl_capture.log
