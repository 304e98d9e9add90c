cd $GRAFT_REPO_ROOT
timeout -k 10 400 python scripts/fuzz_parity.py 240 31 > gpurun_out/r3_fuzz_parity.log 2>&1; echo "parity rc=$?"; tail -12 gpurun_out/r3_fuzz_parity.log
timeout -k 10 400 python scripts/fuzz_partition.py 200 32 > gpurun_out/r3_fuzz_partition.log 2>&1; echo "partition rc=$?"; tail -3 gpurun_out/r3_fuzz_partition.log
timeout -k 10 300 python scripts/fuzz_adapt.py 120 33 > gpurun_out/r3_fuzz_adapt.log 2>&1; echo "adapt rc=$?"; tail -3 gpurun_out/r3_fuzz_adapt.log
