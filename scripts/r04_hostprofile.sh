cd /tmp && export TMPDIR=/tmp && export T8GPU_STEPPER_PROFILE=1 && timeout -k 10 500 python3 $GRAFT_REPO_ROOT/scripts/halo_overhead.py 8 3 200 2>&1 | grep -v amdgpu.ids
