#!/bin/bash
# usage: r04_variants.sh "<ENV=VAL,ENV=VAL>" ...   (one halo_overhead.py run per argument; "default" = no extra environment)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  echo "=== $v"
  ( if [ "$v" != default ]; then for kv in ${v//,/ }; do export "$kv"; done; fi
    T8GPU_HALO_ONLY=c T8GPU_STEPPER_PROFILE=1 timeout -k 10 300 python3 $ROOT/scripts/halo_overhead.py ${HALO_ARGS:-8 3 200} 2>&1 | grep -v "amdgpu.ids\|version\|Hostname\|Librccl" )
done
