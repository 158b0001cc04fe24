#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the C-ABI (CPU only; GPU ASan is not available on this pool).
#   bash scripts/asan/run_asan_host.sh [out.txt]      (~4 minutes: every *_api.hip is recompiled with the host pass instrumented)
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${1:-$REPO/profiles/r05/asan_host.txt}
W=/tmp/dmpc_asan; rm -rf $W; mkdir -p $W
CS=$REPO/chainer_differentiable_mpc_amd/csrc
python3 $CS/build.py > /dev/null 2>&1      # (makes sure the generated headers exist)
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fno-slp-vectorize -I$REPO/include -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer"
pids=""
for f in $CS/*.hip; do
  ( hipcc $FLAGS -c $f -o $W/$(basename $f .hip).o 2> $W/$(basename $f .hip).log ) &
  pids="$pids $!"
done
wait $pids
hipcc $FLAGS -c $REPO/scripts/asan/asan_host_driver.cpp -o $W/driver.o 2> $W/driver.log
hipcc --offload-arch=gfx950 -fsanitize=address,undefined $W/*.o -o $W/asan_host_driver 2> $W/link.log || { cat $W/link.log | tail -20; exit 2; }
{ echo "# scripts/asan/run_asan_host.sh: host pass of every *_api.hip built with -fsanitize=address,undefined, driver on the CPU (no GPU)";
  echo "# library sources: $(python3 -c "import sys; sys.path.insert(0, '$CS'); import build; print(build.source_hash())")";
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 HIP_VISIBLE_DEVICES=-1 $W/asan_host_driver 2>&1 | tail -30;
  echo "exit code: ${PIPESTATUS[0]}"; } > $OUT
cat $OUT
