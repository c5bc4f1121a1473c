#!/bin/bash
# Builds libpfr_hip.so (gfx950 only) next to the sources. Usage: build.sh [-j N]
# The object list is EXPLICIT (a stale object of a retired source in build/ is never linked) and every object depends on every header:
# the package's own (pfr_*.h), the public C-ABI header and the generated thunk table.
set -e
cd "$(dirname "$0")"
ARCH=gfx950
FLAGS="--offload-arch=$ARCH -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast ${PFR_EXTRA_FLAGS}"
SOURCES="pfr_api pfr_comm pfr_plan pfr_igemm pfr_igemm_p pfr_sconv pfr_sconv3 pfr_slin pfr_wgrad pfr_wgrad9 pfr_elementwise pfr_bnfree pfr_head pfr_match pfr_swin pfr_augment"
HEADERS="pfr_common.h pfr_mma.h pfr_igemm.h ../../include/pfr_hip.h pfr_thunks_gen.inc"
mkdir -p build
# the thunk table is regenerated only when the public header changed (so that it does not look new on every build)
if [ ! -f pfr_thunks_gen.inc ] || [ ../../include/pfr_hip.h -nt pfr_thunks_gen.inc ] || [ ../../tools/gen_thunks.py -nt pfr_thunks_gen.inc ]; then
  python3 ../../tools/gen_thunks.py > /dev/null
fi
pids=()
objs=()
for f in $SOURCES; do
  [ -f $f.hip ] || { echo "build.sh: missing source $f.hip" >&2; exit 1; }
  objs+=(build/$f.o)
  stale=0
  [ -f build/$f.o ] || stale=1
  for d in $f.hip $HEADERS; do [ $stale = 1 ] || [ ! $d -nt build/$f.o ] || stale=1; done
  if [ $stale = 1 ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=$ARCH -shared -fPIC "${objs[@]}" -ldl -o libpfr_hip.so
echo "built $(pwd)/libpfr_hip.so"
