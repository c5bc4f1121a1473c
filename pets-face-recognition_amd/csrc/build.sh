#!/bin/bash
# Builds libpfr_hip.so (gfx950 only) next to the sources. Usage: build.sh [-j N]
# The object list is EXPLICIT (a stale object of a retired source in build/ is never linked) and every object depends on every header:
# the package's own (pfr_*.h), the public C-ABI header and the generated thunk table.
set -e
cd "$(dirname "$0")"
ARCH=gfx950
FLAGS="--offload-arch=$ARCH -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast ${PFR_EXTRA_FLAGS}"
SOURCES="pfr_api pfr_comm pfr_plan pfr_igemm pfr_igemm_p pfr_sconv pfr_sconv3 pfr_sstem pfr_slin pfr_wgrad pfr_wgrad9 pfr_elementwise pfr_bnfree pfr_head pfr_match pfr_swin pfr_augment"
HEADERS="pfr_common.h pfr_mma.h pfr_igemm.h ../../include/pfr_hip.h pfr_thunks_gen.inc"
# Variant builds for A/B runs: PFR_BUILD_TAG=nt PFR_EXTRA_FLAGS="-DPFR_LN_NT=1" build.sh  ->  build_nt/*.o, libpfr_hip_nt.so (load it with
# PFR_LIB_PATH=...).  The flags of the last build are recorded next to the objects: a change of flags rebuilds everything (the objects'
# time stamps alone would silently link objects of the previous flags).
B=build${PFR_BUILD_TAG:+_$PFR_BUILD_TAG}
OUT=libpfr_hip${PFR_BUILD_TAG:+_$PFR_BUILD_TAG}.so
mkdir -p $B
if [ ! -f $B/.flags ] || [ "$(cat $B/.flags)" != "$FLAGS" ]; then rm -f $B/*.o; echo "$FLAGS" > $B/.flags; fi
# the thunk table is regenerated only when the public header changed (so that it does not look new on every build)
if [ ! -f pfr_thunks_gen.inc ] || [ ../../include/pfr_hip.h -nt pfr_thunks_gen.inc ] || [ ../../tools/gen_thunks.py -nt pfr_thunks_gen.inc ]; then
  python3 ../../tools/gen_thunks.py > /dev/null
fi
pids=()
objs=()
for f in $SOURCES; do
  [ -f $f.hip ] || { echo "build.sh: missing source $f.hip" >&2; exit 1; }
  objs+=($B/$f.o)
  stale=0
  [ -f $B/$f.o ] || stale=1
  for d in $f.hip $HEADERS; do [ $stale = 1 ] || [ ! $d -nt $B/$f.o ] || stale=1; done
  if [ $stale = 1 ]; then
    hipcc $FLAGS -c $f.hip -o $B/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=$ARCH -shared -fPIC "${objs[@]}" -ldl -o $OUT
echo "built $(pwd)/$OUT"
