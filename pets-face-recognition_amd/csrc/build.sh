#!/bin/bash
# Builds libpfr_hip.so (gfx950 only) next to the sources. Usage: build.sh [-j N]
set -e
cd "$(dirname "$0")"
ARCH=gfx950
FLAGS="--offload-arch=$ARCH -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast ${PFR_EXTRA_FLAGS}"
mkdir -p build
pids=()
python3 ../../tools/gen_thunks.py > /dev/null
for f in pfr_api pfr_comm pfr_plan pfr_igemm pfr_igemm_p pfr_sconv pfr_sconv3 pfr_wgrad pfr_wgrad9 pfr_elementwise pfr_bnfree pfr_head pfr_match pfr_swin pfr_augment; do
  [ -f $f.hip ] || continue
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ pfr_common.h -nt build/$f.o ] || [ pfr_mma.h -nt build/$f.o ] || [ pfr_igemm.h -nt build/$f.o ] || [ ../../include/pfr_hip.h -nt build/$f.o -a $f = pfr_plan ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=$ARCH -shared -fPIC build/*.o -ldl -o libpfr_hip.so
echo "built $(pwd)/libpfr_hip.so"
