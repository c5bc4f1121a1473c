#!/bin/bash
# Profiling build of the kernel library (-DPFR_IGEMM_TRACE: workgroup phase stamps in the GEMM kernels) next to the product
# one: pets-face-recognition_amd/csrc/libpfr_hip_trace.so; use with PFR_LIB_PATH=<that file> python tools/p_trace.py
set -e
cd "$(dirname "$0")/../pets-face-recognition_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast -DPFR_IGEMM_TRACE"
mkdir -p build_trace
pids=()
for f in pfr_api pfr_comm pfr_plan pfr_igemm pfr_igemm_p pfr_sconv pfr_sconv3 pfr_sstem pfr_slin pfr_wgrad pfr_wgrad9 pfr_elementwise pfr_bnfree pfr_head pfr_match pfr_swin pfr_augment; do
  if [ ! -f build_trace/$f.o ] || [ $f.hip -nt build_trace/$f.o ] || [ pfr_igemm.h -nt build_trace/$f.o ] || [ pfr_mma.h -nt build_trace/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build_trace/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build_trace/*.o -ldl -o libpfr_hip_trace.so
echo "built $(pwd)/libpfr_hip_trace.so"
