#!/bin/bash
# Build libtnqs_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libtnqs_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result ${EXTRA_FLAGS:-}"
[ "${EXPERIMENTS:-0}" = "1" ] && FLAGS="$FLAGS -DTNQS_EXPERIMENTS"      # kernel-experiment switches (engine_internal.hpp); never in the shipped build
mkdir -p build
pids=()
for f in kernels.hip kernels_mfma.hip kernels_x3.hip kernels_plane.hip kernels_chi64.hip kernels_gate.hip kernels_f64.hip engine_core.cpp engine_batch.cpp engine_bp.cpp engine_gates.cpp engine_obs.cpp sharding.cpp api.cpp debug.cpp; do
  [ -f "$f" ] || continue
  o=build/${f%.*}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ kernels.hpp -nt "$o" ] || [ engine.hpp -nt "$o" ] || [ engine_internal.hpp -nt "$o" ] || [ launch_util.hpp -nt "$o" ] || [ mfma_common.hpp -nt "$o" ] || [ x3_common.hpp -nt "$o" ] || [ ../../include/tnqs.h -nt "$o" ]; then
    if [[ "$f" == *.hip ]]; then hipcc $FLAGS -c "$f" -o "$o" & else hipcc $FLAGS -x hip -c "$f" -o "$o" & fi
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -ldl -o $OUT
echo "built $OUT"
