#!/bin/bash
# Builds the two C-ABI libraries (one per dimension, like the reference's two crates)
# for gfx950 with hipcc. Outputs land next to the sources: libwgsparkl{2,3}d_hip.so.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
ARCH=${WGS_ARCH:-gfx950}
FLAGS="-O3 -std=c++17 -fPIC -shared --offload-arch=${ARCH} -fno-fast-math -Wall -Wno-unused-variable -Wno-unused-but-set-variable -Wno-unused-value -Wno-unused-result ${WGS_EXTRA_FLAGS:-}"
pids=()
for dim in 3 2; do
  out="libwgsparkl${dim}d_hip.so"
  if [[ "${1:-}" != "force" && -f "$out" ]]; then
    newest=$(ls -t capi.hip *.h *.inc ../../include/wgsparkl_hip.h "$out" | head -1)
    [[ "$newest" == "$out" ]] && continue
  fi
  $HIPCC $FLAGS -DWGS_DIM=$dim capi.hip -o "$out.tmp" && mv "$out.tmp" "$out" &
  pids+=($!)
done
rc=0
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && { wait "$p" || rc=1; }; done
exit $rc
