#!/bin/bash
# Builds the two C-ABI libraries (one per dimension, like the reference's two crates)
# for gfx950 with hipcc. Outputs land next to the sources: libwgsparkl{2,3}d_hip.so.
# -ffp-contract=on: multiply-adds are fused where the SOURCE writes them in one expression (a front-end decision),
# never by the back end across statements: the rounding of a kernel body then does not depend on which kernel it is
# compiled into (the paired and the separate launches of a body are bit-identical by construction), and it is
# measurably faster than the default "fast" here (P2G 37 -> 35 us, G2P 43 -> 41 us at C2).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
ARCH=${WGS_ARCH:-gfx950}
FLAGS="-O3 -std=c++17 -fPIC -shared --offload-arch=${ARCH} -fno-fast-math -ffp-contract=on -Wall -Wno-unused-variable -Wno-unused-but-set-variable -Wno-unused-value -Wno-unused-result ${WGS_EXTRA_FLAGS:-}"
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
