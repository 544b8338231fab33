# usage: bash tools/build_variant.sh NAME [extra hipcc flags]  -> tools/tmp_libs/NAME.so (3D library of the current tree)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/tmp_libs
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -w -DWGS_DIM=3 "$@" wgsparkl_amd/csrc/capi.hip -o tools/tmp_libs/$name.so
echo built tools/tmp_libs/$name.so
