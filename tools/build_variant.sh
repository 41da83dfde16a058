#!/bin/bash
# Build an alternate libreface_hip.so with one replaced .hip source, for same-box A/B runs:
#   tools/build_variant.sh gemm /tmp/gemm_old.hip base   ->  reface_amd/lib/alt/base.so   (use with REFACE_HIP_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
unit=$1; src=$2; tag=$3
mkdir -p $ROOT/reface_amd/lib/alt
extra=""
[ "$unit" = attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $extra $VARIANT_DEFS -I$ROOT/reface_amd/csrc -I$ROOT/include -c $src -o $ROOT/reface_amd/lib/alt/$tag.$unit.o
objs=""
for u in gemm norm attention elementwise encoder ffn; do
  if [ $u = $unit ]; then objs="$objs $ROOT/reface_amd/lib/alt/$tag.$unit.o"; else objs="$objs $ROOT/reface_amd/lib/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/reface_amd/lib/alt/$tag.so $objs
echo $ROOT/reface_amd/lib/alt/$tag.so
