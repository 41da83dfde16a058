#!/bin/bash
# Build an alternate libreface_hip.so with one replaced .hip source, for same-box A/B runs:
#   tools/build_variant.sh gemm /tmp/gemm_old.hip base   ->  reface_amd/lib/alt/base.so   (use with REFACE_HIP_LIB=...)
# Compile flags come from reface_amd/build.py (one source of truth), plus -DRF_EXPERIMENT: only variant builds read the RF_* tuning /
# timing-decomposition environment switches (csrc/common.h tune_env); $VARIANT_DEFS adds more defines.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
unit=$1; src=$2; tag=$3
mkdir -p $ROOT/reface_amd/lib/alt
flags=$(cd $ROOT && python -m reface_amd.build --print-flags $unit)
/opt/rocm/bin/hipcc $flags -DRF_EXPERIMENT $VARIANT_DEFS -I$ROOT/reface_amd/csrc -I$ROOT/include -c $src -o $ROOT/reface_amd/lib/alt/$tag.$unit.o
objs=""
for u in gemm gemm_f16 norm attention elementwise encoder ffn smallconv attnin; do
  if [ $u = $unit ]; then objs="$objs $ROOT/reface_amd/lib/alt/$tag.$unit.o"; else objs="$objs $ROOT/reface_amd/lib/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/reface_amd/lib/alt/$tag.so $objs
if nm -u -C $ROOT/reface_amd/lib/alt/$tag.so | grep -q "rf::"; then echo "variant has undefined rf:: symbols (host stubs missing):"; nm -u -C $ROOT/reface_amd/lib/alt/$tag.so | grep "rf::" | head -4; exit 1; fi
echo $ROOT/reface_amd/lib/alt/$tag.so
