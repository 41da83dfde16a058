#!/bin/bash
# Numerics experiment: libraries whose split-bf16 operand producers (GroupNorm / LayerNorm apply, rf_split_bf16) first round the fp32 value to
# fp16 (a16.so) or to bf16 (abf16.so) -- the A operand an "fp16 activations x split-fp16 weights" two-pass GEMM mode would see.  The attention
# kernel keeps its full split.  The shipped sources are not touched: norm.hip / elementwise.hip are compiled from a scratch copy whose common.h
# has the rounding patched into split4_bf16.    tools/build_a16_variants.sh  ->  reface_amd/lib/alt/{a16,abf16}.so  (use with REFACE_HIP_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/reface_amd/lib/alt
for tag in a16 abf16; do
  T0=$(mktemp -d); T=$T0/a/b; mkdir -p $T; ln -s $ROOT/include $T0/include          # common.h includes "../../include/reface_hip.h"
  cp $ROOT/reface_amd/csrc/*.h $ROOT/reface_amd/csrc/norm.hip $ROOT/reface_amd/csrc/elementwise.hip $T/
  python3 - $T/common.h $tag <<'PY'
import sys
p, tag = sys.argv[1], sys.argv[2]
s = open(p).read()
old = """__device__ __forceinline__ void split4_bf16(const float* f, u32x2_t& hi, u32x2_t& lo) {
"""
cast = "_Float16" if tag == "a16" else "__bf16"
new = """__device__ __forceinline__ void split4_bf16(const float* f_in, u32x2_t& hi, u32x2_t& lo) {
    const float f[4] = {(float)(%s)f_in[0], (float)(%s)f_in[1], (float)(%s)f_in[2], (float)(%s)f_in[3]};
""" % ((cast,) * 4)
assert old in s
open(p, "w").write(s.replace(old, new))
PY
  objs=""
  for u in gemm norm attention elementwise encoder ffn; do
    if [ $u = norm ] || [ $u = elementwise ]; then
      flags=$(cd $ROOT && python -m reface_amd.build --print-flags $u)
      /opt/rocm/bin/hipcc $flags -I$T -I$ROOT/include -c $T/$u.hip -o $ROOT/reface_amd/lib/alt/$tag.$u.o
      objs="$objs $ROOT/reface_amd/lib/alt/$tag.$u.o"
    else objs="$objs $ROOT/reface_amd/lib/$u.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/reface_amd/lib/alt/$tag.so $objs
  rm -rf $T0
  echo $ROOT/reface_amd/lib/alt/$tag.so
done
