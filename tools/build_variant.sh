#!/bin/bash
# A/B builds of the kernel library: tools/build_variant.sh <output.so> <extra hipcc flags...>   (objects in a scratch dir; the
# product library is untouched).  Select the variant at run time with --backend <output.so> (bench.py: --shim-flags).
OUT=$1; shift
D=$(mktemp -d /tmp/ffh_variant_XXXX)
for f in dlrm_flexflow_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -I include "$@" -c $f -o $D/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $D/*.o && rm -rf $D && echo built $OUT
