#!/bin/bash
# tools/q_variant_build.sh NAME -DFLAG...: den_lazy.hip compiled with the flags, linked with the library's other objects into tools/variants/NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $ROOT/tools/variants $ROOT/build/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -I $ROOT/include -c $ROOT/pychain_amd/csrc/den_lazy.hip -o $ROOT/build/var_$name/den_lazy.hip.o 2>/dev/null
objs=$(ls $ROOT/build/obj/*.o | grep -v den_lazy.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $ROOT/build/var_$name/den_lazy.hip.o -o $ROOT/tools/variants/$name.so
echo built $name
