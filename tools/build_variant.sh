#!/bin/bash
# Developer A/B helper: build libmi_oov with extra -D flags for one source (default lsh64.hip) into lib/ab/<name>.so
#   bash tools/build_variant.sh base ""; bash tools/build_variant.sh nt "-DMI_EXP_NT=1"; bash tools/build_variant.sh x "-DMI_EXP=1" score.hip
set -e
cd "$(dirname "$0")/../improving-inductive-oov-recsys_amd"
mkdir -p lib/ab
src=${3:-lsh64.hip}
make -C csrc -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 \
  -I../include -Wall -Wno-unused-function $2 -c csrc/$src -o lib/ab/$1.o
objs=""
for o in api.cpp lsh.hip lsh64.hip lsh64p.hip exchange.hip mlp.hip hash.hip gather.hip score.hip linear3.hip evalrows.hip; do
  if [ "$o" = "$src" ]; then objs="$objs lib/ab/$1.o"; else objs="$objs lib/obj/$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/ab/$1.so $objs
rm lib/ab/$1.o
echo built lib/ab/$1.so
