#!/bin/bash
# Developer A/B helper: build libmi_oov with extra -D flags for lsh64.hip into lib/ab/<name>.so
#   bash tools/build_variant.sh base ""; bash tools/build_variant.sh nt "-DMI_EXP_NT=1"
set -e
cd "$(dirname "$0")/../improving-inductive-oov-recsys_amd"
mkdir -p lib/ab
make -C csrc -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 \
  -I../include -Wall -Wno-unused-function $2 -c csrc/lsh64.hip -o lib/ab/$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o lib/ab/$1.so lib/obj/api.cpp.o lib/obj/lsh.hip.o lib/ab/$1.o \
  lib/obj/hash.hip.o lib/obj/gather.hip.o lib/obj/score.hip.o
rm lib/ab/$1.o
echo built lib/ab/$1.so
