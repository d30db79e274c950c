#!/bin/bash
# Tuning: build variants/libpsvae_<name>.so with extra -D flags on gemm_f32.hip (or SRC=gemm_bf16 ...; other objects from the normal build).
# usage: tools/build_variant.sh name -DPSVAE_D_BIG=4 ...      then run with PSVAE_LIB=variants/libpsvae_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
SRC=${SRC:-gemm_f32}
mkdir -p variants
python -m pseudo_speaker_vae_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude "$@" -c pseudo_speaker_vae_amd/csrc/$SRC.hip -o variants/${SRC}_$name.o
objs=$(ls pseudo_speaker_vae_amd/build/*.o | grep -v /$SRC.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libpsvae_$name.so variants/${SRC}_$name.o $objs
echo variants/libpsvae_$name.so
