#!/bin/bash
# Builds libpfn_hip.so for gfx950 (MI355X).  Usage: csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../libpfn_hip.so
SRCS="pfn_api.hip gemm.hip gemm_tn.hip attention.hip rowwise.hip bar.hip optim.hip gp_prior.hip mlp_prior.hip"
mkdir -p ../_build
pids=()
for f in $SRCS; do
  o=../_build/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ pfn_device.h -nt "$o" ] || [ pfn_kernels.h -nt "$o" ] || [ ../../include/pfn_hip.h -nt "$o" ]; then
    extra=""
    # attention.hip: no SLP packing -- v_pk_add/mul_f32 beside MFMAs issue slower than the scalar ops they replace
    [ "$f" = attention.hip ] && extra="-fno-slp-vectorize"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $extra "$@" -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT ../_build/pfn_api.o ../_build/gemm.o ../_build/gemm_tn.o ../_build/attention.o ../_build/rowwise.o ../_build/bar.o ../_build/optim.o ../_build/gp_prior.o ../_build/mlp_prior.o
echo "built $(realpath $OUT)"
