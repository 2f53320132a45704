#!/bin/bash
# Ablation / experiment builds of libpfn_hip.so: [SRC=gemm.hip] tools/build_variants.sh name "<extra hipcc flags for $SRC>" [name flags ...]
# (SRC defaults to attention.hip) -> transformerscandobayesianinference_amd/_variants/libpfn_<name>.so (travels to the GPU box;
# select with PFN_LIB=<path> in the tools/ scripts, or by name in tools/bench_kv_variants.py).  Every build is ~3 MB of snapshot per gpurun call:
# rm -rf transformerscandobayesianinference_amd/_variants when the experiment is over.
set -e
cd "$(dirname "$0")/../transformerscandobayesianinference_amd/csrc"
mkdir -p ../_variants
bash build.sh > /dev/null
SRC=${SRC:-attention.hip}
STEM=${SRC%.hip}
EXTRA=""
[ "$SRC" = attention.hip ] && EXTRA="-fno-slp-vectorize"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  objs=""
  for o in pfn_api gemm attention rowwise bar optim gp_prior mlp_prior; do
    if [ $o = $STEM ]; then objs="$objs ../_variants/${STEM}_$name.o"; else objs="$objs ../_build/$o.o"; fi
  done
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $EXTRA $flags -c $SRC -o ../_variants/${STEM}_$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../_variants/libpfn_$name.so $objs &&
    rm ../_variants/${STEM}_$name.o && echo built $name ) &
done
wait
