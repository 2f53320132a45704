#!/bin/bash
# Ablation / experiment builds of libpfn_hip.so: tools/build_variants.sh name "<extra hipcc flags for attention.hip>" [name flags ...]
# -> transformerscandobayesianinference_amd/_variants/libpfn_<name>.so (travels to the GPU box; select with PFN_LIB=<path>)
set -e
cd "$(dirname "$0")/../transformerscandobayesianinference_amd/csrc"
mkdir -p ../_variants
bash build.sh > /dev/null
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -fno-slp-vectorize $flags -c attention.hip -o ../_variants/attention_$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../_variants/libpfn_$name.so ../_build/pfn_api.o ../_build/gemm.o ../_variants/attention_$name.o ../_build/rowwise.o ../_build/bar.o ../_build/optim.o ../_build/gp_prior.o ../_build/mlp_prior.o &&
    rm ../_variants/attention_$name.o && echo built $name ) &
done
wait
