#!/bin/bash
# Dev probes (GPU-box micro-benchmarks with s_memtime phase stamps): built into tools/_build/ (git-ignored; travels with gpurun).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
for p in gemm_probe gemm256_probe gemm8p_probe bufdma_probe attn_probe ff_probe l1_probe ffs_probe; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I miles-credit_amd/csrc tools/$p.hip -o tools/_build/$p
  echo "built tools/_build/$p"
done
