#!/bin/bash
# Builds stand-alone libraries of the GEMM kernel with different -D knobs for in-process A/B timing
# (tools/gemm_ab2.py).  Usage: bash tools/build_gemm_variants.sh tag1 "-DFOO=1" tag2 "-DFOO=2" ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/omnilingual-asr_amd/csrc
O=$R/omnilingual-asr_amd/lib
mkdir -p "$O" /tmp/gemmvar
rm -f /tmp/gemmvar/*.o
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$C"
while [ $# -gt 1 ]; do
    /opt/rocm/bin/hipcc $FL $2 -x hip -c $C/gemm_bf16.hip -o /tmp/gemmvar/gemm_$1.o &
    shift 2
done
/opt/rocm/bin/hipcc $FL -x hip -c $C/host_api.cpp -o /tmp/gemmvar/host_api.o &
wait
for f in /tmp/gemmvar/gemm_*.o; do
    tag=$(basename $f .o); tag=${tag#gemm_}
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $O/abl_gemm_$tag.so $f /tmp/gemmvar/host_api.o
    echo built $O/abl_gemm_$tag.so
done
