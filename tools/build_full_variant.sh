#!/bin/bash
# Builds whole-library variants (gemm_bf16.hip compiled with extra -D knobs, every other object as in the product)
# for step-level A/B runs through OASR_HIP_LIBRARY.  Usage: [SRC=dwconv] bash tools/build_full_variant.sh tag1 "-DFOO" tag2 "-DBAR" ...
# (SRC names the one source file the knobs apply to; default gemm_bf16)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/omnilingual-asr_amd/csrc
O=$R/omnilingual-asr_amd/lib
B=$R/omnilingual-asr_amd/build
make -s -C "$C" -j8
SRC=${SRC:-gemm_bf16}
mkdir -p /tmp/fullvar
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$C"
OTHERS=$(ls $B/*.o | grep -v "/$SRC.o")
while [ $# -gt 1 ]; do
    /opt/rocm/bin/hipcc $FL $2 -x hip -c $C/$SRC.hip -o /tmp/fullvar/${SRC}_$1.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $O/abl_full_$1.so /tmp/fullvar/${SRC}_$1.o $OTHERS
    echo built $O/abl_full_$1.so
    shift 2
done
