#!/bin/bash
# SQ / GRBM counter passes over the kernels that dominate the step: the three GEMM shapes of a block
# and the attention kernel.  One rocprofv3 run per pass and target, counters only (no tracing domains
# besides the kernel trace).  Usage on the GPU box: bash tools/pmc_kernels.sh <tag> [target ...]
# (targets: ffn_up ffn_down dxd_res attn64 dwconv; default the first four)
#   -> gpurun_out/<tag>_pmc/<target>_p<i>/...  and a table on stdout (tools/pmc_table.py)
set -e
TAG=${1:-r02}
shift || true
TARGETS=${*:-ffn_up ffn_down dxd_res attn64}
want() { case " $TARGETS " in *" $1 "*) return 0;; esac; return 1; }
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${TAG}_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"
 "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
)
run() {  # name, program args...
    local name=$1; shift
    local i=0
    for C in "${PASSES[@]}"; do
        i=$((i+1))
        timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${name}_p$i" -- "$@" > "$OUT/${name}_p$i.log" 2>&1 \
            || { echo "pass $i of $name failed"; tail -3 "$OUT/${name}_p$i.log"; exit 1; }
    done
}
# layout flags as in the encoder: weights packed everywhere; up-projection output and down-projection input packed too
want ffn_up && run ffn_up   python3 "$R/tools/gemm_one.py" 48128 4096 1024 1 5
want ffn_down && run ffn_down python3 "$R/tools/gemm_one.py" 48128 1024 4096 3 3
want dxd_res && run dxd_res  python3 "$R/tools/gemm_one.py" 48128 1024 1024 3 1
want attn64 && run attn64   python3 "$R/tools/attn_one.py" 3
want dwconv && run dwconv python3 "$R/tools/dwconv_one.py"
python3 "$R/tools/pmc_table.py" "$OUT"
