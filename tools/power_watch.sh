#!/bin/bash
# Samples sclk (and the power fields the user may read) twice a second while the FFN up-projection GEMM runs back
# to back for 10 s: the clock the chip holds under the dominant kernel.  Usage on the GPU box: bash tools/power_watch.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ';' ; echo; sleep 0.5; done ) > gpurun_out/power_watch.txt &
W=$!
timeout -k 10 60 python3 - <<'PY'
import sys, time
sys.path.insert(0, "omnilingual-asr_amd")
import torch
from omnilingual_asr import _hip
lib = _hip.lib()
m, n, k = 48128, 4096, 1024
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
a = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn((n, k), generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
bias = torch.randn((n,), generator=g).to(dev)
c = torch.zeros((m, n), dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
time.sleep(3)
t0 = time.time()
while time.time() - t0 < 10:
    for _ in range(200):
        lib.oasr_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), None, c.data_ptr(), m, n, k, 1.0, 1, st)
    torch.cuda.synchronize()
time.sleep(3)
PY
wait $W
cat gpurun_out/power_watch.txt | cut -c1-260 | awk 'NR%3==1'
