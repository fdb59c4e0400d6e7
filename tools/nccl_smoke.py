#!/usr/bin/env python3
"""One-rank rehearsal of the collective calls bench.py makes under backend "nccl" (RCCL): process-group
set-up with a device id, barrier, the record all-gather on device tensors, the MAX all-reduce.
Run:  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
          --master-port 29611 tools/nccl_smoke.py"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "omnilingual-asr_amd")); sys.path.insert(0, str(ROOT))
import numpy as np, torch, torch.distributed as dist
from omnilingual_asr import distributed as odist
from omnilingual_asr.models.inference.asr_pipeline import ChunkRecord
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
dist.barrier()
recs = [ChunkRecord(i, i * 480000, (i + 1) * 480000, 1499, np.arange(5, dtype=np.int32) + i,
                    np.arange(5, dtype=np.int32) * 3) for i in range(4)]
g = odist.all_gather_records(odist.pack_records(recs, 6, 16))
back = odist.unpack_records(g)
assert [r.chunk_idx for r in back] == [0, 1, 2, 3] and list(back[3].tokens) == [3, 4, 5, 6, 7]
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
dist.barrier()
dist.destroy_process_group()
print("nccl one-rank rehearsal ok: world", os.environ.get("WORLD_SIZE"), "gathered", tuple(g.shape), g.device)
