import sys; sys.path.insert(0,'/root/repo/omnilingual-asr_amd'); sys.path.insert(0,'/root/repo')
import torch
from omnilingual_asr import _hip
lib=_hip.lib()
for (batch,t_pad,heads,lens) in [(2,128,8,[124,99]),(3,1504,2,[1499,700,64])]:
    d=heads*64; g=torch.Generator().manual_seed(1)
    rows=(batch*t_pad+255)//256*256
    qkv=torch.randn((rows,3*d),generator=g).to(torch.bfloat16).cuda()
    emb=torch.zeros((96,64)); emb[:73]=torch.randn((73,64),generator=g); emb=emb.to(torch.bfloat16).cuda()
    ln=torch.tensor(lens,dtype=torch.int32).cuda()
    outs=[]
    for i in range(6):
        ctx=torch.zeros((rows,d),dtype=torch.bfloat16,device='cuda')
        rc=lib.oasr_attention_relkey(qkv.data_ptr(),emb.data_ptr(),ln.data_ptr(),ctx.data_ptr(),batch,t_pad,heads,64,64,8,torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); outs.append(ctx)
    for i in range(1,6):
        diff=(outs[i].float()-outs[0].float()).abs()
        print(batch,t_pad,heads,'run',i,'equal',torch.equal(outs[i],outs[0]),'maxdiff',diff.max().item(),'nbad',int((diff>0).sum()))
