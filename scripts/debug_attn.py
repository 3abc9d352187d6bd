import math, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sam-decoding_amd"))
import numpy as np, torch
import samd_hip
def dev(a): return torch.as_tensor(np.asarray(a), dtype=torch.int32).cuda()
def run(q,k,v,L,n,mask_rows,H,Hkv,dtype):
    n_pad=64; D=128; max_len=k.shape[1]
    mask = torch.tensor(np.array(mask_rows + [0]*(64-n), dtype=np.uint64).view(np.int64), device="cuda")
    out = torch.full((n_pad,H,D), 7.0, device="cuda").to(dtype)
    wsb = samd_hip.lib().samd_tree_attention_workspace(n_pad,H,D)
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    samd_hip.check(samd_hip.lib().samd_tree_attention(samd_hip._ptr(q), samd_hip._ptr(k), samd_hip._ptr(v), samd_hip._ptr(out),
        samd_hip.torch_dtype_code(dtype), n_pad,H,Hkv,D,max_len, samd_hip._ptr(mask), samd_hip._ptr(dev([L])), samd_hip._ptr(dev([n])),
        1.0/math.sqrt(D), samd_hip._ptr(ws), wsb, samd_hip.current_stream()))
    torch.cuda.synchronize()
    return out, ws.view(torch.float32)
H=Hkv=2; D=128; max_len=256; dtype=torch.float16
# case A: L=0,n=1 -> out[0]=v[:,0]
q=torch.randn((64,H,D),device="cuda").to(dtype); k=torch.randn((Hkv,max_len,D),device="cuda").to(dtype)
v=torch.arange(Hkv*max_len*D,device="cuda").reshape(Hkv,max_len,D).float()
v=((v%97)/97.0).to(dtype)
out,ws=run(q,k,v,0,1,[1],H,Hkv,dtype)
print("A got", out[0,0,:8].tolist()); print("A want", v[0,0,:8].tolist())
print("ws split0 row0 h0:", ws[:8].tolist(), ws[128:130].tolist())
# case B: q=0 -> mean of v over L+n keys (chain mask, row n-1 sees all)
L,n=100,4
q=torch.zeros((64,H,D),device="cuda").to(dtype)
out,ws=run(q,k,v,L,n,[1,3,7,15],H,Hkv,dtype)
for i in range(n):
    want=v[0,:L+i+1].float().mean(0)
    print("B row",i,"got",out[i,0,:4].tolist(),"want",want[:4].tolist(), "maxerr", (out[i,0].float()-want).abs().max().item())
# case C: v = onehot by key index -> out gives softmax probs
L,n=60,4
v2=torch.zeros((Hkv,max_len,D),device="cuda").to(dtype)
for kk in range(64): v2[:,kk,kk]=1.0
q=torch.randn((64,H,D),device="cuda").to(dtype)
out,ws=run(q,k,v2,L,n,[1,3,7,15],H,Hkv,dtype)
for i in range(n):
    s=(q[i,0].float()@k[0,:L+i+1].float().T)/math.sqrt(D); p=torch.softmax(s,-1)
    print("C row",i,"maxerr",(out[i,0,:L+i+1].float()-p).abs().max().item(), "got",out[i,0,:4].tolist(),"want",p[:4].tolist())
