import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"sam-decoding_amd")]
os.environ["SAMD_HEAD_GRAPH"]="0"; os.environ["SAMD_EAGLE_GRAPH"]="0"
import torch, samd_hip
from samd_hip.llama import LlamaRunner
import bench
from samd.tree_model.eagle2 import Eagle2Head
from samd.tree_model.device_head import DeviceHead
mcfg=dict(bench.LLAMA3_8B); mcfg["num_hidden_layers"]=1
dtype=torch.bfloat16
runner=LlamaRunner.random_init(mcfg, 8192, dtype, seed=0)
tree_cfg=dict(hidden_size=4096,intermediate_size=14336,num_attention_heads=32,num_key_value_heads=8,vocab_size=128256,rms_norm_eps=1e-5,rope_theta=500000.0,bias=True)
head=Eagle2Head(tree_cfg,dtype=dtype,device="cuda"); head.random_init(seed=3,std=0.02)
dh=DeviceHead(head,runner)
g=torch.Generator(device="cuda").manual_seed(0)
T=3
hs=torch.randn((T,4096),generator=g,device="cuda").to(dtype); ids=torch.randint(3,128256,(T+1,),generator=g,device="cuda")
# canaries: guard tensors allocated around everything, checked after every library call
guards=[torch.full((1<<16,), 0x5A5A5A5A, dtype=torch.int32, device="cuda") for _ in range(8)]
ids_ref=ids.cpu()
def check(tag):
    torch.cuda.synchronize()
    bad=[i for i,gd in enumerate(guards) if not bool((gd==0x5A5A5A5A).all())]
    if bad or not torch.equal(ids.cpu(), ids_ref):
        print("CORRUPTION after", tag, "guards", bad, "ids", ids.tolist(), flush=True); sys.exit(1)
L=samd_hip.lib()
import ctypes
names=["samd_rope_rows","samd_attention_block","samd_gemm_skinny","samd_gemm_skinny_silu","samd_rmsnorm","samd_argmax_rows","samd_rope_kv_write_cs"]
class Wrap:
    def __init__(self,f,name): self.f,self.name=f,name
    def __call__(self,*a):
        r=self.f(*a); check(self.name); return r
for nm in names:
    setattr(L, nm, Wrap(getattr(L,nm), nm))
dh._lib=L
for it in range(40):
    if it%20==0: dh.reset()
    dh.eagle2_draft(head,hs,ids); check(f"draft {it}")
    print("draft",it,"ok L",int(dh.L.item()),flush=True)
