import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"sam-decoding_amd")]
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
seq=[int(x) for x in sys.argv[1].split(",")]
os.environ["SAMD_EAGLE_GRAPH"]=sys.argv[2] if len(sys.argv)>2 else "1"
g=torch.Generator(device="cuda").manual_seed(0)
for T in seq:
    if T==0:
        dh.reset(); print("reset",flush=True); continue
    hs=torch.randn((T,4096),generator=g,device="cuda").to(dtype); ids=torch.randint(3,128256,(T+1,),generator=g,device="cuda")
    toks,par=dh.eagle2_draft(head,hs,ids)
    torch.cuda.synchronize(); print(T,"ok",toks[1:4].tolist(),flush=True)
