import os, sys, torch
sys.path.insert(0, "/root/repo")
from cm3p_amd import kernels as K
DEV="cuda"
torch.manual_seed(0)
B, S, nh = 4, 333, 2
lens = [333, 200, 97, 64]
window=64
qkv = (torch.randn(B, S, 3, nh, 64, device=DEV) * 0.7).bfloat16()
do = torch.randn(B * S, nh * 64, device=DEV).bfloat16()
mask = torch.zeros(B, S, dtype=torch.uint8, device=DEV)
for b, n in enumerate(lens):
    mask[b, :n] = 1
do = do * mask.reshape(B * S, 1).to(do.dtype)
inv_freq = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, device=DEV, dtype=torch.float32) / 64))
cos, sin = K.rope_table(torch.arange(S, device=DEV), inv_freq)
out, lse = K.attn_fwd(qkv, mask, B, S, nh, window, 0.125)
idx = torch.nonzero(mask.flatten()).flatten()
cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
qkv_p = qkv.reshape(B * S, 3, nh, 64)[idx].contiguous()
do_p = do[idx].contiguous()
pos = (idx % S).contiguous()
cos_p, sin_p = K.rope_table(pos, inv_freq)
out_p, lse_p = K.attn_fwd_varlen(qkv_p, cu, B, max(lens), nh, window, 0.125)
r={}
for rep in range(2):
  for mode in ("0","1"):
    os.environ["CM3P_ATTN_BAND_MERGED"]=mode
    r[("pad",mode,rep)] = K.attn_bwd(qkv, out, do, lse, mask, B, S, nh, window, 0.125, (cos, sin), False).reshape(B*S,3,nh,64)[idx].clone()
    r[("pack",mode,rep)] = K.attn_bwd_varlen(qkv_p, out_p, do_p, lse_p, cu, B, max(lens), nh, window, 0.125, (cos_p, sin_p)).clone()
torch.cuda.synchronize()
def cmp(a,b):
    d=(r[a].float()-r[b].float()).abs()
    bad=(d>0).nonzero()
    print(a,b,"equal" if bad.numel()==0 else f"max {d.max().item():.3e} n {bad.shape[0]} first {bad[:4].tolist()} rows {sorted(set(bad[:,0].tolist()))[:10]} thirds {sorted(set(bad[:,1].tolist()))}")
cmp(("pad","0",0),("pad","1",0)); cmp(("pack","0",0),("pack","1",0)); cmp(("pad","0",0),("pack","0",0)); cmp(("pad","1",0),("pack","1",0))
cmp(("pad","1",0),("pad","1",1)); cmp(("pack","1",0),("pack","1",1))
