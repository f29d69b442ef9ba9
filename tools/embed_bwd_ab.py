#!/usr/bin/env python3
"""Embedding + LayerNorm backward at the C2 shape (131072 tokens, 3167 x 768 table): the atomic scatter-add against the id-ordered
kernels (CM3P_EMBED_BWD=atomic | sorted), uniform ids and 40 % of the tokens on one id.  Development aid."""
import os, sys, torch
sys.path.insert(0, '/root/repo')
from cm3p_amd import kernels as K, _lib
T, V, H = 131072, 3167, 768
g = torch.Generator().manual_seed(0)
ids = torch.randint(0, V, (T,), generator=g).cuda()
table = torch.randn(V, H, generator=g).cuda(); w = torch.ones(H).cuda(); dy = torch.randn(T, H, generator=g).cuda()
_, _, mean, rstd = K.embed_ln_fwd(ids, table, w, 1e-5, None, None, want_bf16=False)
for impl in ("atomic", "sorted", "atomic", "sorted"):
    os.environ["CM3P_EMBED_BWD"] = impl
    K.embed_ln_bwd(dy, ids, table, w, mean, rstd, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.embed_ln_bwd(dy, ids, table, w, mean, rstd, 0)
    e1.record(); torch.cuda.synchronize()
    print(impl, e0.elapsed_time(e1) / 10, "ms (uniform ids)")
ids2 = ids.clone(); ids2[torch.rand(T, device='cuda') < 0.4] = 5
for impl in ("atomic", "sorted"):
    os.environ["CM3P_EMBED_BWD"] = impl
    K.embed_ln_bwd(dy, ids2, table, w, mean, rstd, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.embed_ln_bwd(dy, ids2, table, w, mean, rstd, 0)
    e1.record(); torch.cuda.synchronize()
    print(impl, e0.elapsed_time(e1) / 10, "ms (40 % of the tokens one id)")
