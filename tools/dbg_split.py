import sys, numpy as np, torch
sys.path.insert(0, '.')
from image_caption_amd import ops
for (M,K,N) in [(2048,32,2048),(2048,64,2048),(1536,512,10000),(4096,64,1024)]:
    rng=np.random.default_rng(0)
    x=rng.standard_normal((M,K)).astype(np.float32); w=(rng.standard_normal((K,N))/np.sqrt(K)).astype(np.float32)
    got=ops.dense_split(torch.from_numpy(x).cuda(), ops.pack_weight_split(torch.from_numpy(w).cuda())).cpu().numpy()
    ref=x.astype(np.float64)@w
    bad=np.abs(got-ref)>1e-4
    print(M,K,N,'bad',bad.sum(), 'rows', np.unique(np.nonzero(bad)[0]//32)[:20], 'cols', np.unique(np.nonzero(bad)[1]//32)[:20], 'nonfinite', (~np.isfinite(got)).sum())
