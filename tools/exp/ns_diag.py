import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cmtf_pls_amd.synthetic import synthetic_shard_device
from cmtf_pls_amd import tPLS
I = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
J = int(sys.argv[2]) if len(sys.argv) > 2 else 256
X, Y = synthetic_shard_device((I, J, J), 32, 10, error=0.1, seed=215, device="cuda:0")
rng = np.random.default_rng(215)
A0 = rng.normal(0, 1, size=(I, 10)); C = rng.normal(0, 1, size=(32, 10))
Yc = torch.from_numpy(A0 @ C.T).cuda()
print("Y var", float(Y.var()), "clean var", float(Yc.var()), "resid var", float((Y - Yc).var()), "finite", bool(torch.isfinite(Y).all()))
for r in (0, 1, 65535, 65536, 131072, I - 1):
    print(r, "Y resid rms", float((Y[r] - Yc[r]).pow(2).mean().sqrt()), "X row rms", float(X[r].double().pow(2).mean().sqrt()))
for alg in ("direct", "xcov"):
    m = tPLS(3, dtype="float32", algorithm=alg)
    m.fit(X, Y, max_iter=30)
    print(alg, "n_iter", m.n_iter_, "R2X", m.R2X, "R2Y", m.R2Y, flush=True)
