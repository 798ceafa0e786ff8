"""Device-side trace of one ego-net through the direct solver (a libgcc_amd.so built with -DGCC_POSEMB_DEVDEBUG prints the
Gram-Schmidt sweeps).  python tests/tools/posemb_item_debug.py tests/golden/posemb_item_s4_v1_b126.npz <seed>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcc_amd.posemb import DevicePosEmb
from gcc_amd.sampler import BatchedCSR
from oracle import posemb as P

z = np.load(sys.argv[1])
seed = int(sys.argv[2])
rp, ci = z["row_ptr"], z["col_idx"]
n = len(rp) - 1
i32 = dict(dtype=torch.int32, device="cuda")
q = BatchedCSR(1, torch.tensor([0, n], **i32), torch.tensor([0, len(ci)], **i32), torch.zeros(n, **i32), torch.zeros(n, **i32),
               torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda())
pe = DevicePosEmb(1, n, 32, device="cuda", seed=seed)
evals, raw = torch.zeros(1, 32, device="cuda"), torch.zeros(n, 32, device="cuda")
pe(q, evals=evals, raw=raw)
torch.cuda.synchronize()
print("status", pe.status.cpu().tolist())
U = raw.cpu().numpy().astype(np.float64)
G = U.T @ U - np.eye(32)
print("orth max", np.abs(G).max(), "at", np.unravel_index(np.abs(G).argmax(), G.shape))
M = P.normalized_adjacency(rp, ci).toarray()
ev = evals.cpu().numpy()[0]
print("evals", np.round(ev, 6).tolist())
print("resid", np.abs(M @ U - U * ev).max(axis=0).round(7).tolist())
