import sys, json
sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT","/root/repo"))
import torch, bench
import labrador_snark_amd as lab
q, n, r = bench.Q_REF33, 1024, 32
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
k = lab.RuntimeConstants(q, n, r, 1024, 1024, 1024)
ctx = lab.Context(q, 0, max(65536, n * r + r * r)); ctx.use_torch_stream()
dSs, sts, dtapes, crs = bench.build_workload(lab, ctx, k, q, dev, 64, 50, 1000)
step = lambda: lab.prove_core_batch(ctx, k, dSs, sts, crs, dtapes, device_outputs=True)
step(); torch.cuda.synchronize()
ticks = torch.zeros(2 * 16384 + 64, dtype=torch.int64, device=dev)
ctx.gemm_diag(4, ticks)
step(); torch.cuda.synchronize()
ctx.gemm_diag(0, None)
t = ticks.cpu().numpy()[:2*16384].reshape(-1, 2); t = t[t[:, 1] > 0]
import numpy as np
print(len(t), "blocks; loop shader ticks median", np.median(t[:,0]), "const ticks median", np.median(t[:,1]), "-> us", np.median(t[:,1])/100, "clock GHz", (t[:,0]/t[:,1]).mean()*0.1)
print("p10/p90 us", np.percentile(t[:,1],10)/100, np.percentile(t[:,1],90)/100)
