import sys; sys.path.insert(0,'.')
import torch, time, labrador_snark_amd as lab
for q in ((1<<32)+15, 8191):
    ctx=lab.Context(q,0); ctx.use_torch_stream()
    n=1<<20
    v=ctx.vec(n)
    k=lab.RuntimeConstants((1<<32)+15,2,2,4,3,3)
    crs=lab.CRS(ctx,k,bytes(range(32)))
    crs._expand_into(v,0,n,0); torch.cuda.synchronize()
    t0=time.perf_counter(); crs._expand_into(v,0,n,0); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("q=%d: CRS expand %d polys (%d limbs, incl. reduce+NTT) in %.2f ms -> %.3e coeff/s"%(q,n,ctx.nlimbs,dt*1e3,n*64/dt))
    v.close(); ctx.close()
