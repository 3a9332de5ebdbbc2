import torch, time, sys
sys.path.insert(0,'.')
from sympa_amd import data, ops
dev=torch.device('cuda:0')
table=data.trained_like_table(5041,4).to(dev)
pairs=data.sample_pairs(5041,65536).to(dev)
go=torch.rand(65536,device=dev,dtype=torch.float64)
z1=table[pairs[:,0]].contiguous(); z2=table[pairs[:,1]].contiguous()
grad=torch.zeros_like(table)
def timed(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
print('dense backward (no atomics): %.1f us'%timed(lambda: ops.siegel_dist_backward(z1,z2,go)))
print('scatter backward (atomics):  %.1f us'%timed(lambda: ops.model_backward(table,pairs,go,grad_table=grad)))
print('forward pre-gathered:        %.1f us'%timed(lambda: ops.siegel_dist_forward(z1,z2)))
