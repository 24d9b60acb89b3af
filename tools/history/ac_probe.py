import time, torch, os, sys
sys.path.insert(0, os.getcwd())
from oracle import mednext_oracle as MO
torch.set_num_threads(min(64, os.cpu_count()))
print("cpus", os.cpu_count())
st = MO.init_state(seed=0)
kw = dict(n_channels=32, exp_r=2, kernel_size=3, block_counts=[2]*9)
x = torch.rand(1,1,112,112,112)
with torch.no_grad():
    t0=time.time(); y32 = MO.forward(st, x, **kw); print("fp32", time.time()-t0)
    t0=time.time(); y32 = MO.forward(st, x, **kw); print("fp32", time.time()-t0)
    t0=time.time()
    with torch.autocast("cpu", torch.bfloat16):
        y16 = MO.forward(st, x, **kw)
    print("autocast", time.time()-t0, y16.dtype)
    d = (torch.sigmoid(y16.float())-torch.sigmoid(y32)).abs()
    print("autocast vs fp32 max dP", float(d.max()), "mean", float(d.mean()))
