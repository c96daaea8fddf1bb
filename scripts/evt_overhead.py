import sys, time
sys.path.insert(0, '/root/repo')
import torch
from cybervision_amd import correlation, synth
W=4096
img1,img2,_=synth.make_pair(W,W); steps=synth.optimal_scale_steps(W,W)
p1,p2=synth.box_pyramid(img1,steps),synth.box_pyramid(img2,steps)
d1=[torch.from_numpy(p).cuda() for p in p1]; d2=[torch.from_numpy(p).cuda() for p in p2]
dev=correlation.create_gpu_context(stream=torch.cuda.current_stream().cuda_stream)
pc=correlation.PointCorrelations(dev,(W,W),(W,W),synth.F_HORIZONTAL)
out_xy=torch.empty((W,W,2),dtype=torch.int32,device='cuda'); out_corr=torch.empty((W,W),dtype=torch.float32,device='cuda')
def step():
    pc.first_pass=True
    for i in range(steps+1):
        k=steps-i; pc.correlate_images(d1[k],d2[k],1.0/float(1<<k))
    pc.complete(out_xy=out_xy,out_corr=out_corr)
for prof in (False, True, False, True):
    pc.set_profiling(prof, False)
    step(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/10
    if prof: pc.get_kernel_times()
    print('profiling', prof, 'ms/step', round(dt*1e3,3))
