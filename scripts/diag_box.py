import sys
sys.path.insert(0,'/root/repo')
import torch
from cybervision_amd import correlation, synth
W=2048; T=float(sys.argv[1])
img1,img2,_=synth.make_pair(W,W,tilt_deg=T); steps=synth.optimal_scale_steps(W,W)
p1,p2=synth.box_pyramid(img1,steps),synth.box_pyramid(img2,steps)
dev=correlation.create_gpu_context(); pc=correlation.PointCorrelations(dev,(W,W),(W,W),synth.f_tilt(T) if T else synth.F_HORIZONTAL)
pc.set_profiling(False,True)
for i in range(steps+1):
    k=steps-i; pc.correlate_images(p1[k],p2[k],1.0/(1<<k))
    c=pc.get_counters()
    v=c['candidates']; print('k',k,'declined',c['whole_corridor_pixels'],'odd',(v>>32)&255,'ncol',(v>>40)&255,'H',(v>>48)&255,'area',(v>>56)&255)
