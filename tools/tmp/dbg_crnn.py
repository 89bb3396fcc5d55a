import sys, torch
sys.path[:0]=['/root/repo','/root/repo/deep-cine-cardiac-mri_amd']
from cine_hip import ops
import torch.nn.functional as Fn
torch.manual_seed(0)
dev=torch.device('cuda')
for (c,h,w) in ((6,24,20),(16,200,200),(16,52,16)):
    wt=torch.randn(c,c,3,3,device=dev)*0.1
    wp=ops.pack_conv3x3(wt)
    xf=torch.randn(1,c,h,w,device=dev); xb=torch.randn(1,c,h,w,device=dev)
    af=torch.randn(1,c,h,w,device=dev); ab=torch.randn(1,c,h,w,device=dev)
    ref_f=torch.relu(Fn.conv2d(xf,wt,padding=1)+af); ref_b=torch.relu(Fn.conv2d(xb,wt,padding=1)+ab)
    yf=torch.empty_like(xf); yb=torch.empty_like(xb); of=torch.full_like(xf,7.); ob=torch.full_like(xb,3.)
    ops.crnn_step2(wp,(xf,af,yf,of,True),(xb,ab,yb,ob,False))
    torch.cuda.synchronize()
    print(c,h,w,'pair: yf',float((yf-ref_f).abs().max()),'yb',float((yb-ref_b).abs().max()),'of(store)',float((of-ref_f).abs().max()),'ob(acc)',float((ob-3-ref_b).abs().max()))
    yf2=torch.empty_like(xf); of2=torch.full_like(xf,7.)
    ops.crnn_step2(wp,(xf,af,yf2,of2,False))
    print('   single: yf',float((yf2-ref_f).abs().max()),'of(acc)',float((of2-7-ref_f).abs().max()))
    y3=torch.empty_like(xf)
    ops.crnn_step(xf,wp,af,y3)
    print('   old entry: y',float((y3-ref_f).abs().max()))
