import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'oracle')); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, fisr_oracle as O
from fisr_amd.splitfmt import *
import test_gpu_parity as T
shape=(1, 8, 32, 16, 0, 6, 0, False)
n,h,w,c0,c1,cout,flags,use_res=shape
rng=np.random.default_rng(hash(shape)%(2**31)+2)
x=rng.standard_normal((n,h,w,c0)).astype(np.float32)
wt=(rng.standard_normal((3,3,c0,cout))*np.sqrt(2.0/(9*c0))).astype(np.float32)
b=rng.standard_normal(cout).astype(np.float32)
got=T.hip_conv(x,wt,b,prec="f16f8",out_f32=True).astype(np.float64)
xh=x.astype(np.float16).astype(np.float64); xl=fp8_e4m3_decode(fp8_e4m3_encode(np.clip((x-xh.astype(np.float32))*2**14,-448,448))).astype(np.float64)*2.0**-14
xh8=fp8_e4m3_decode(fp8_e4m3_encode(np.clip(xh,-448,448))).astype(np.float64)
mx=np.abs(wt).max(); wexp=4-int(np.floor(np.log2(mx)))     # r05: wh8 = fp8(w_h 2^wexp), wl8 = fp8(w_l 2^(wexp+14)) (conv3x3.h)
wh=wt.astype(np.float16).astype(np.float64); wl=fp8_e4m3_decode(fp8_e4m3_encode((wt-wh.astype(np.float32)).astype(np.float64)*2.0**(wexp+14))).astype(np.float64)*2.0**-(wexp+14)
wh8=fp8_e4m3_decode(fp8_e4m3_encode(wh*2.0**wexp)).astype(np.float64)*2.0**-wexp
z=np.zeros(cout)
main=O.conv2d(xh,wh,b); c1_=O.conv2d(xl,wh8,z); c2_=O.conv2d(xh8,wl,z)
d=(got-main).ravel()
A=np.stack([c1_.ravel(),c2_.ravel()],1)
coef,res,_,_=np.linalg.lstsq(A,d,rcond=None)
print('coefficients of (c1,c2) in got-main:', coef, 'residual rms', np.sqrt(((d-A@coef)**2).mean()), 'd rms', np.sqrt((d**2).mean()))
# per-tap decomposition: which taps contribute
for tap in range(9):
    w1=np.zeros_like(wh8); w1[tap//3,tap%3]=wh8[tap//3,tap%3]; w2=np.zeros_like(wl); w2[tap//3,tap%3]=wl[tap//3,tap%3]
    t1=O.conv2d(xl,w1,z).ravel(); t2=O.conv2d(xh8,w2,z).ravel()
    print('tap',tap,'corr with t1 %.3f t2 %.3f'%(np.dot(d,t1)/np.dot(t1,t1), np.dot(d,t2)/np.dot(t2,t2)))
