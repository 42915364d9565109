import sys,os,time
sys.path.insert(0,'/root/repo')
import torch
from dump1090_amd import Demodulator
d=[Demodulator(fix=False) for _ in range(2)]
iq=torch.empty(1<<30,dtype=torch.uint8,device='cuda:0')
d[0].synth_noise(iq,0,seed=20260922,sigma_q16=941)
torch.cuda.synchronize()
out=[]
for rep in range(3):
    ms=[]
    for i in range(60):
        x=d[i%2]
        x.detect(iq)
        _,_,info=x.fetch()
        ms.append(round(info['scan_ms']*1000))
    out.append(ms)
    time.sleep(1.0)
for ms in out: print(ms)
