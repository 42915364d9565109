import json, os, subprocess, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT)
import torch
from dump1090_amd import Demodulator
gib=8
path="/dev/shm/modes_e2e.bin"
d=Demodulator(fix=False)
with open(path,"wb") as f:
    for k in range(gib):
        iq=torch.empty(1<<30,dtype=torch.uint8,device="cuda:0")
        d.synth_noise(iq,k<<30,seed=77,sigma_q16=941)
        if k==gib-1: d.fill(iq[-480:],127)
        iq.cpu().numpy().tofile(f)
d.close(); del iq; torch.cuda.empty_cache()
exe=os.path.join(ROOT,"dump1090_amd","bin","dump1090_amd")
def run(env_extra, extra=[]):
    t0=time.perf_counter()
    p=subprocess.run([exe,"--ifile",path,"--raw","--no-fix","--timing"]+extra,stdout=subprocess.PIPE,stderr=subprocess.PIPE,env=dict(os.environ,**env_extra),check=True)
    dt=time.perf_counter()-t0
    tim=json.loads([l for l in p.stderr.decode().splitlines() if l.startswith("{")][-1])
    return round(dt,3), tim["total_s"], tim["init_s"], tim["stream_s"]
run({})
for name,env,extra in (("unmap-as-you-go",{},[]),("keep-mapping",{"MODES_HOST_KEEP_MAPPING":"1"},[]),("unmap-as-you-go",{},[]),("keep-mapping",{"MODES_HOST_KEEP_MAPPING":"1"},[]),("unmap-as-you-go",{},[]),("keep-mapping",{"MODES_HOST_KEEP_MAPPING":"1"},[]),("no-mmap",{},["--no-mmap"]),("no-mmap",{},["--no-mmap"])):
    print(name, run(env,extra), flush=True)
os.remove(path)
