import numpy as np, ctypes as C, sys
sys.path.insert(0,'/root/repo')
from oracle import oracle as O
from tfrec_amd import synth
L=O.lib()
# coefficients as in capi.hip
coefs={
 'tfa2':(float.fromhex('0x1.27f98b1037a14p-8'), float.fromhex('0x1.cd1527f4a26e2p+0'), -float.fromhex('0x1.a36a1c41c6995p-1'), 356),
 'tfa3':(float.fromhex('0x1.7ed02b18a270dp-10'), float.fromhex('0x1.e397ac010fc89p+0'), -float.fromhex('0x1.ca2cf85850d62p-1'), 640),
 'tx22':(float.fromhex('0x1.461fa1a309718p-10'), float.fromhex('0x1.e5d4f47377e30p+0'), -float.fromhex('0x1.ce36282a35d90p-1'), 694),
 'whb':(float.fromhex('0x1.14a67102a1ffdp-7'), float.fromhex('0x1.b949652fa3970p+0'), -float.fromhex('0x1.83dd316f714e0p-1'), 512),
}
src=r'''
#include <stdint.h>
#include <string.h>
typedef struct {double dn1,dn2,yn,yn1;} bq;
static inline double step(bq*f,double b0,double a1,double a2,double dn){
  double b1=b0+b0,b2=b0;
  double y=((b2*f->dn2+a1*f->yn)+(b0*dn+b1*f->dn1))+a2*f->yn1;
  f->yn1=f->yn; f->yn=y; f->dn2=f->dn1; f->dn1=dn; return y;}
/* x[n] inputs; for each start p = seg, 2*seg, ...: samples until zero-start state == true state bitwise; hist in units of 32 samples (cap bins) */
void conv(const double*x,long n,double b0,double a1,double a2,long seg,long*hist,int bins,long*never){
  bq t={0,0,0,0};
  static double ty[1<<22], ty1[1<<22];
  for(long i=0;i<n;i++){ step(&t,b0,a1,a2,x[i]); ty[i]=t.yn; ty1[i]=t.yn1; }
  for(long p=seg;p+seg<=n;p+=seg){
    bq z={0,0,0,0}; long k; int ok=0;
    for(k=0;k<seg;k++){ step(&z,b0,a1,a2,x[p+k]);
      if(k>=1 && memcmp(&z.yn,&ty[p+k],8)==0 && memcmp(&z.yn1,&ty1[p+k],8)==0){ok=1;break;} }
    if(!ok){(*never)++;continue;}
    int b=(int)(k/32); if(b>=bins)b=bins-1; hist[b]++;
  }
}
'''
open('/tmp/tfrec_conv_c.c','w').write(src)
import subprocess
subprocess.check_call(['gcc','-O2','-ffp-contract=off','-shared','-fPIC','-o','/tmp/tfrec_conv_c.so','/tmp/tfrec_conv_c.c'])
cl=C.CDLL('/tmp/tfrec_conv_c.so')
nb=48
bins=130
H={k:np.zeros(bins,dtype=np.int64) for k in coefs}; NV={k:C.c_long(0) for k in coefs}
for s in range(int(sys.argv[1]) if len(sys.argv)>1 else 6):
    iq=synth.gen_stream(1000,s,nb)
    dec=np.empty(2*nb*8192,dtype=np.int16)
    L.orc_decimate(iq.ctypes.data, iq.size//2, 0, dec.ctypes.data)
    I=dec[0::2].astype(np.int64); Q=dec[1::2].astype(np.int64)
    pwr=np.abs(I)+np.abs(Q); trig=np.nonzero(pwr>500)[0]
    pI=np.concatenate(([0],I[:-1])); pQ=np.concatenate(([0],Q[:-1]))
    cr=I*pI+Q*pQ; cj=Q*pI-I*pQ
    fm=np.trunc(np.arctan2(cj.astype(np.float64),cr.astype(np.float64))*(16384.0/np.pi))
    nrzs=np.clip(cr,-1000000000,1000000000).astype(np.float64)
    for name,(b0,a1,a2,W) in coefs.items():
        # in-window flag: g - lastTrigger(g) < W
        last=np.full(len(I),-10**9,dtype=np.int64); last[trig]=trig; last=np.maximum.accumulate(last)
        inw=(np.arange(len(I))-last)<W
        x=np.ascontiguousarray((nrzs if name=='whb' else fm)[inw])
        cl.conv(x.ctypes.data_as(C.c_void_p),C.c_long(len(x)),C.c_double(b0),C.c_double(a1),C.c_double(a2),C.c_long(4096),H[name].ctypes.data_as(C.c_void_p),bins,C.byref(NV[name]))
for name in coefs:
    h=H[name]; tot=h.sum()+NV[name].value
    cum=np.cumsum(h)/max(1,tot)
    print(name,'segments',tot,'never(>=4096)',NV[name].value)
    print('  slots->cum frac:',' '.join('%d:%.4f'%(b+1,cum[b]) for b in (3,7,9,11,13,15,17,19,21,23,25,27,29,31,35,39,47,63,95,127)))
