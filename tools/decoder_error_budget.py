"""Where the bf16-mode mel error of the StyleTTS decoder comes from: the NumPy oracle with bf16 rounding switched on at\nselected points (weights / conv inputs / conv1 outputs feeding a norm / the residual stream), on the reference fixture\ne2e_styletts_v1_T64.  CPU only.  python tools/decoder_error_budget.py"""
import sys, math, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zvx_oracle as O
from zerovox_amd import config as zcfg, weights as zw
def bf(x):
    x=np.ascontiguousarray(x,np.float32); u=x.view(np.uint32); u2=((u+0x7fff+((u>>16)&1))&0xffff0000).astype(np.uint32); return u2.view(np.float32)
g=np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'e2e_styletts_v1_T64.npz'))
cfg=zcfg.medium_modelcfg('styletts'); sd=zw.tts_state_dict(cfg,0)
feat=g['features']; spk=g['spk']; ref=g['mel'].T
# flags: W round weights; A round conv inputs; S round residual stream (stored tensors); N stats from rounded tensor; C round conv outputs feeding norms
def run(W,A,S,C,NS):
    sdq=dict(sd)
    def conv(x,p,pad):
        w=O.fold_wn(sd,p) if hasattr(O,'fold_wn') else None
        xin=bf(x) if A else x
        if W:
            key=p
            sdl={k:v for k,v in sd.items() if k.startswith(p+'.')}
            wv=O.fold_wn(sd,p); 
            b=sd.get(p+'.bias')
            y=O.conv1d(xin, bf(wv), b, padding=pad)
        else:
            y=O._wn_conv(xin,sd,p,pad)
        return y
    def inorm(x, xs=None):
        src = xs if xs is not None else x
        mu=src.mean(axis=1,keepdims=True); var=((src-mu)**2).mean(axis=1,keepdims=True)
        return (x-mu)/np.sqrt(var+np.float32(1e-5))
    def store(x): return bf(x) if S else x
    def cstore(x): return bf(x) if C else x
    def normed(x, gam, bet):   # x is the stored tensor (maybe rounded); stats from x (rounded) or exact NS
        y=inorm(x)
        return y*gam[:,None]+bet[:,None]
    p='_mel_decoder'
    e=store(feat.T.astype(np.float32))
    def resblk(x,pp):
        has_sc=(pp+'.conv1x1.weight_v') in sd or (pp+'.conv1x1.weight') in sd
        sc=conv(x,pp+'.conv1x1',0) if has_sc else x
        r=normed(x,sd[pp+'.norm1.weight'],sd[pp+'.norm1.bias']); r=cstore(conv(O.leaky_relu(r,0.2),pp+'.conv1',1))
        r=normed(r,sd[pp+'.norm2.weight'],sd[pp+'.norm2.bias']); r=conv(O.leaky_relu(r,0.2),pp+'.conv2',1)
        return store((sc+r)/np.float32(math.sqrt(2)))
    def adain(x,pp):
        h=O.linear(spk.reshape(-1),sd[pp+'.fc.weight'],sd[pp+'.fc.bias']); Cn=x.shape[0]
        return (1+h[:Cn,None])*inorm(x)+h[Cn:,None]
    def ablk(x,pp):
        r=O.leaky_relu(adain(x,pp+'.norm1'),0.2); r=cstore(conv(r,pp+'.conv1',1))
        r=O.leaky_relu(adain(r,pp+'.norm2'),0.2); r=conv(r,pp+'.conv2',1)
        has_sc=(pp+'.conv1x1.weight_v') in sd or (pp+'.conv1x1.weight') in sd
        sc=conv(x,pp+'.conv1x1',0) if has_sc else x
        return store((r+sc)/np.float32(math.sqrt(2)))
    x=resblk(resblk(e,p+'.encode.0'),p+'.encode.1')
    asr=store(O.instance_norm1d(cstore(conv(e,p+'.asr_res.0',0)),sd[p+'.asr_res.1.weight'],sd[p+'.asr_res.1.bias']))
    res=True
    for i in range(5):
        if res: x=np.concatenate([x,asr],axis=0)
        x=ablk(x,f'{p}.decode.{i}')
        if i==2: res=False
    return conv(x,p+'.to_out.0',0).T
def err(a): d=a-ref; return f"max {np.abs(d).max():.3e} rms {np.sqrt((d**2).mean()):.3e}"
print('ref rms', np.sqrt((ref**2).mean()))
print('exact            ', err(run(0,0,0,0,0)))
print('W only           ', err(run(1,0,0,0,0)))
print('W + conv inputs  ', err(run(1,1,0,0,0)))
print('W+A + c1 outputs ', err(run(1,1,0,1,0)))
print('W+A + stream     ', err(run(1,1,1,0,0)))
print('all (current)    ', err(run(1,1,1,1,0)))
