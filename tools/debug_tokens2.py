import sys, os, ctypes as C, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from dir_amd import _capi, engine, synth
from oracle import nnops as N, tokens as OT
from oracle.golden_inputs import bone_uv
from test_gpu_tokens import pgcn_shapes, stage_sd
SEED=1234
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def err(a,b): return float(np.abs(np.asarray(a,np.float64)-np.asarray(b,np.float64)).max())
sdn = synth.synth_state_dict(pgcn_shapes(), SEED)
sd = {('gcn.'+k): dev(v) for k,v in sdn.items()}
keep=[]; layers = engine.pack_pgcn(sd,'gcn',keep)
P = N.Params(sdn)
B=3
x = synth.synth_input('dbg.x',(B,21,128),SEED)
for nl in (1,2,3,4):
    out = torch.empty(B,21,128,device='cuda'); scratch = torch.zeros(2,B,21,256,device='cuda')
    _capi.check(_capi.lib().dir_pgcn_stack_forward(layers,nl,_capi.ptr(dev(x)),None,_capi.ptr(out),21*128,_capi.ptr(scratch),B,_capi.stream_ptr()),'p')
    ref = OT.pgcn_stack(x, P, num_layers=nl)
    print('pgcn %d layers: err %.3e scale %.3f' % (nl, err(out.cpu().numpy(), ref), np.abs(ref).max()))
    if nl == 2:
        x1 = OT.graphconv_layer(x, P.sub('gconv_layers.0'))
        W = sdn['gconv_layers.1.gconv.W']
        h0 = np.einsum('bjc,jcd->bjd',x1,W[0]); h1=np.einsum('bjc,jcd->bjd',x1,W[1])
        s = scratch[1].cpu().numpy()
        e = np.abs(s[...,:128]-h0)
        print('   layer1 h0 err', e.max(), 'per node', e.max(axis=(0,2)).round(4), 'h1 err', err(s[...,128:],h1))
# regress emb
sdn2, sd2 = stage_sd(16)
keep=[]; st = engine.StageOp(sd2,'st',16,1,torch.float32,0,keep)
P2 = N.Params(sdn2)
tok = synth.synth_input('rg.tok',(B,42,64),SEED)
pl, pr, off = synth.synth_input('rg.pl',(B,64),SEED), synth.synth_input('rg.pr',(B,64),SEED), synth.synth_input('gt.off',(B,3),SEED)
o = [torch.empty(B,64,device='cuda'),torch.empty(B,64,device='cuda'),torch.empty(B,3,device='cuda'),torch.empty(B,42,64,device='cuda')]
_capi.check(_capi.lib().dir_regress_forward(C.byref(st.reg),_capi.ptr(dev(tok)),_capi.ptr(dev(pl)),_capi.ptr(dev(pr)),_capi.ptr(dev(off)),_capi.ptr(o[0]),_capi.ptr(o[1]),_capi.ptr(o[2]),_capi.ptr(o[3]),B,_capi.stream_ptr()),'r')
emb = OT.token_mlp(tok.transpose(0,2,1),P2.sub('proj_feat_emb')).transpose(0,2,1)
R = P2.sub('regressor'); fl, fr = tok[:,:21].reshape(B,-1), tok[:,21:].reshape(B,-1)
print('regress emb err', err(o[3].cpu().numpy(),emb), 'para_l', err(o[0].cpu().numpy(), N.linear(np.concatenate([fl,pl],1),R['mano_left.weight'],R['mano_left.bias'])),
      'off', err(o[2].cpu().numpy(), N.linear(np.concatenate([fl,fr,off],1),R['offset.weight'],R['offset.bias'])))
# gpos only, with hand-made inputs
xyz=[synth.synth_input('gt.xyz%d'%h,(B,21,3),SEED)*np.float32(0.05) for h in range(2)]
uvs=[bone_uv('gt.uv%d'%h,B,16) for h in range(2)]
featm = np.zeros((B,16,16,256),np.float32)
x0=torch.zeros(2,B,21,128,device='cuda'); gp=torch.zeros(2,B,21,128,device='cuda')
_capi.check(_capi.lib().dir_grid_tokens_forward(_capi.ptr(dev(featm)),0,16,256,0,0,_capi.ptr(dev(uvs[0])),_capi.ptr(dev(uvs[1])),_capi.ptr(dev(xyz[0])),_capi.ptr(dev(xyz[1])),_capi.ptr(dev(off)),st.img2joint,st.pos_emb,C.byref(st.gpos),_capi.ptr(x0),_capi.ptr(gp),B,_capi.stream_ptr()),'g')
G = P2.sub('global_pos_emb')
q = xyz[0]/np.float32(0.15) - off[:,None]/2
hid = N.relu(N.batchnorm(N.conv1d_k1(q.transpose(0,2,1), G['0.weight'], G['0.bias']), G.sub('1')))   # [B,128,21]
gref = N.conv1d_k1(hid, G['3.weight'], G['3.bias']).transpose(0,2,1)
g = gp[0].cpu().numpy()
e = np.abs(g-gref)
print('gpos left err', e.max(), 'per token', e.max(axis=(0,2)).round(3))
print(' per batch', e.max(axis=(1,2)).round(3), ' per channel block', e.reshape(B,21,4,32).max(axis=(0,1,3)).round(3))
print(' g[0,0,:6]', g[0,0,:6], 'ref', gref[0,0,:6])
print(' g[0,5,:6]', g[0,5,:6], 'ref', gref[0,5,:6])
# linear probe: does kernel output match ref for token j computed with another token's input?
for jj in range(3):
    d = np.abs(g[0,jj][None,:]-gref[0]).max(1); print('  kernel token',jj,'closest ref token', int(d.argmin()), d.min())
