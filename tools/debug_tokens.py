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
# ---- PGCN: check h0/h1 of layer 0 from scratch
sdn = synth.synth_state_dict(pgcn_shapes(), SEED)
sd = {('gcn.'+k): dev(v) for k,v in sdn.items()}
keep=[]; layers = engine.pack_pgcn(sd,'gcn',keep)
B=3
x = synth.synth_input('dbg.x',(B,21,128),SEED)
out = torch.empty(B,21,128,device='cuda'); scratch = torch.zeros(2,B,21,256,device='cuda')
_capi.check(_capi.lib().dir_pgcn_stack_forward(layers,1,_capi.ptr(dev(x)),None,_capi.ptr(out),21*128,_capi.ptr(scratch),B,_capi.stream_ptr()),'p')
W = sdn['gconv_layers.0.gconv.W']
h0 = np.einsum('bjc,jcd->bjd',x,W[0]); h1=np.einsum('bjc,jcd->bjd',x,W[1])
s = scratch[0].cpu().numpy()
print('pgcn h0 err',err(s[...,:128],h0),'h1 err',err(s[...,128:],h1), 'scale', np.abs(h0).max())
P = N.Params(sdn)
ref1 = OT.graphconv_layer(x, P.sub('gconv_layers.0'))
print('pgcn layer0 out err', err(out.cpu().numpy(), ref1), 'scale', np.abs(ref1).max())
g0 = OT.pgraphconv(x, P.sub('gconv_layers.0.gconv'))
A1 = OT.edge_softmax(P['gconv_layers.0.gconv.e_1'], OT.adjacency_mask())
print('A1 row0', A1[0][[1,5,9,13,17]], 'row4', A1[4][3])
# ---- bone proj
S,dist=16,1
uv = bone_uv('bone.uv%d'%S,2,S); feat = synth.synth_input('bone.feat%d'%S,(2,21,64),SEED)
ref,mask = OT.bone_proj(uv,feat,S,dist,return_mask=True)
emb = np.concatenate([feat,feat],1)
o = torch.empty(2,S,S,2560,device='cuda'); 
_capi.check(_capi.lib().dir_bone_proj_forward(_capi.ptr(dev(uv)),_capi.ptr(dev(uv)),_capi.ptr(dev(emb)),_capi.ptr(o),None,None,2,S,float(dist),0,_capi.stream_ptr()),'b')
got = o.cpu().numpy().transpose(0,3,1,2)
gm = (got[:,:1280].reshape(2,20,64,S,S)!=0).any(2)   # [2,20,S,S]
rm = mask.transpose(0,3,1,2)
print('bone mask mismatches', int((gm!=rm).sum()), 'of', rm.size, 'ref true', int(rm.sum()), 'got true', int(gm.sum()))
idx = np.argwhere(gm!=rm)[:8]; print(idx)
print('bone val err', err(got[:,:1280],ref), 'right copy err', err(got[:,1280:],ref))
# ---- grid tokens
sdn2, sd2 = stage_sd(16)
keep=[]; st = engine.StageOp(sd2,'st',16,1,torch.float32,0,keep)
P2 = N.Params(sdn2)
featm = synth.synth_input('gt.feat',(B,256,16,16),SEED)
uvs=[bone_uv('gt.uv%d'%h,B,16) for h in range(2)]
xyz=[synth.synth_input('gt.xyz%d'%h,(B,21,3),SEED)*np.float32(0.05) for h in range(2)]
off=synth.synth_input('gt.off',(B,3),SEED)
fb = dev(featm.transpose(0,2,3,1))
x0=torch.zeros(2,B,21,128,device='cuda'); gp=torch.zeros(2,B,21,128,device='cuda')
_capi.check(_capi.lib().dir_grid_tokens_forward(_capi.ptr(fb),0,16,256,0,0,_capi.ptr(dev(uvs[0])),_capi.ptr(dev(uvs[1])),_capi.ptr(dev(xyz[0])),_capi.ptr(dev(xyz[1])),_capi.ptr(dev(off)),st.img2joint,st.pos_emb,C.byref(st.gpos),_capi.ptr(x0),_capi.ptr(gp),B,_capi.stream_ptr()),'g')
for h,side in enumerate(('left','right')):
    img = OT.img2joint(featm,uvs[h],P2.sub('img2joint_'+side))
    pos = OT.token_mlp(xyz[h].transpose(0,2,1)/np.float32(0.15),P2.sub('pos_emb_'+side)).transpose(0,2,1)
    q = xyz[h]/np.float32(0.15) + (off[:,None]/2)*(1 if h else -1)
    gref = OT.token_mlp(q.transpose(0,2,1),P2.sub('global_pos_emb')).transpose(0,2,1)
    e = np.abs(x0[h].cpu().numpy()-(pos+img))
    print(side,'x0 err',e.max(),'per-token max',e.max(axis=(0,2)).round(3),'gpos err',err(gp[h].cpu().numpy(),gref), 'scales', np.abs(img).max(), np.abs(pos).max())
