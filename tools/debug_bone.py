import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dir_amd import _capi, synth
from oracle import tokens as OT
from oracle.golden_inputs import bone_uv
SEED=1234
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
g = dict(np.load('tests/golden/g5_bone.npz'))
for S,dist in ((16,1),(32,2)):
    uv = g['S%d.uv'%S]; feat = synth.synth_input('bone.feat%d'%S,(2,21,64),SEED)
    ref = g['S%d.y'%S]
    _, mask = OT.bone_proj(uv,feat,S,dist,return_mask=True)      # [2,S,S,20]
    emb = np.concatenate([feat,feat],1)
    duv, demb = dev(uv), dev(emb)
    o = torch.empty(2,S,S,2560,device='cuda')
    _capi.check(_capi.lib().dir_bone_proj_forward(_capi.ptr(duv),_capi.ptr(duv),_capi.ptr(demb),_capi.ptr(o),None,None,2,S,float(dist),0,_capi.stream_ptr()),'b')
    got = o.cpu().numpy()                                         # [2,S,S,2560]
    gm = (got[...,:1280].reshape(2,S,S,20,64)!=0).any(-1)
    rm = (ref.reshape(2,20,64,S,S)!=0).any(2).transpose(0,2,3,1)
    print('S',S,'oracle-vs-golden mask mismatches', int((mask!=rm).sum()), ' hip-vs-golden', int((gm!=rm).sum()), 'true count', int(rm.sum()))
    for b,y,x,bone in np.argwhere(gm!=rm)[:10]:
        pa, ch = OT.PARENT[bone], OT.CHILD[bone]
        P = ((uv[b].astype(np.float32)+1)/2*S).astype(np.float32)
        p = np.array([[x+0.5,y+0.5]],np.float32)
        d32 = OT.lineseg_dists(p, P[pa][None], P[ch][None])[0]
        d64 = OT.lineseg_dists(p.astype(np.float64), P[pa][None].astype(np.float64), P[ch][None].astype(np.float64))[0]
        print('   b',b,'y',y,'x',x,'bone',bone,'got',bool(gm[b,y,x,bone]),'ref',bool(rm[b,y,x,bone]),'dist32 %.9g dist64 %.12g'%(d32,d64), 'a',P[pa],'b',P[ch])
    print('   value err where both', float(np.abs(got[...,:1280].reshape(2,S,S,20,64).transpose(0,3,4,1,2).reshape(2,1280,S,S)-ref)[np.repeat((gm==rm).transpose(0,3,1,2),64,1)].max()))
