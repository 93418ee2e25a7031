"""train_step eager against GraphedTrainStep (everything but the optimiser as one HIP graph) at B images ([BACKBONE=hrnet_w48]): seconds per step and bit-equality of the
parameters after the same number of steps.  python tools/bench_train_graphed.py [batch]"""
import json, os, sys, time
import numpy as np, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from dir_amd import synth
from dir_amd.optim import FlatAdamW
from dir_amd.train import step as TSTEP
B=int(sys.argv[1]) if len(sys.argv)>1 else 32
shapes={k:tuple(v) for k,v in json.load(open(os.path.join(ROOT,'tests','golden','manifest_dir.json'))).items()}
if os.environ.get('BACKBONE','resnet50')=='hrnet_w48':        # BASELINE configs[4]: HRNet-W48 + init + 4 refinement stages
    from dir_amd.models.dir import DIR
    shapes={k:tuple(v.shape) for k,v in DIR(21,'unused',0,backbone='hrnet_w48',extra_stages=int(os.environ.get('EXTRA_STAGES','2'))).state_dict().items()}
sd=synth.synth_state_dict(shapes,1234)
is_buf=lambda k: any(t in k for t in ('running_','num_batches','mano_layer','img_gird','seg_loss.weight'))
def make():
    params={k:torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(v)).cuda()) for k,v in sd.items() if not is_buf(k)}
    buffers={k:torch.from_numpy(np.ascontiguousarray(v)).cuda() for k,v in sd.items() if is_buf(k) and 'num_batches' not in k}
    opt=FlatAdamW(list(params.values()),lr=1e-5); opt.set_inactive(TSTEP.inactive_parameters(params))
    return params,buffers,opt
rng=np.random.RandomState(0)
dv=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
img=dv(synth.synth_input('train.img.0',(B,3,256,256),1234))
target,meta={},{}
for s in ('left','right'):
    target['joint_2d_'+s]=dv(rng.uniform(-1,1,(B,21,3)).astype(np.float32)); target['mesh_2d_'+s]=dv(rng.uniform(-1,1,(B,778,3)).astype(np.float32))
    target['joint_3d_'+s]=dv(rng.normal(0,0.05,(B,21,3)).astype(np.float32)); target['mesh_3d_'+s]=dv(rng.normal(0,0.05,(B,778,3)).astype(np.float32))
    meta['center_'+s]=dv(rng.normal(0,0.1,(B,1,3)).astype(np.float32))
target['seg']=dv(rng.randint(0,3,(B,1,256,256)).astype(np.float32)); target['dense']=dv(rng.rand(B,3,256,256).astype(np.float32))
faces=tuple(dv(synth.loss_faces(s,1234).astype(np.int64)) for s in ('left','right'))
N=int(os.environ.get('NSTEP','8'))
p1,b1,o1=make()
for i in range(N): TSTEP.train_step(p1,b1,img,target,meta,faces,o1,overlap_allreduce=False)
torch.cuda.synchronize(); t0=time.time()
for i in range(5): TSTEP.train_step(p1,b1,img,target,meta,faces,o1,overlap_allreduce=False)
torch.cuda.synchronize(); te=(time.time()-t0)/5
p2,b2,o2=make()
gs=TSTEP.GraphedTrainStep(p2,b2,o2,faces)
for i in range(N): l=gs(img,target,meta)
torch.cuda.synchronize(); t0=time.time()
for i in range(5): l=gs(img,target,meta)
torch.cuda.synchronize(); tg=(time.time()-t0)/5
print('eager %.4f s  graphed %.4f s  params equal: %s  loss %s' % (te,tg,torch.equal(o1.flat_param,o2.flat_param), float(sum(v for v in l.values()))))
