import sys, numpy as np, torch
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
from oracle import cpu_ref
from hip_helpers import levels
torch.set_num_threads(int(sys.argv[1]) if len(sys.argv)>1 else 8)
w = cpu_ref.synthetic_vgg19_weights(bias_std=cpu_ref.TEST_BIAS_STD)
c, s = levels(1024,1536,3,1), levels(1024,1536,3,2)
init=(0.7*c[0]+0.3*cpu_ref.synthetic_image(1024,1536,seed=3)).astype(np.float32)
fx=np.load(os.path.join(R,'tests/golden/traj_lbfgs_1024x1536_L2_16.npz'))
rec=[]
for img, step in cpu_ref.run_process(c,s,init,w,"lbfgs",6,record=rec):
    pass
tot=np.array([np.array(r["rows"])[:,0].sum() for r in rec]); ref=fx["rows"][:len(tot),:,0].sum(axis=1)
print("threads",torch.get_num_threads(),"oracle",tot); print("ref   ",ref); print("rel",np.abs(tot-ref)/ref)
