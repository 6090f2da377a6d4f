import sys, math, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unires_amd as U
from unires_amd._project import _channel_plan
from bench import rigid_matrix
dev=torch.device('cuda:0')
dim_y=(256,256,256)
eye=torch.eye(4,dtype=torch.float64)
D=torch.diag(torch.tensor([1,1,6,1.],dtype=torch.float64))
p=torch.rand(dim_y,device=dev)
rigid=rigid_matrix([2.3,-1.7,3.1],[0.05,-0.08,0.03])
po=U._proj_info(dim_y,eye,(256,256,42),eye@D,rigid=rigid,device=dev)
x=[U._input(torch.rand((256,256,42),device=dev),eye@D,1.8e-4,po)]
y=U._output(torch.zeros(dim_y,device=dev),eye,0.006)
plan=_channel_plan(x,y,'super-resolution',True)
q=torch.empty_like(p)
for _ in range(3):
    plan.matvec(p,0.9,0.006,out=q)
    plan.proj_apply(0,'At',x[0].dat)
torch.cuda.synchronize()
