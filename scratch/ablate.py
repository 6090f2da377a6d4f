import sys, math, torch, time, os
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unires_amd as U
from unires_amd import _ops, spatial
from unires_amd._project import _channel_plan
from bench import rigid_matrix
dev=torch.device('cuda:0')
dim_y=(256,256,256)
eye=torch.eye(4,dtype=torch.float64)
D=torch.diag(torch.tensor([1,1,6,1.],dtype=torch.float64))
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
p=torch.rand(dim_y,device=dev)
rigid=rigid_matrix([2.3,-1.7,3.1],[0.05,-0.08,0.03])
po=U._proj_info(dim_y,eye,(256,256,42),eye@D,rigid=rigid,device=dev)
x=[U._input(torch.rand((256,256,42),device=dev),eye@D,1.8e-4,po)]
y=U._output(torch.zeros(dim_y,device=dev),eye,0.006)
plan=_channel_plan(x,y,'super-resolution',True)
q=torch.empty_like(p)
print('dbg',os.environ.get('UNIRES_DBG'),'At %.1f us  matvec %.1f us' % (timeit(lambda:plan.proj_apply(0,'At',x[0].dat)), timeit(lambda:plan.matvec(p,0.9,0.006,out=q))))
