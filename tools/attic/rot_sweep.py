import sys, math, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unires_amd as U
from unires_amd import _ops, spatial
from unires_amd._project import _channel_plan
from bench import rigid_matrix
dev=torch.device('cuda:0')
dim_y=(256,256,256)
eye=torch.eye(4,dtype=torch.float64)
D=torch.diag(torch.tensor([1,1,6,1.],dtype=torch.float64))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
p=torch.rand(dim_y,device=dev)
for rot,axis in [(0,None),(0.02,'all'),(0.1,'z'),(0.1,'x'),(0.1,'y'),(0.1,'all')]:
    r=[0,0,0]
    if axis=='all': r=[rot,rot,rot]
    elif axis: r['xyz'.index(axis)]=rot
    rigid=rigid_matrix([2.3,-1.7,3.1], r)
    po=U._proj_info(dim_y,eye,(256,256,42),eye@D,rigid=rigid,device=dev)
    from unires_amd._plan import proj_matrix
    mat,dg=proj_matrix(po,'super-resolution')
    M=spatial._m12(mat)
    t_pull=timeit(lambda:_ops.pull_affine(p,M,dg))
    g=_ops.pull_affine(p,M,dg)
    t_push=timeit(lambda:_ops.push_affine(g,M,dim_y))
    x=[U._input(torch.rand((256,256,42),device=dev),eye@D,1.8e-4,po)]
    y=U._output(torch.zeros(dim_y,device=dev),eye,0.006)
    plan=_channel_plan(x,y,'super-resolution',True)
    q=torch.empty_like(p)
    t_mv=timeit(lambda:plan.matvec(p,0.9,0.006,out=q))
    t_A=timeit(lambda:plan.proj_apply(0,'A',p))
    t_At=timeit(lambda:plan.proj_apply(0,'At',x[0].dat))
    print('rot %.2f %-4s pull %7.1f push(direct) %7.1f | A(pull_conv) %7.1f At(push convup) %7.1f matvec %7.1f us' % (rot,axis,t_pull,t_push,t_A,t_At,t_mv))
