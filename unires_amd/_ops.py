"""Tensor-level wrappers of the op-level C ABI (one per nitorch / torch call
on the path).  All tensors must be float32 CUDA(HIP) tensors; outputs are
allocated with torch (PyTorch is device-memory / stream plumbing only)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, f3, f12, i3

FOV_TOL = 5e-2  # nitorch's in-FOV tolerance for extrapolate=False


def _stream():
    """The caller's current HIP stream ON THE CURRENT DEVICE.  Every entry point runs under
    ``on_device`` below, so "current device" is the device of the tensors it was handed."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _find_device(obj, depth=0):
    """Device of the first CUDA tensor reachable from the arguments (tensors, lists of tensors,
    _input / _output structs, plans)."""
    if isinstance(obj, torch.Tensor):
        return obj.device if obj.is_cuda else None
    dev = getattr(obj, 'device', None)
    if isinstance(dev, torch.device) and dev.type == 'cuda':
        return dev
    if depth < 3:
        if isinstance(obj, (list, tuple)):
            for o in obj:
                d = _find_device(o, depth + 1)
                if d is not None:
                    return d
        dat = getattr(obj, 'dat', None)
        if isinstance(dat, torch.Tensor) and dat.is_cuda:
            return dat.device
    return None


def on_device(fn):
    """Run ``fn`` with the device of its tensors as the current HIP device: the library allocates
    plan workspace with hipMalloc and launches on the current device's stream, so a call made
    while another GPU is current would otherwise land on the wrong one (one process per GPU is the
    normal mode, but nothing should depend on it)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = None
        for a in list(args) + list(kwargs.values()):
            dev = _find_device(a)
            if dev is not None:
                break
        if dev is None or (dev.index is not None and dev.index == torch.cuda.current_device()):
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


def _vol(t, name='dat'):
    """(…,X,Y,Z) float32 device tensor -> contiguous 3-D view + leading shape."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('unires_amd: %s must be a CUDA/HIP tensor (no CPU path)' % name)
    if t.dtype != torch.float32:
        raise TypeError('unires_amd: %s must be float32' % name)
    if t.dim() < 3 or any(s != 1 for s in t.shape[:-3]):
        raise ValueError('unires_amd: %s must be (X,Y,Z) with optional leading 1-dims' % name)
    lead = tuple(t.shape[:-3])
    return t.reshape(t.shape[-3:]).contiguous(), lead


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _taps_arg(taps):
    keep = [np.ascontiguousarray(np.asarray(k, dtype=np.float32)) for k in taps]
    arr = _lib.c_fptrx3(*[k.ctypes.data_as(C.POINTER(C.c_float)) for k in keep])
    return arr, keep, i3([len(k) for k in keep])


@on_device
def pull_affine(src, M, gdim, fov_tol=FOV_TOL):
    """grid_pull(src, affine_grid(M, gdim)) - linear, zero bound, extrapolate=False."""
    s, lead = _vol(src, 'src')
    out = torch.empty(tuple(gdim), dtype=torch.float32, device=s.device)
    check(_lib.load().unires_pull3d_affine(_ptr(s), i3(s.shape), f12(M), _ptr(out), i3(gdim),
                                           fov_tol, _stream()))
    return out.reshape(lead + tuple(gdim))


@on_device
def pull_grad_affine(src, M, gdim, fov_tol=FOV_TOL):
    """grid_grad(src, affine_grid(M, gdim)) -> (gdim, 3): spatial gradient of the trilinear sample."""
    s, lead = _vol(src, 'src')
    out = torch.empty(tuple(gdim) + (3,), dtype=torch.float32, device=s.device)
    check(_lib.load().unires_pull_grad3d_affine(_ptr(s), i3(s.shape), f12(M), _ptr(out), i3(gdim),
                                                fov_tol, _stream()))
    return out.reshape(lead + tuple(gdim) + (3,))


@on_device
def push_affine(src, M, ddim, alpha=1.0, out=None, fov_tol=FOV_TOL):
    """grid_push(src, affine_grid(M, src.shape), shape=ddim); out += if given."""
    s, lead = _vol(src, 'src')
    acc = out is not None
    if out is None:
        out = torch.empty(tuple(ddim), dtype=torch.float32, device=s.device)
    elif not out.is_contiguous() or tuple(out.shape[-3:]) != tuple(ddim):
        raise ValueError('unires_amd: bad out tensor')
    check(_lib.load().unires_push3d_affine(_ptr(s), i3(s.shape), f12(M), _ptr(out), i3(ddim),
                                           alpha, fov_tol, int(acc), _stream()))
    return out if acc else out.reshape(lead + tuple(ddim))


@on_device
def conv_down(src, taps, stride, scl=0.0, scl_dim=0):
    """F.conv3d(src, outer(taps), stride=stride) [+ even/odd scaling]."""
    s, lead = _vol(src, 'src')
    arr, keep, nt = _taps_arg(taps)
    ddim = []
    for d in range(3):
        num = s.shape[d] - len(keep[d])
        if num < 0 or num % int(stride[d]) != 0:
            raise ValueError('unires_amd: conv dims: need hi = (lo-1)*stride + ntaps')
        ddim.append(num // int(stride[d]) + 1)
    out = torch.empty(tuple(ddim), dtype=torch.float32, device=s.device)
    check(_lib.load().unires_conv_down3d(_ptr(s), i3(s.shape), arr, nt, i3(stride), _ptr(out),
                                         i3(ddim), float(scl), int(scl_dim), _stream()))
    return out.reshape(lead + tuple(ddim))


@on_device
def conv_up(src, taps, stride, scl=0.0, scl_dim=0):
    """F.conv_transpose3d(S(scl) src, outer(taps), stride=stride)."""
    s, lead = _vol(src, 'src')
    arr, keep, nt = _taps_arg(taps)
    ddim = [(s.shape[d] - 1) * int(stride[d]) + len(keep[d]) for d in range(3)]
    out = torch.empty(tuple(ddim), dtype=torch.float32, device=s.device)
    check(_lib.load().unires_conv_up3d(_ptr(s), i3(s.shape), arr, nt, i3(stride), _ptr(out),
                                       i3(ddim), float(scl), int(scl_dim), _stream()))
    return out.reshape(lead + tuple(ddim))


def _vx3(vx):
    if vx is None:
        return (1.0, 1.0, 1.0)
    if isinstance(vx, torch.Tensor):
        vx = vx.detach().cpu().tolist()
    return tuple(float(v) for v in vx)


@on_device
def grad_fwd_zero(dat, vx=None):
    s, _ = _vol(dat)
    out = torch.empty((3,) + tuple(s.shape), dtype=torch.float32, device=s.device)
    check(_lib.load().unires_grad_fwd_zero(_ptr(s), i3(s.shape), f3(_vx3(vx)), _ptr(out),
                                           _stream()))
    return out


@on_device
def div_fwd_zero(dat3, vx=None):
    if dat3.dim() != 4 or dat3.shape[0] != 3:
        raise ValueError('unires_amd: divergence input must be (3,X,Y,Z)')
    if not dat3.is_cuda or dat3.dtype != torch.float32:
        raise RuntimeError('unires_amd: divergence input must be a float32 CUDA/HIP tensor')
    s = dat3.contiguous()
    out = torch.empty(tuple(s.shape[1:]), dtype=torch.float32, device=s.device)
    check(_lib.load().unires_div_fwd_zero(_ptr(s), i3(s.shape[1:]), f3(_vx3(vx)), _ptr(out),
                                          _stream()))
    return out


@on_device
def dtd(dat, vx=None, a=0.0, c=1.0):
    """a*dat + c*DtD(dat)."""
    s, lead = _vol(dat)
    out = torch.empty_like(s)
    check(_lib.load().unires_dtd(_ptr(s), i3(s.shape), f3(_vx3(vx)), float(a), float(c),
                                 _ptr(out), _stream()))
    return out.reshape(lead + tuple(s.shape))
