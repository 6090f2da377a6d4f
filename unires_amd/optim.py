"""nitorch.core.optim-shaped entry points used by UniRes
(cg: unires/_update.py:9,142-148; get_gain: unires/run.py:7,100)."""
import math

import torch


def get_gain(obj, monotonicity='increasing'):
    """Normalised gain of the last objective value (host-side scalar logic)."""
    vals = obj.tolist() if isinstance(obj, torch.Tensor) else [float(v) for v in obj]
    if len(vals) <= 1:
        return torch.tensor(float('inf'), dtype=torch.float64)
    gain = vals[-1] - vals[-2] if monotonicity == 'increasing' else vals[-2] - vals[-1]
    span = max(vals) - min(vals)
    # plain float arithmetic (IEEE: x/0 -> inf, 0/0 -> nan, as the tensor version would give)
    out = gain / span if span != 0.0 else (float('nan') if gain == 0.0 else math.copysign(float('inf'), gain))
    return torch.tensor(out, dtype=torch.float64)


def cg(A, b, x=None, precond=None, max_iter=None, tolerance=1e-5, verbose=False,
       sum_dtype=torch.float64, inplace=True, stop='E', rho=1.0, lam=1.0):
    """Solve A x = b by conjugate gradients, entirely on the device.

    ``A`` is a :class:`unires_amd._plan.ChannelPlan` (the fused
    ``sum tau AtA + rho lam^2 DtD`` of one channel) instead of a Python callable:
    the whole iteration - matvec, float64 dots, alpha/beta, objective and the
    ``|gain| < tolerance`` test - is enqueued without host round trips.
    """
    mode = 'none'
    if precond is not None:
        # only the objects made by unires_amd._update._precond are device-resident
        mode = getattr(precond, 'mode', None)
        if mode not in ('none', 'jacobi'):
            raise NotImplementedError('precond must come from unires_amd._update._precond '
                                      '(identity or Jacobi, unires/_update.py:80-102,136-137)')
        if mode == 'jacobi' and precond.plan is not A:
            raise ValueError('preconditioner was built for another channel plan')
    if sum_dtype != torch.float64:
        raise NotImplementedError('dot products accumulate in float64')
    if max_iter is None:
        # nitorch: 10 numel.  With a tolerance the solve is enqueued chunk by chunk and takes any budget;
        # without one every iteration would really run, and one captured solve holds 4 096 of them
        max_iter = min(10 * b.numel(), 2 ** 31 - 1) if tolerance else 4096  # (int32 at the C ABI)
    if x is None:
        x = torch.zeros_like(b)
    elif not inplace:
        x = x.clone()
    A.cg(b, x, rho, lam, max_iter=max_iter, tolerance=tolerance, stop=stop, sync=True,
         precond=mode)
    return x
