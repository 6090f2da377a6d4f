"""Image I/O of the drop-in: ``_read_image`` / ``_write_image`` with the contract of
unires/_util.py:134-226 (same arguments, same 8-tuple, same errors), on the package's own NIfTI-1
codec (``nifti.py``) instead of nitorch.io."""
import os

import numpy as np
import torch

from . import nifti


def _as_tensor(v, dtype, device):
    t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
    return t.to(device=device, dtype=dtype)


def _read_image(data, device='cpu', is_ct=False):
    """``data``: a ``.nii`` / ``.nii.gz`` path, or a ``[dat, mat]`` pair (array-likes or tensors).

    Returns ``(dat, dim, mat, fname, direc, nam, file, ct)``: float32 (X, Y, Z) voxels with every
    non-finite value replaced by zero, its shape, the float64 4x4 voxel-to-world matrix, the file
    path / directory / name (``None`` for in-memory data), the header the file was read with, and
    the CT flag.  Anything that does not squeeze to three dimensions raises ``ValueError``."""
    fname = direc = nam = header = None
    if isinstance(data, str):
        voxels, affine, header = nifti.read(data)
        dat = _as_tensor(voxels, torch.float32, device)
        mat = _as_tensor(affine, torch.float64, device)
        fname = data
        direc, nam = os.path.split(os.path.abspath(data))
    else:
        dat = _as_tensor(data[0], torch.float32, device)
        mat = _as_tensor(data[1], torch.float64, device)
    dat = torch.nan_to_num(dat.squeeze(), nan=0.0, posinf=0.0, neginf=0.0)
    if dat.dim() != 3:
        raise ValueError("Input image dimension required to be 3D, recieved {:}D!".format(dat.dim()))
    return dat, tuple(dat.shape), mat, fname, direc, nam, header, bool(is_ct)


def _bids_name(fname):
    """``sub-01_T1w.nii`` -> ``sub-01_space-unires_T1w.nii`` (the tag goes before the suffix)."""
    folder, name = os.path.split(fname)
    head, sep, suffix = name.rpartition('_')
    return os.path.join(folder, head + sep + 'space-unires_' + suffix)


def _write_image(dat, fname, bids=False, mat=torch.eye(4), file=None, dtype='float32',
                 do_print=False):
    """Writes ``dat`` with affine ``mat`` as NIfTI-1 (float32); returns the path written."""
    if bids:
        fname = _bids_name(fname)
    voxels = torch.as_tensor(dat).detach().cpu().numpy()
    affine = torch.as_tensor(mat).detach().cpu().double().numpy()
    nifti.write(fname, voxels, affine)
    if do_print:
        print(f"Output saved to: {fname}")
    return fname
