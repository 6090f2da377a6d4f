"""_read_image / _write_image - host-side mirror of unires/_util.py:134-226 on the built-in
NIfTI-1 reader (the reference goes through nitorch.io)."""
import os

import torch

from . import nifti


def _read_image(data, device='cpu', is_ct=False):
    """Reads image data (unires/_util.py:134-193): ``data`` is a path (.nii | .nii.gz) or
    [dat, mat].  Returns (dat float32 (X,Y,Z), dim, mat float64 (4,4), fname, direc, nam,
    file, ct); non-finite voxels are zeroed, anything but 3-D raises ValueError."""
    if isinstance(data, str):
        arr, aff, hdr = nifti.read(data)
        dat = torch.from_numpy(arr).to(device)
        mat = torch.from_numpy(aff).to(device).type(torch.float64)
        fname = data
        direc, nam = os.path.split(os.path.abspath(fname))
        file = hdr
    else:
        dat = data[0]
        if not isinstance(dat, torch.Tensor):
            dat = torch.tensor(dat)
        dat = dat.float().to(device)
        mat = data[1]
        if not isinstance(mat, torch.Tensor):
            mat = torch.tensor(mat)
        mat = mat.double().to(device)
        file = fname = direc = nam = None
    dat = dat.squeeze()
    dim = tuple(dat.shape)
    if len(dim) != 3:
        raise ValueError("Input image dimension required to be 3D, recieved {:}D!".format(len(dim)))
    dat = dat.clone()
    dat[~torch.isfinite(dat)] = 0.0
    return dat, dim, mat, fname, direc, nam, file, bool(is_ct)


def _write_image(dat, fname, bids=False, mat=torch.eye(4), file=None, dtype='float32',
                 do_print=False):
    """Write data to nifti (unires/_util.py:214-226); returns the path written."""
    if bids:
        p, n = os.path.split(fname)
        s = n.split('_')
        fname = os.path.join(p, '_'.join(s[:-1] + ['space-unires'] + [s[-1]]))
    nifti.write(fname, torch.as_tensor(dat).detach().cpu().numpy(),
                torch.as_tensor(mat).detach().cpu().double().numpy())
    if do_print:
        print(f"Output saved to: {fname}")
    return fname
