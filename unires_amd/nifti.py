"""Minimal NIfTI-1 single-file reader / writer (.nii, .nii.gz) - the on-disk format either
side of the hot path.  The reference reads through nitorch.io
(nibabel underneath; neither is a dependency here): voxel data as float32 with the
scl_slope / scl_inter scaling applied, affine = sform if set, else qform, else the pixdim
diagonal - the same precedence nibabel's ``get_best_affine`` uses."""
import gzip
import struct

import numpy as np

_DTYPES = {2: 'u1', 4: 'i2', 8: 'i4', 16: 'f4', 64: 'f8', 256: 'i1', 512: 'u2', 768: 'u4',
           1024: 'i8', 1280: 'u8'}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith('.gz') else open(path, mode)


def _quaternion_affine(b, c, d, qfac, pixdim, offset):
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    zooms = np.array([pixdim[1], pixdim[2], pixdim[3] * (-1.0 if qfac < 0 else 1.0)])
    M = np.eye(4)
    M[:3, :3] = R * zooms[None, :]
    M[:3, 3] = offset
    return M


def read(path):
    """Returns (data float32 ndarray, affine float64 (4,4), header dict)."""
    with _open(path, 'rb') as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError('not a NIfTI-1 file: too short')
    end = '<' if struct.unpack('<i', raw[:4])[0] == 348 else '>'
    if struct.unpack(end + 'i', raw[:4])[0] != 348:
        raise ValueError('not a NIfTI-1 file: sizeof_hdr != 348')
    if raw[344:348] not in (b'n+1\0',):
        raise ValueError('only single-file NIfTI-1 (magic n+1) is supported')
    dim = struct.unpack(end + '8h', raw[40:56])
    datatype, bitpix = struct.unpack(end + '2h', raw[70:74])
    pixdim = struct.unpack(end + '8f', raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + '3f', raw[108:120])
    qform_code, sform_code = struct.unpack(end + '2h', raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(end + '6f', raw[256:280])
    srow = np.array(struct.unpack(end + '12f', raw[280:328]), dtype=np.float64).reshape(3, 4)
    if datatype not in _DTYPES:
        raise ValueError('unsupported NIfTI datatype %d' % datatype)
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    n = int(np.prod(shape))
    dt = np.dtype(end + _DTYPES[datatype])
    off = int(vox_offset)
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order='F')
    data = arr.astype(np.float32)
    if slope not in (0.0,) and not (slope == 1.0 and inter == 0.0) and np.isfinite(slope):
        data = data * np.float32(slope) + np.float32(inter)
    if sform_code > 0:
        affine = np.vstack([srow, [0, 0, 0, 1.0]])
    elif qform_code > 0:
        affine = _quaternion_affine(qb, qc, qd, pixdim[0], pixdim, (qx, qy, qz))
    else:
        affine = np.diag([pixdim[1] or 1.0, pixdim[2] or 1.0, pixdim[3] or 1.0, 1.0])
    hdr = dict(dim=shape, datatype=datatype, pixdim=pixdim[1:4], slope=slope, inter=inter,
               qform_code=qform_code, sform_code=sform_code, endian=end)
    return np.ascontiguousarray(data), affine.astype(np.float64), hdr


def write(path, data, affine):
    """float32 NIfTI-1 with the affine in the sform (code 2, 'aligned')."""
    data = np.asarray(data, dtype=np.float32)
    if data.ndim < 1 or data.ndim > 7:
        raise ValueError('NIfTI-1 stores 1 to 7 dimensions')
    affine = np.asarray(affine, dtype=np.float64)
    hdr = bytearray(348)
    struct.pack_into('<i', hdr, 0, 348)
    dims = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    struct.pack_into('<8h', hdr, 40, *dims)
    struct.pack_into('<2h', hdr, 70, 16, 32)
    vox = np.sqrt((affine[:3, :3] ** 2).sum(0))
    pix = [1.0] + vox.tolist() + [1.0] * 4
    struct.pack_into('<8f', hdr, 76, *pix)
    struct.pack_into('<3f', hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2  # xyzt_units: millimetres
    struct.pack_into('<2h', hdr, 252, 0, 2)
    struct.pack_into('<12f', hdr, 280, *affine[:3, :].reshape(-1).tolist())
    hdr[344:348] = b'n+1\0'
    with _open(path, 'wb') as f:
        f.write(bytes(hdr))
        f.write(b'\0\0\0\0')
        f.write(np.asfortranarray(data).tobytes(order='F'))
