"""NIfTI-1 reader / writer and the _read_image / _write_image mirrors (CPU only)."""
import gzip
import os
import struct

import numpy as np
import pytest
import torch

from unires_amd import nifti
from unires_amd._util import _read_image, _write_image


def test_round_trip_and_gzip(tmp_path):
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((7, 5, 9)).astype(np.float32)
    aff = np.array([[0, -2.0, 0, 10], [1.5, 0, 0, -4], [0, 0, 3.0, 2.5], [0, 0, 0, 1]])
    for name in ('a.nii', 'a.nii.gz'):
        p = str(tmp_path / name)
        nifti.write(p, vol, aff)
        out, a2, hdr = nifti.read(p)
        assert out.dtype == np.float32 and np.array_equal(out, vol)
        assert np.allclose(a2, aff) and hdr['sform_code'] == 2
        assert np.allclose(hdr['pixdim'], [1.5, 2.0, 3.0])
    assert open(str(tmp_path / 'a.nii.gz'), 'rb').read(2) == b'\x1f\x8b'


def _hand_built(path, end, qform):
    """int16 volume, slope/intercept, qform-only or header-less affine, chosen endianness."""
    shape = (4, 3, 2)
    vals = np.arange(24, dtype=np.int16).reshape(shape, order='F')
    hdr = bytearray(348)
    struct.pack_into(end + 'i', hdr, 0, 348)
    struct.pack_into(end + '8h', hdr, 40, 3, 4, 3, 2, 1, 1, 1, 1)
    struct.pack_into(end + '2h', hdr, 70, 4, 16)
    struct.pack_into(end + '8f', hdr, 76, -1.0 if qform else 1.0, 2.0, 3.0, 4.0, 1, 1, 1, 1)
    struct.pack_into(end + '3f', hdr, 108, 352.0, 0.5, 10.0)
    if qform:  # 90 degree rotation about z: b = c = 0, d = sin(45 deg)
        struct.pack_into(end + '2h', hdr, 252, 1, 0)
        struct.pack_into(end + '6f', hdr, 256, 0.0, 0.0, np.sqrt(0.5), 5.0, 6.0, 7.0)
    hdr[344:348] = b'n+1\0'
    with open(path, 'wb') as f:
        f.write(bytes(hdr) + b'\0\0\0\0' + vals.astype(end + 'i2').tobytes(order='F'))
    return vals


@pytest.mark.parametrize('end', ['<', '>'])
def test_reads_scaled_int16_with_qform(tmp_path, end):
    p = str(tmp_path / 'q.nii')
    vals = _hand_built(p, end, qform=True)
    out, aff, hdr = nifti.read(p)
    assert np.allclose(out, vals * 0.5 + 10.0) and hdr['endian'] == end
    want = np.array([[0, -3.0, 0, 5], [2.0, 0, 0, 6], [0, 0, -4.0, 7], [0, 0, 0, 1]])  # qfac = -1
    assert np.allclose(aff, want, atol=1e-6)
    p2 = str(tmp_path / 'p.nii')
    _hand_built(p2, end, qform=False)
    assert np.allclose(nifti.read(p2)[1], np.diag([2.0, 3.0, 4.0, 1.0]))


def test_read_write_image_mirrors(tmp_path):
    vol = torch.rand(6, 5, 4)
    vol[1, 1, 1] = float('nan')
    mat = torch.diag(torch.tensor([1.0, 1.0, 6.0, 1.0], dtype=torch.float64))
    p = _write_image(vol.nan_to_num(7.0), str(tmp_path / 'sub-01_T1w.nii.gz'), bids=True, mat=mat)
    assert p.endswith('sub-01_space-unires_T1w.nii.gz') and os.path.exists(p)
    dat, dim, m, fname, direc, nam, file, ct = _read_image(p)
    assert dim == (6, 5, 4) and dat.dtype == torch.float32 and m.dtype == torch.float64
    assert torch.equal(m, mat) and nam == 'sub-01_space-unires_T1w.nii.gz' and ct is False
    dat2, dim2, m2, *_rest = _read_image([vol.numpy(), mat.numpy()], is_ct=True)
    assert dat2[1, 1, 1] == 0 and _rest[-1] is True and _rest[0] is None
    with pytest.raises(ValueError, match='3D'):
        _read_image([torch.rand(3, 3), mat])
    with pytest.raises(ValueError):
        with gzip.open(str(tmp_path / 'bad.nii.gz'), 'wb') as f:
            f.write(b'\0' * 400)
        nifti.read(str(tmp_path / 'bad.nii.gz'))


def test_reads_the_reference_demo_volume_if_present():
    """BASELINE configs[0] names data/t1_icbm_normal_1mm_pn0_rf0.nii.gz (BrainWeb T1, 1 mm)."""
    p = '/root/reference/data/t1_icbm_normal_1mm_pn0_rf0.nii.gz'
    if not os.path.exists(p):
        pytest.skip('reference data not on this machine')
    dat, dim, mat, *_ = _read_image(p)
    assert dim == (181, 217, 181)
    assert torch.allclose((mat[:3, :3] ** 2).sum(0).sqrt(), torch.ones(3, dtype=torch.float64))
