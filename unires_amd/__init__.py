"""unires_amd - MI355X-native (gfx950) implementation of the UniRes ADMM y-update.

Host-side mirror of the reference's operator interface for this path
(``unires/_project.py`` + the y-block of ``unires/_update.py``), backed by the
hand-written HIP library ``libunires_hip.so`` (C ABI: ``include/unires_hip.h``).
"""
from ._host import tune_runtime

tune_runtime()  # (ROC_SIGNAL_POOL_SIZE, unless set: before anything here can initialise the HIP runtime)

from . import _lib  # noqa: F401,E402
from .struct import _input, _output, _proj_op, settings  # noqa: F401,E402
from ._project import (_apply_scaling, _check_adjoint, _DtD, _proj, _proj_apply,  # noqa: F401,E402
                       _proj_info)
from ._core import _init_y_dat  # noqa: F401,E402
from ._update import (_admm_aux, _compute_nll, _precond, _step_size, _update_admm,  # noqa: F401,E402
                      _update_scaling, _update_y, _update_zw)

from .run import fit, _get_sched  # noqa: F401,E402
from ._util import _read_image, _write_image  # noqa: F401,E402
from ._rigid import (_expm, _rigid_match, _update_rigid, _update_rigid_channel,  # noqa: F401,E402
                     affine_basis)

__all__ = ['fit', '_get_sched', '_update_rigid', '_update_rigid_channel', '_rigid_match', '_expm',
           'affine_basis', '_read_image', '_write_image', '_input', '_output', '_proj_op', 'settings', '_proj_info', '_proj_apply', '_proj',
           '_DtD', '_apply_scaling', '_check_adjoint', '_update_admm', '_update_y', '_update_zw', '_compute_nll', '_step_size', '_admm_aux', '_init_y_dat', '_precond', '_update_scaling']
