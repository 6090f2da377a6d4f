"""ctypes binding of ``libunires_hip.so`` (C ABI in ``include/unires_hip.h``).

The library is the product: if it is missing this module raises - there is no
CPU / eager fallback anywhere in ``unires_amd``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (UNIRES_LIB: another build of the SAME library - A/B measurements of kernel changes on one box; never a fallback)
LIB_PATH = os.environ.get('UNIRES_LIB') or os.path.join(_HERE, 'libunires_hip.so')

UNIRES_MAX_TAPS = 32
OP = {'A': 0, 'At': 1, 'AtA': 2}
REGIME_IDENTITY, REGIME_DENOISE, REGIME_SUPERRES = 0, 1, 2
# nitorch cg(stop=...): 'e' = residual norm; anything else = the objective 0.5 sum x (A(x) - 2b) (UniRes passes 'max_gain').
# 'max_gain' runs GUARDED (C ABI mode 3): the objective comes from the recurred residual while its gain is >= 4 x tolerance
# and is evaluated afresh - a second A(x), nitorch's own arithmetic - from there on, so close decisions are nitorch's;
# 'max_gain_fresh' evaluates it afresh after every iteration (mode 1), 'max_gain_recurred' never (mode 2).
STOP = {'e': 0, 'max_gain': 3, 'max_gain_guarded': 3, 'max_gain_fresh': 1, 'max_gain_recurred': 2}
PRECOND = {'none': 0, 'identity': 0, 'jacobi': 1, 'fft': 2}

c_i32x3 = C.c_int32 * 3
c_f32x3 = C.c_float * 3
c_f32x12 = C.c_float * 12
c_fptrx3 = C.POINTER(C.c_float) * 3


class Repeat(C.Structure):
    """``unires_repeat_t``."""
    _fields_ = [('dim_x', c_i32x3), ('dim_g', c_i32x3), ('M', c_f32x12), ('ratio', c_i32x3),
                ('ntaps', c_i32x3), ('taps', c_fptrx3), ('scl', C.c_float),
                ('dim_thick', C.c_int32), ('tau', C.c_float)]


# name -> (restype, argtypes); must list every symbol include/unires_hip.h declares
SIGNATURES = {
    'unires_last_error': (C.c_char_p, []),
    'unires_abi_version': (C.c_int, []),
    'unires_pull3d_affine': (C.c_int, [C.c_void_p, c_i32x3, c_f32x12, C.c_void_p, c_i32x3,
                                       C.c_float, C.c_void_p]),
    'unires_pull_grad3d_affine': (C.c_int, [C.c_void_p, c_i32x3, c_f32x12, C.c_void_p, c_i32x3,
                                            C.c_float, C.c_void_p]),
    'unires_push3d_affine': (C.c_int, [C.c_void_p, c_i32x3, c_f32x12, C.c_void_p, c_i32x3,
                                       C.c_float, C.c_float, C.c_int, C.c_void_p]),
    'unires_conv_down3d': (C.c_int, [C.c_void_p, c_i32x3, c_fptrx3, c_i32x3, c_i32x3, C.c_void_p,
                                     c_i32x3, C.c_float, C.c_int32, C.c_void_p]),
    'unires_conv_up3d': (C.c_int, [C.c_void_p, c_i32x3, c_fptrx3, c_i32x3, c_i32x3, C.c_void_p,
                                   c_i32x3, C.c_float, C.c_int32, C.c_void_p]),
    'unires_grad_fwd_zero': (C.c_int, [C.c_void_p, c_i32x3, c_f32x3, C.c_void_p, C.c_void_p]),
    'unires_div_fwd_zero': (C.c_int, [C.c_void_p, c_i32x3, c_f32x3, C.c_void_p, C.c_void_p]),
    'unires_dtd': (C.c_int, [C.c_void_p, c_i32x3, c_f32x3, C.c_float, C.c_float, C.c_void_p,
                             C.c_void_p]),
    'unires_plan_create': (C.c_int, [C.POINTER(C.c_void_p), c_i32x3, c_f32x3, C.c_int32,
                                     C.c_int32, C.POINTER(Repeat), C.c_float]),
    'unires_plan_destroy': (C.c_int, [C.c_void_p]),
    'unires_plan_set_repeat': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Repeat)]),
    'unires_plan_set_concurrency': (C.c_int, [C.c_void_p, C.c_int32]),
    'unires_plan_workspace_bytes': (C.c_int64, [C.c_void_p]),
    'unires_orient_of': (C.c_int, [c_f32x12, c_i32x3, c_i32x3]),
    'unires_plan_repeat_info': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32 * 8]),
    'unires_plan_time_matvecs': (C.c_int, [C.c_void_p, C.c_int32]),
    'unires_plan_matvec_time': (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    'unires_proj_apply': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    'unires_ata_matvec': (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    'unires_rhs_assemble': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                      C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    'unires_atx_assemble': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    'unires_rhs_from_atx': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                      C.c_float, C.c_void_p, C.c_void_p]),
    'unires_precond_build': (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p,
                                       C.c_void_p]),
    'unires_precond_apply': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'unires_cg_solve': (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                  C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p]),
    'unires_cg_solve_many': (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_float),
                                       C.POINTER(C.c_float), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.c_int32, C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                       C.POINTER(C.c_double), C.POINTER(C.c_void_p)]),
    'unires_zw_update': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.c_int32, c_i32x3,
                                   c_f32x3, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    'unires_nll_prior': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.c_int32, c_i32x3,
                                   c_f32x3, C.c_void_p, C.c_void_p]),
    'unires_scaling_sums': (C.c_int, [C.c_void_p, C.c_void_p, c_i32x3, C.c_int32, C.c_void_p,
                                      C.c_void_p]),
    'unires_rigid_sums': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i32x3, C.c_float * 72,
                                    C.c_void_p, C.c_void_p]),
    'unires_clean_fov': (C.c_int, [C.c_void_p, c_i32x3, c_f32x12, c_i32x3, C.c_void_p]),
    'unires_masked_sse': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'unires_mark_create': (C.c_int, [C.POINTER(C.c_void_p)]),
    'unires_mark_destroy': (C.c_int, [C.c_void_p]),
    'unires_mark_signal': (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    'unires_mark_read': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
}

_lib = None

# status code -> exception type, mirroring the reference's Python errors
# (ValueError('Undefined operator'/'Undefined method'), unires/_project.py:123-126)
_EXC = {1: ValueError, 2: ValueError, 3: ValueError, 4: RuntimeError, 5: MemoryError,
        6: NotImplementedError}


def load():
    """Load (once) and return the bound library; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'unires_amd: %s is missing. Build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.unires_abi_version() != 1:
        raise RuntimeError('unires_amd: ABI version mismatch')
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().unires_last_error().decode()
        raise _EXC.get(status, RuntimeError)(msg)


def i3(v):
    return c_i32x3(*[int(t) for t in v])


def f3(v):
    return c_f32x3(*[float(t) for t in v])


def f12(v):
    return c_f32x12(*[float(t) for t in v])
