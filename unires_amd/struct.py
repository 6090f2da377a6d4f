"""Data structures the hot path reads - counterparts of unires/struct.py:4-111.

Only the fields the y-update path touches carry meaning here; the rest of the
reference's ``settings`` (I/O, registration, plotting) is out of scope.
"""


class _input:
    """One observed image (unires/struct.py:4-22)."""

    def __init__(self, dat=None, mat=None, tau=1.0, po=None):
        self.dat = dat      # (X,Y,Z) float32 device tensor
        self.dim = None if dat is None else tuple(dat.shape)
        self.mat = mat      # (4,4) float64 affine
        self.tau = tau      # noise precision
        self.po = po        # _proj_op
        self.ct = False
        self.mu = 1.0
        self.sd = 1.0
        self.rigid_q = None


class _output:
    """One reconstructed channel (unires/struct.py:25-33)."""

    def __init__(self, dat=None, mat=None, lam=None):
        self.dat = dat      # (X,Y,Z) float32 device tensor, updated in place by the CG
        self.dim = None if dat is None else tuple(dat.shape)
        self.mat = mat
        self.lam = lam
        self.lam0 = lam


class _proj_op:
    """Projection-operator descriptor (unires/struct.py:36-54)."""

    def __init__(self):
        self.dim_x = None
        self.mat_x = None
        self.vx_x = None
        self.dim_y = None
        self.mat_y = None
        self.vx_y = None
        self.dim_yx = None
        self.mat_yx = None
        self.ratio = None
        self.smo_ker = None     # dense (1,1,kx,ky,kz) float32, as the reference stores it
        self.smo_ker_1d = None  # its separable factors (what the kernels consume)
        self.rigid = None
        self.scl = None
        self.dim_thick = None
        self.D_x = None
        self.D_y = None


class settings:
    """The subset of unires/struct.py:57-111 that reaches the hot path
    (SURVEY.md Appendix A), same names and defaults."""

    def __init__(self):
        self.alpha = 1.0
        self.bound = 'zero'
        self.cgs_max_iter = 20
        self.cgs_tol = 1e-3
        self.cgs_verbose = False
        self.device = 'cuda'
        self.diff = 'forward'
        self.do_proj = None
        self.gap = 0.0
        self.interpolation = 'linear'
        self.method = None
        self.profile_ip = 2
        self.profile_tp = 0
        self.reg_scl = 4.0
        self.rho = None
        self.rho_scl = 1.0
        self.tolerance = 1e-4
        self.max_iter = 512
        self.sched_num = 3
        self.rigid_mod = 1
        self.scaling = False
        self.unified_rigid = False
        self.rigid_samp = 1
        self.rigid_basis = None  # set by _update_rigid / fit: affine_basis('SE')
        self.clean_fov = False
        self.do_print = 0
        # build-side knob: which nitorch-cg objective branch to reproduce
        # ('max_gain' = what the reference passes, unires/_update.py:145)
        self.cgs_stop = 'max_gain'
        # 'jacobi': _precond (the reference has it commented out, _update.py:136);
        # 'fft': build-side FFT-diagonal preconditioner (circulant a I + rho lam^2 DtD)
        self.cgs_precond = 'none'
        # build-side knob: enqueue the (independent) channels of the y-update on separate HIP streams
        # (True / False), or 'auto': streams whenever there are several channels.  Each plan is then told about
        # its neighbours (unires_plan_set_concurrency) and sizes its persistent kernels for them.  CG iterations/s,
        # streams vs one channel after the other (profiles/r06_overlap_scan.txt): 181 x 217 x 181 +20 .. +24 %,
        # 256^3 x 3 +6 %, thick axes x / y / z +8 %, 384^3 x 4 +5 %.  (Rounds 3 - 5 stopped at 10 M voxels: with
        # every matvec kernel filling the chip three streams bought nothing at 256^3.)
        self.channel_streams = 'auto'
        # ADMM iterations the host may run ahead of its GPU before it sleeps (blocking-sync events; 0: never
        # waits - the runtime then spins in the launch calls once the hardware queue is full), _host.Pacer
        self.host_pace = 2
        # build-side knob: keep sum_n tau_n At x_n across ADMM iterations (recomputed on change)
        self.cache_atx = True
