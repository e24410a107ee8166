"""ctypes binding of libstrive_hip.so (include/strive_hip.h).

The product loads exactly one library: ``strive_amd/libstrive_hip.so`` built by hipcc for gfx950
(strive_amd/build.py).  If it is missing or cannot be loaded every HIP-backed operator raises
``StriveHipError`` -- there is no CPU fallback.  ``StriveLib(path)`` is also instantiated by the
CPU-side tests with the host-emulation build of the same sources (tests/hipemu); the product never
does that.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, 'libstrive_hip.so')

MAXL = 4


class StriveHipError(RuntimeError):
    pass


fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
dp = C.POINTER(C.c_double)


class StriveMLP(C.Structure):
    _fields_ = [('nlayers', C.c_int32), ('dims', C.c_int32 * (MAXL + 1)),
                ('w', C.c_void_p * MAXL), ('wt', C.c_void_p * MAXL), ('b', C.c_void_p * MAXL),
                ('ln_g', C.c_void_p * MAXL), ('ln_b', C.c_void_p * MAXL),
                ('wf', C.c_void_p * MAXL), ('wbf', C.c_void_p * MAXL), ('wsc', C.c_float * MAXL)]


class StriveGNN(C.Structure):
    _fields_ = [('mlp_in', StriveMLP), ('edge', StriveMLP), ('update', StriveMLP), ('mlp_out', StriveMLP),
                ('D', C.c_int32), ('NC', C.c_int32)]


class StriveGRU(C.Structure):
    _fields_ = [('wih', C.c_void_p * 3), ('whh', C.c_void_p * 3), ('wih_t', C.c_void_p * 3),
                ('whh_t', C.c_void_p * 3), ('bih', C.c_void_p * 3), ('bhh', C.c_void_p * 3),
                ('whh_f', C.c_void_p * 3), ('wih_f', C.c_void_p * 3), ('whh_bf', C.c_void_p * 3), ('wih_bf', C.c_void_p * 3),
                ('hh_sc', C.c_float * 3), ('ih_sc', C.c_float * 3)]


class StriveMap(C.Structure):
    _fields_ = [('raster', C.c_void_p), ('dx', C.c_void_p), ('M', C.c_int32), ('C', C.c_int32),
                ('H', C.c_int32), ('W', C.c_int32), ('lwise', C.c_void_p), ('wwise', C.c_void_p),
                ('L', C.c_int32), ('Wc', C.c_int32), ('raster_px4', C.c_void_p)]


class StriveCNN(C.Structure):
    _fields_ = [('w', C.c_void_p * 6), ('b', C.c_void_p * 6), ('gn_g', C.c_void_p * 6), ('gn_b', C.c_void_p * 6),
                ('fc_wt', C.c_void_p), ('fc_b', C.c_void_p), ('w1_frag', C.c_void_p),
                ('w2_frag', C.c_void_p), ('w3_frag', C.c_void_p), ('w4_frag', C.c_void_p), ('w5_frag', C.c_void_p), ('w6_frag', C.c_void_p),
                ('w_torch', C.c_void_p * 6), ('wscale', C.c_float * 6), ('xscale', C.c_float * 6), ('conv2_plain', C.c_int32)]


class StriveRasterJob(C.Structure):
    _fields_ = [('verts', C.c_void_p), ('ring_ptr', C.c_void_p), ('shape_ptr', C.c_void_p), ('shape_kind', C.c_void_p),
                ('tile_ptr', C.c_void_p), ('tile_shapes', C.c_void_p), ('H', C.c_int32), ('W', C.c_int32), ('out_pitch', C.c_int32),
                ('flip_rows', C.c_int32), ('dx_x', C.c_double), ('dx_y', C.c_double), ('half_width', C.c_double)]


class StriveScenes(C.Structure):
    _fields_ = [('NA', C.c_int32), ('NS', C.c_int32), ('B', C.c_int32), ('max_n', C.c_int32), ('n_edges', C.c_int64),
                ('ptr', C.c_void_p), ('scene_of', C.c_void_p)]


class StriveAvoidColl(C.Structure):
    _fields_ = [('pair_off', C.c_void_p), ('P', C.c_int32), ('cent_x', C.c_void_p), ('rad', C.c_void_p), ('buffer', C.c_float),
                ('pair_valid', C.c_void_p), ('i0', C.c_void_p), ('i1', C.c_void_p), ('w0', C.c_void_p), ('w1', C.c_void_p),
                ('scale', C.c_int32), ('NE', C.c_int32), ('env_agent', C.c_void_p), ('env_of_agent', C.c_void_p),
                ('env_lw', C.c_void_p), ('env_mapix', C.c_void_p), ('env_pdist', C.c_void_p), ('gl', C.c_int32),
                ('gw', C.c_int32), ('lin_l', C.c_void_p), ('lin_w', C.c_void_p), ('init_z', C.c_void_p), ('NZ', C.c_int32), ('D', C.c_int32),
                ('prior_den', C.c_float), ('init_den', C.c_float),
                ('w_veh', C.c_float), ('w_env', C.c_float), ('w_prior', C.c_float), ('w_init', C.c_float)]


class StriveAdvGen(C.Structure):
    _fields_ = [('base', StriveAvoidColl), ('ne_ptr', C.c_void_p), ('slot_ne', C.c_void_p), ('atk_mask', C.c_void_p),
                ('scene_alive', C.c_void_p), ('t0', C.c_int32), ('use_infront', C.c_int32), ('infront', C.c_float), ('w_crash', C.c_float),
                ('w_plan', C.c_float), ('w_prior_atk', C.c_float), ('w_init_atk', C.c_float)]


class StriveDecoder(C.Structure):
    _fields_ = [('gnn', StriveGNN), ('gru', StriveGRU), ('cnn', StriveCNN), ('map', StriveMap),
                ('state_mean', C.c_float * 6), ('state_std', C.c_float * 6),
                ('att_mean', C.c_float * 2), ('att_std', C.c_float * 2),
                ('a_mean', C.c_float), ('a_std', C.c_float), ('ddh_mean', C.c_float), ('ddh_std', C.c_float),
                ('dt', C.c_float), ('max_hdot', C.c_float), ('max_s', C.c_float), ('scene_par', C.c_void_p)]


class StriveLaneNode(C.Structure):
    _fields_ = [('n', C.c_int32), ('node', C.c_int32 * 4), ('pad', C.c_int32 * 3), ('len', C.c_double * 4)]


class StrivePlannerMap(C.Structure):
    _fields_ = [('xy', C.c_void_p), ('succ', C.c_void_p), ('pred', C.c_void_p), ('succ_ptr', C.c_void_p), ('succ_idx', C.c_void_p),
                ('succ_len', C.c_void_p), ('pred_ptr', C.c_void_p), ('pred_idx', C.c_void_p), ('pred_len', C.c_void_p),
                ('edges', C.c_void_p), ('edge_ix', C.c_void_p), ('cell_ptr', C.c_void_p), ('cell_edges', C.c_void_p),
                ('N', C.c_int32), ('M', C.c_int32), ('gnx', C.c_int32), ('gny', C.c_int32),
                ('gx0', C.c_double), ('gy0', C.c_double), ('gcell', C.c_double)]


class StrivePlannerCfg(C.Structure):
    _fields_ = [('dt', C.c_double), ('preddt', C.c_double), ('xydistmax', C.c_double), ('smax', C.c_double), ('accmax', C.c_double),
                ('interacdist', C.c_double), ('col_plim', C.c_double), ('score_wmin', C.c_double), ('score_wfac', C.c_double),
                ('cdistmax', C.c_double), ('tmax', C.c_double),
                ('predsfacs', C.c_double * 4), ('predafacs', C.c_double * 4), ('planaccfacs', C.c_double * 4),
                ('nsteps', C.c_int32), ('npredsfacs', C.c_int32), ('npredafacs', C.c_int32), ('nplanaccfacs', C.c_int32),
                ('plannspeeds', C.c_int32)]


class StrivePlanner(C.Structure):
    _fields_ = [('cfg', StrivePlannerCfg), ('nmaps', C.c_int32), ('maps', StrivePlannerMap * 4),
                ('B', C.c_int32), ('NO', C.c_int32), ('NR', C.c_int32), ('ego_idx', C.c_int32),
                ('ptr', C.c_void_p), ('scene_map', C.c_void_p), ('init', C.c_void_p), ('row_obj', C.c_void_p), ('row_scene', C.c_void_p)]


P = C.c_void_p
I = C.c_int32
SZ = C.c_size_t
F4 = C.c_float * 4

# name -> (restype, argtypes).  Every symbol declared in include/strive_hip.h is listed here and
# tests/test_abi.py checks the two stay in sync.
PROTOTYPES = {
    'strive_abi_version': (C.c_int, []),
    'strive_last_error': (C.c_char_p, []),
    'strive_set_option': (C.c_int, [C.c_char_p, C.c_int64]),
    'strive_get_option': (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    'strive_reset_options': (C.c_int, []),
    'strive_option_count': (C.c_int32, []),
    'strive_option_name': (C.c_char_p, [C.c_int32]),
    'strive_map_crop_u8': (C.c_int, [C.POINTER(StriveMap), P, F4, F4, P, I, P, P]),
    'strive_map_rasterize': (C.c_int, [C.POINTER(StriveRasterJob), P, P]),
    'strive_coll_point': (C.c_int, [C.POINTER(StriveMap), P, P, P, I, I, I, P, P, P, P, P]),
    'strive_map_cnn_workspace_bytes': (SZ, [I]),
    'strive_map_cnn_fwd': (C.c_int, [C.POINTER(StriveMap), C.POINTER(StriveCNN), P, F4, F4, P, I, P, P, SZ, P]),
    'strive_map_cnn_fwd_from_crop': (C.c_int, [C.POINTER(StriveCNN), P, I, P, P, SZ, P]),
    'strive_map_cnn_bench_layer': (C.c_int, [C.POINTER(StriveMap), C.POINTER(StriveCNN), I, P, F4, F4, P, I, P, P, SZ, P]),
    'strive_mlp_fwd': (C.c_int, [C.POINTER(StriveMLP), P, I, P, P]),
    'strive_gnn_workspace_bytes': (SZ, [C.POINTER(StriveGNN), C.POINTER(StriveScenes)]),
    'strive_gnn_fwd': (C.c_int, [C.POINTER(StriveGNN), C.POINTER(StriveScenes), P, P, P, P, P, SZ, P]),
    'strive_rollout_scene_resident': (C.c_int, [C.POINTER(StriveDecoder), C.POINTER(StriveScenes)]),
    'strive_rollout_tape_bytes': (SZ, [C.POINTER(StriveDecoder), C.POINTER(StriveScenes), I]),
    'strive_rollout_workspace_bytes': (SZ, [C.POINTER(StriveDecoder), C.POINTER(StriveScenes), I]),
    'strive_rollout_fwd': (C.c_int, [C.POINTER(StriveDecoder), C.POINTER(StriveScenes), P, P, P, P, P, P, P, P, I,
                                     P, P, SZ, P, SZ, P]),
    'strive_rollout_bwd': (C.c_int, [C.POINTER(StriveDecoder), C.POINTER(StriveScenes), P, P, P, P, I, P, P,
                                     P, SZ, P, SZ, P]),
    'strive_veh_coll_fwd': (C.c_int, [C.POINTER(StriveScenes), P, I, P, I, P, P, C.c_float, P, P, P, P]),
    'strive_interp_traj_fwd': (C.c_int, [P, I, I, I, P, P, P, P, P, P]),
    'strive_interp_traj_bwd': (C.c_int, [P, P, I, I, I, I, P, P, P, P, P, P]),
    'strive_rect_iou': (C.c_int, [P, P, P, P, I, P, P]),
    'strive_veh_coll_bwd': (C.c_int, [C.POINTER(StriveScenes), P, I, P, I, P, P, C.c_float, P, P, P, P]),
    'strive_avoid_coll_workspace_bytes': (SZ, [C.POINTER(StriveScenes), C.POINTER(StriveAvoidColl), I]),
    'strive_avoid_coll_fwd': (C.c_int, [C.POINTER(StriveScenes), C.POINTER(StriveMap), C.POINTER(StriveAvoidColl), P, I, P, P, P,
                                        P, P, SZ, P]),
    'strive_avoid_coll_bwd': (C.c_int, [C.POINTER(StriveScenes), C.POINTER(StriveAvoidColl), P, I, P, P, P, P, P, SZ, P, P, P]),
    'strive_adv_gen_workspace_bytes': (SZ, [C.POINTER(StriveScenes), C.POINTER(StriveAdvGen), I]),
    'strive_adv_gen_fwd': (C.c_int, [C.POINTER(StriveScenes), C.POINTER(StriveMap), C.POINTER(StriveAdvGen), P, P, I, P, P, P, P, P,
                                     P, P, SZ, P]),
    'strive_adv_gen_bwd': (C.c_int, [C.POINTER(StriveScenes), C.POINTER(StriveAdvGen), P, P, I, P, P, P, P, P, SZ, P, P, P, P]),
    'strive_coll_point_rows': (C.c_int, [C.POINTER(StriveMap), P, I, P, P, P, I, I, I, P, P, P, P, P]),
    'strive_mlp_param_count': (SZ, [C.POINTER(StriveMLP)]),
    'strive_gnn_param_count': (SZ, [C.POINTER(StriveGNN)]),
    'strive_gru_param_count': (SZ, []),
    'strive_map_cnn_param_count': (SZ, []),
    'strive_mlp_bwd': (C.c_int, [C.POINTER(StriveMLP), P, P, I, P, P, P]),
    'strive_gnn_bwd_workspace_bytes': (SZ, [C.POINTER(StriveGNN), C.POINTER(StriveScenes)]),
    'strive_pack_dense': (C.c_int, [P, C.c_int32, C.c_int32, C.c_float, P, P, P, P]),
    'strive_pack_split_gather': (C.c_int, [P, I, P, I, C.c_float, P, P]),
    'strive_gnn_bwd': (C.c_int, [C.POINTER(StriveGNN), C.POINTER(StriveScenes), P, P, P, P, P, P, P, SZ, P]),
    'strive_map_cnn_bwd_workspace_bytes': (SZ, [I]),
    'strive_map_cnn_bwd': (C.c_int, [C.POINTER(StriveMap), C.POINTER(StriveCNN), P, F4, F4, P, I, P, P, P, SZ, P]),
    'strive_map_cnn_bwd_bench_dgrad': (C.c_int, [I, I, P, SZ, P]),
    'strive_rollout_train_workspace_bytes': (SZ, [C.POINTER(StriveDecoder), C.POINTER(StriveScenes), I]),
    'strive_rollout_bwd_train': (C.c_int, [C.POINTER(StriveDecoder), C.POINTER(StriveScenes), P, P, P, P, P, I, P, P, P, P, P, P, P,
                                           P, SZ, P, SZ, P]),
    'strive_map_cnn_keep_bytes': (SZ, [I]),
    'strive_map_cnn_fwd_keep': (C.c_int, [C.POINTER(StriveMap), C.POINTER(StriveCNN), P, F4, F4, P, I, P, P, SZ, P, SZ, I, I, P]),
    'strive_map_cnn_bwd_kept': (C.c_int, [C.POINTER(StriveMap), C.POINTER(StriveCNN), P, F4, F4, P, I, P, P, P, SZ, P, SZ, P]),
    'strive_map_cnn_bwd_kept_range': (C.c_int, [C.POINTER(StriveMap), C.POINTER(StriveCNN), P, F4, F4, P, I, P, P, P, SZ, I, I, P, SZ, P]),
    'strive_rollout_keep_bytes': (SZ, [C.POINTER(StriveDecoder), C.POINTER(StriveScenes), I]),
    'strive_rollout_fwd_keep': (C.c_int, [C.POINTER(StriveDecoder), C.POINTER(StriveScenes), P, P, P, P, P, P, P, P, I,
                                          P, P, SZ, P, SZ, P, SZ, P]),
    'strive_rollout_bwd_train_kept': (C.c_int, [C.POINTER(StriveDecoder), C.POINTER(StriveScenes), P, P, P, P, P, I, P, P, P, P, P, P,
                                                P, P, SZ, P, SZ, P, SZ, P]),
    'strive_bicycle_step': (C.c_int, [C.POINTER(StriveDecoder), P, P, P, P, P, P, P, I, P]),
    'strive_rel_pose': (C.c_int, [P, P, P, P, P, P, I, I, P]),
    'strive_planner_workspace_bytes': (SZ, [C.POINTER(StrivePlanner), I, I]),
    'strive_planner_rollout': (C.c_int, [C.POINTER(StrivePlanner), P, P, I, P, I, P, I, I, P, P, P, P, SZ, P]),
    'strive_planner_routes': (C.c_int, [C.POINTER(StrivePlanner), I, P, I, I, P, P, P, P, P]),
}


ABI_VERSION = 17   # include/strive_hip.h STRIVE_ABI_VERSION


_instances = []      # every loaded library (the product's; the tests' host-emulated build): sync_all_options_from_env() walks them


def sync_all_options_from_env():
    for lib in _instances:
        lib.sync_options_from_env()


class StriveLib(object):
    def __init__(self, path=None, require_all=True):
        self.path = DEFAULT_LIB if path is None else path
        if not os.path.exists(self.path):
            raise StriveHipError('HIP library not found at %s -- run `python -m strive_amd.build` '
                                 '(hipcc, gfx950); there is no CPU fallback' % self.path)
        try:
            self.cdll = C.CDLL(self.path)
        except OSError as e:
            raise StriveHipError('cannot load %s: %s' % (self.path, e))
        self.missing = []
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError:
                self.missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
            setattr(self, '_' + name, fn)
        if require_all and self.missing:
            raise StriveHipError('%s lacks symbols: %s' % (self.path, ', '.join(self.missing)))
        if not self.missing or 'strive_abi_version' not in self.missing:
            v = self._strive_abi_version()
            if v != ABI_VERSION:
                raise StriveHipError('ABI version mismatch: library %d, binding %d' % (v, ABI_VERSION))
        if 'strive_set_option' not in self.missing:
            self.sync_options_from_env()
            _instances.append(self)

    def call(self, name, *args):
        fn = getattr(self, '_' + name, None)
        if fn is None:
            raise StriveHipError('symbol %s missing from %s' % (name, self.path))
        rc = fn(*args)
        if rc != 0:
            raise StriveHipError('%s failed (%d): %s' % (name, rc, self._strive_last_error().decode()))

    def query(self, name, *args):
        return getattr(self, '_' + name)(*args)

    # ---- options (include/strive_hip.h, ABI 17): the library reads no environment variable; this host maps STRIVE_<NAME> onto them ----
    def option_names(self):
        return [self._strive_option_name(i).decode() for i in range(self._strive_option_count())]

    def set_option(self, name, value):
        if self._strive_set_option(name.encode(), int(value)) != 0:
            raise StriveHipError(self._strive_last_error().decode())

    def get_option(self, name):
        v = C.c_int64(0)
        if self._strive_get_option(name.encode(), C.byref(v)) != 0:
            raise StriveHipError(self._strive_last_error().decode())
        return int(v.value)

    def sync_options_from_env(self, environ=None):
        """Every option back to its default, then STRIVE_<NAME>=<int> from the environment on top (A/B runs, tests).  Called when the
        library is loaded; call it again after changing such a variable inside the process."""
        environ = os.environ if environ is None else environ
        self._strive_reset_options()
        for name in self.option_names():
            v = environ.get('STRIVE_' + name.upper())
            if v is not None and v != '':
                self.set_option(name, int(v))


_default = None


def get_lib():
    """The product's single library instance (hipcc build).  Raises StriveHipError if unavailable."""
    global _default
    if _default is None:
        _default = StriveLib(DEFAULT_LIB)
    return _default


class _TensorArg(object):
    """ctypes argument that keeps its tensor alive until the foreign call has been issued (temporaries such as
    ``x.contiguous()`` would otherwise be freed between taking the address and making the call)."""
    __slots__ = ('tensor', '_as_parameter_')

    def __init__(self, t):
        self.tensor = t
        self._as_parameter_ = C.c_void_p(t.data_ptr())


def ptr(t):
    """Device (or, under the test emulation, host) address of a contiguous tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), 'non-contiguous tensor passed to the C ABI'
    return _TensorArg(t)


def f4(vals):
    return F4(*[float(v) for v in vals])


def stream_ptr(t):
    """hipStream_t of torch's current stream on t's device (NULL stream for host tensors)."""
    import torch
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)
