"""Drop-in check of the boundary in the BUILD CONTAINER: the reference's OWN loop functions
(/root/reference/src/utils/{init,adv_gen,sol}_optim.py and refine_traffic_optim.py's closure structure), imported from the
reference checkout through ``strive_amd.dropin.install(reference_src=..., loops='reference')``, run unchanged on top of this
package's TrafficModel and loss modules.  No GPU here, so the C ABI behind those modules is the host-emulation build of the
same .hip sources (tests/hipemu) -- this test is about names, signatures, dict keys, tensor shapes, autograd plumbing and
the numbers coming out, not about speed.  Skipped where /root/reference does not exist (the GPU box)."""
import contextlib
import io
import os
import sys
import types

import numpy as np
import pytest
import torch

REF_SRC = '/root/reference/src'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason='the reference checkout only exists in the build container')

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hipemu'))


def _devkit_stubs():
    """nuscenes-devkit / pyquaternion / tqdm-free import stand-ins: the reference's datasets/nuscenes_utils.py imports the
    devkit at module level; nothing of it is called on this path."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules.setdefault(name, m)

    class _Empty(object):
        def __init__(self, *a, **k):
            pass
    mod('nuscenes')
    mod('nuscenes.map_expansion')
    mod('nuscenes.map_expansion.map_api', NuScenesMap=_Empty)
    mod('nuscenes.map_expansion.arcline_path_utils', discretize_lane=lambda *a, **k: None)
    mod('nuscenes.nuscenes', NuScenes=_Empty)
    mod('nuscenes.utils')
    mod('nuscenes.utils.splits', create_splits_scenes=lambda *a, **k: {})
    mod('pyquaternion', Quaternion=_Empty)
    for alias, typ in (('int', int), ('bool', bool), ('float', float)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)


@pytest.fixture(scope='module')
def dropin_env():
    import build as emu_build
    from strive_amd import _lib as L, ops, dropin
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in ('models', 'losses', 'utils', 'datasets', 'torch_geometric')}
    _devkit_stubs()
    names = dropin.install(reference_src=REF_SRC, loops='reference')
    emu = L.StriveLib(emu_build.build(), require_all=True)
    orig = (ops._lib_for, L.get_lib)
    ops._lib_for = lambda *tensors: emu           # CPU tensors + the emulated library: test infrastructure only
    L.get_lib = lambda: emu
    yield names
    ops._lib_for, L.get_lib = orig
    for k in list(sys.modules):
        if k.split('.')[0] in ('models', 'losses', 'utils', 'datasets'):
            del sys.modules[k]
    sys.modules.update(saved)


def _setup(FT=2, sizes=(3,)):     # every scene needs an attacker (the reference's AdvGenLoss indexes max() over them)
    from models.traffic_model import TrafficModel                 # resolves to strive_amd's
    from datasets.utils import MeanStdNormalizer, NUSC_BIKE_PARAMS
    from strive_amd import synth
    from strive_amd.constants import state_norm_tensors, att_norm_tensors
    m = TrafficModel(4, FT, 256, 2)
    sd = synth.fill_state_dict(m.state_dict())
    m.load_state_dict(sd)
    m.set_normalizer(MeanStdNormalizer(*state_norm_tensors()))
    m.set_att_normalizer(MeanStdNormalizer(*att_norm_tensors()))
    m.set_bicycle_params(NUSC_BIKE_PARAMS)
    m.eval()
    batch, map_idx = synth.make_batch(list(sizes), key='dropin', FT=FT)
    raster, dx = synth.make_raster(1024, 1024)
    env = synth.SyntheticMapEnv(raster, dx)
    return m, sd, batch, map_idx, env


def test_names_resolve_to_the_right_owner(dropin_env):
    import models.traffic_model, losses.adv_gen_nusc, utils.adv_gen_optim, utils.sol_optim, utils.init_optim   # noqa: E401
    import utils.logger, utils.scenario_gen, datasets.nuscenes_utils, datasets.utils                          # noqa: E401
    assert models.traffic_model.__name__ == 'strive_amd.models.traffic_model'
    assert losses.adv_gen_nusc.__name__ == 'strive_amd.losses.adv_gen_nusc'
    for mod in (utils.adv_gen_optim, utils.sol_optim, utils.init_optim, utils.logger):
        assert mod.__file__.startswith(REF_SRC), mod.__file__                     # the reference's own files
    assert utils.scenario_gen.__file__.startswith(REF_SRC)
    assert utils.scenario_gen.detach_embed_info.__module__ == 'strive_amd.utils.scenario_gen'     # ... with ours laid over
    assert utils.scenario_gen.log_metric.__module__ == 'utils.scenario_gen'                        # ... and theirs kept
    assert datasets.nuscenes_utils.get_map_obs.__module__ == 'strive_amd.datasets.nuscenes_utils'
    assert datasets.utils.MeanStdNormalizer.__module__ == 'strive_amd.datasets.utils'
    assert set(dropin_env) >= {'models.traffic_model', 'losses.adv_gen_nusc', 'utils.adv_gen_optim'}


def _oracle(sd, FT):
    from util import oracle_model
    return oracle_model(sd, FT=FT)


def _weights():
    import make_golden as mg
    return mg.LOOP_WEIGHTS


def test_reference_adv_loop_runs_on_the_hip_modules(dropin_env):
    """The reference's run_adv_gen_optim ('ego' planner), unchanged, one iteration on 3 agents with FT = 2 -- and the same
    numbers as the oracle's restatement of that loop."""
    import utils.adv_gen_optim as ref_adv
    from utils.scenario_gen import detach_embed_info
    from oracle import loops as oloops
    from util import assert_close
    m, sd, batch, map_idx, env = _setup()
    with torch.no_grad():
        emb = detach_embed_info(m.embed(batch, map_idx, env))
    NA = batch.past.shape[0]
    ego = torch.zeros((NA,), dtype=torch.bool)
    ego[batch.ptr[:-1]] = True
    pm, pv = emb['prior_out']
    tp, op = (pm[ego], pv[ego]), (pm[~ego], pv[~ego])
    z0 = emb['posterior_out'][0].clone()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        z, fin, out, agt, tt = ref_adv.run_adv_gen_optim(z0, 0.05, _weights(), m, batch, env, map_idx, 1, emb, 'ego', tp, op, 0, 0.0)
    assert z.shape == (NA, 32) and fin.shape == (NA, 1, 2, 4) and len(agt) == 1 and torch.isfinite(z).all()
    orc = _oracle(sd, 2)
    with torch.no_grad():
        emb_o = orc.embed(batch, map_idx, env)
    assert_close(emb['map_feat'], emb_o['map_feat'], 1e-4, 2e-5, 'map_feat')
    z_o = oloops.adv_loop(orc, batch, map_idx, env, emb_o, emb_o['posterior_out'][0].clone(), _weights(), 1, 0.05,
                          (emb_o['prior_out'][0][ego], emb_o['prior_out'][1][ego]),
                          (emb_o['prior_out'][0][~ego], emb_o['prior_out'][1][~ego]), feasibility_time=0, feasibility_infront_min=0.0)
    # one Adam step moves every entry by ~lr in the direction of its gradient sign: agreement means the two-rollout
    # complementary-detach closure, the losses and the backward produced the same gradient signs
    assert float(((z.detach() - z0) * (z_o - z0) > 0).float().mean()) > 0.98
    assert_close(z.detach(), z_o, 0, 0.11, 'z after one reference iteration')


@pytest.mark.slow
def test_reference_init_sol_and_refine_run_on_the_hip_modules(dropin_env):
    import importlib.util
    import utils.init_optim as ref_init
    import utils.sol_optim as ref_sol
    from utils.scenario_gen import detach_embed_info
    m, sd, batch, map_idx, env = _setup()
    with torch.no_grad():
        emb = detach_embed_info(m.embed(batch, map_idx, env))
    NA = batch.past.shape[0]
    ego = torch.zeros((NA,), dtype=torch.bool)
    ego[batch.ptr[:-1]] = True
    pm, pv = emb['prior_out']
    tp, op = (pm[ego], pv[ego]), (pm[~ego], pv[~ego])
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
        z1, traj1, _ = ref_init.run_init_optim(emb['posterior_out'][0].clone(), batch.future_gt[:, :, :4], batch.future_vis, 0.05,
                                               _weights(), m, batch, env, map_idx, 1, emb, emb['prior_out'])
        fin = traj1.unsqueeze(1)
        z3, sol, _ = ref_sol.run_find_solution_optim(z1.detach(), fin, 2, 0.05, _weights(), m, batch, env, map_idx, 1, emb, tp, op)
    assert traj1.shape == (NA, 2, 4) and sol.shape == (NA, 1, 2, 4) and torch.isfinite(z3).all()
    # refine_traffic_optim(): the reference's function body, loaded from its file without running the script's imports of
    # the dataset / viz stack (only the function's own globals are needed)
    src = open(os.path.join(REF_SRC, 'refine_traffic_optim.py')).read()
    start = src.index('def refine_traffic_optim(')
    end = src.index('def run_one_epoch(')
    ns = {'torch': torch, 'optim': torch.optim, 'tqdm': __import__('tqdm'),
          'detach_embed_info': detach_embed_info, 'AvoidCollLoss': sys.modules['losses.adv_gen_nusc'].AvoidCollLoss}
    exec(compile(src[start:end], 'reference:refine_traffic_optim.py', 'exec'), ns)
    import make_golden as mg
    with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
        init_pred, z, res, emb2 = ns['refine_traffic_optim'](batch, map_idx, env, m, mg.REFINE_WEIGHTS, 1, 2, 2, True, 0.05)
    assert init_pred.shape == (NA, 2, 4) and res.shape == (NA, 1, 2, 4) and torch.isfinite(z).all()
