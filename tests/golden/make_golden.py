#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Only runs in the build container, where the reference is mounted read-only at /root/reference
(it does not exist on the GPU box, and none of its source is copied here -- the committed
fixtures are data: inputs are regenerated from strive_amd.synth's counter-based generator and only
the reference's outputs are stored).

The reference imports three third-party packages that are not installed (torch_geometric,
nuscenes-devkit, pyquaternion).  They are replaced by the import stand-ins below:
  * nuscenes / pyquaternion: empty symbols, needed only so module-level imports succeed;
  * torch_geometric.nn.MessagePassing: restates the published semantics of PyG 1.7.1's
    ``propagate`` for ``flow='source_to_target'``, ``aggr='max'`` (gather ``*_j`` with
    ``edge_index[0]``, ``*_i`` with ``edge_index[1]`` on dim -2, call ``message``, scatter-max
    by target with 0 for empty rows, call ``update``);
  * torch_geometric.data: the Data/Batch containers of strive_amd.graph.
numpy aliases removed in numpy>=1.24 (np.int/np.bool/np.float) are restored for the reference.

Usage:  python tests/golden/make_golden.py   (writes tests/golden/*.npz)
"""
import inspect
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
REF_SRC = '/root/reference/src'

sys.path.insert(0, REPO)
from strive_amd import synth                                    # noqa: E402
from strive_amd.graph import Data, Batch                        # noqa: E402
from strive_amd.constants import NUSC_BIKE_PARAMS, state_norm_tensors, att_norm_tensors  # noqa: E402


# ------------------------------------------------------------------------------------------
# import stand-ins
# ------------------------------------------------------------------------------------------

def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

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

    class MessagePassing(torch.nn.Module):
        def __init__(self, aggr='add', flow='source_to_target', node_dim=-2):
            super().__init__()
            assert aggr == 'max' and flow == 'source_to_target'
            self._msg_args = [p for p in inspect.signature(self.message).parameters]
            self._upd_args = [p for p in inspect.signature(self.update).parameters][1:]

        def propagate(self, edge_index, **kw):
            src, dst = edge_index[0], edge_index[1]
            margs = {}
            for name in self._msg_args:
                base, suffix = name[:-2], name[-2:]
                v = kw.get(base)
                if v is None or not torch.is_tensor(v):
                    margs[name] = None
                else:
                    margs[name] = v.index_select(-2, dst if suffix == '_i' else src)
            msg = self.message(**margs)
            n = kw['x'].shape[0]
            out = torch.zeros((n, msg.shape[1]), dtype=msg.dtype, device=msg.device)
            if msg.shape[0] > 0:
                out = out.scatter_reduce(0, dst.view(-1, 1).expand_as(msg), msg, reduce='amax', include_self=False)
            return self.update(out, **{k: kw.get(k) for k in self._upd_args})

    class DataLoader(object):
        def __init__(self, *a, **k):
            raise RuntimeError('not available in the golden harness')

    mod('torch_geometric')
    mod('torch_geometric.nn', MessagePassing=MessagePassing)
    mod('torch_geometric.data', Data=Data, Batch=Batch, DataLoader=DataLoader)
    mod('torch_geometric.utils')
    for alias, typ in (('int', int), ('bool', bool), ('float', float)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)
    if not hasattr(np, 'product'):          # removed in numpy 2 (the planner calls it)
        np.product = np.prod


def import_reference():
    _install_stubs()
    if sys.path[0] != REF_SRC:
        sys.path.insert(0, REF_SRC)       # must beat the installed HF 'datasets' package
    for name in list(sys.modules):
        if name == 'datasets' or name.startswith('datasets.'):
            del sys.modules[name]
    import importlib
    R = types.SimpleNamespace()
    R.traffic_model = importlib.import_module('models.traffic_model')
    R.interaction_net = importlib.import_module('models.interaction_net')
    R.common = importlib.import_module('models.common')
    R.transforms = importlib.import_module('utils.transforms')
    R.nutils = importlib.import_module('datasets.nuscenes_utils')
    R.dutils = importlib.import_module('datasets.utils')
    R.map_env = importlib.import_module('datasets.map_env')
    R.adv_losses = importlib.import_module('losses.adv_gen_nusc')
    R.tm_losses = importlib.import_module('losses.traffic_model')
    R.loss_common = importlib.import_module('losses.common')
    R.scenario_gen = importlib.import_module('utils.scenario_gen')
    R.init_optim = importlib.import_module('utils.init_optim')
    R.adv_optim = importlib.import_module('utils.adv_gen_optim')
    R.sol_optim = importlib.import_module('utils.sol_optim')
    R.planner_base = importlib.import_module('planners.planner')
    R.planner = importlib.import_module('planners.hardcode_goalcond_nusc')
    return R


# ------------------------------------------------------------------------------------------
# shared fixtures (also imported by the tests to regenerate inputs)
# ------------------------------------------------------------------------------------------

RASTER_HW = 1024


def build_inputs(sizes, key, FT=12, NC=2, M=1, window=None):
    """(batch, map_idx, raster, dx).  window=None keeps the default 120 m spread."""
    if window is None:
        batch, map_idx = synth.make_batch(sizes, key=key, FT=FT, NC=NC, M=M)
    else:
        scenes = [synth.make_scene(n, '%s/%d' % (key, b), FT=FT, NC=NC, window=window) for b, n in enumerate(sizes)]
        batch = Batch.from_data_list(scenes)
        map_idx = torch.tensor([b % M for b in range(len(sizes))], dtype=torch.long)
    raster, dx = synth.make_raster(RASTER_HW, RASTER_HW, M=M)
    return batch, map_idx, raster, dx


def ref_map_env(R, raster, dx):
    env = object.__new__(R.map_env.NuScenesMapEnv)
    env.nusc_raster = raster
    env.nusc_dx = dx
    env.bounds = [-17.0, -38.5, 60.0, 38.5]
    env.L = 256
    env.W = 256
    env.map_list = ['synthetic-%d' % i for i in range(raster.shape[0])]
    env.num_layers = raster.shape[1]
    env.device = torch.device('cpu')
    return env


def ref_model(R, NC=2, FT=12, key='weights'):
    m = R.traffic_model.TrafficModel(4, FT, 256, NC)
    sd = synth.fill_state_dict(m.state_dict(), key=key)
    m.load_state_dict(sd)
    m.set_normalizer(R.dutils.MeanStdNormalizer(*state_norm_tensors()))
    m.set_att_normalizer(R.dutils.MeanStdNormalizer(*att_norm_tensors()))
    m.set_bicycle_params(NUSC_BIKE_PARAMS)
    m.eval()
    return m, sd


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print('wrote %s (%.1f KB)' % (name, os.path.getsize(path) / 1024.0))


# ------------------------------------------------------------------------------------------
# fixture generators
# ------------------------------------------------------------------------------------------

def g1_ops(R):
    out = {}
    frame = synth.f32(synth.counter_uniform((7, 4), 'g1/frame', -2.0, 2.0))
    poses = synth.f32(synth.counter_uniform((7, 5, 4), 'g1/poses', -3.0, 3.0))
    out['t2f_fwd'] = npy(R.transforms.transform2frame(frame, poses))
    out['t2f_inv'] = npy(R.transforms.transform2frame(frame, poses, inverse=True))
    # bicycle step incl. clamp-active rows
    st = synth.f32(synth.counter_uniform((9, 6), 'g1/state', -1.0, 1.0))
    st[:, 0:2] *= 100.0
    st[:, 4] = torch.tensor([0.0, 0.2, 3.0, 49.9, 12.0, 0.01, 7.0, 25.0, 1.0])
    st[:, 5] = torch.tensor([0.0, 6.2, -6.2, 0.1, -0.1, 0.3, 0.0, 1.0, -1.0])
    a = synth.f32(synth.counter_uniform((9,), 'g1/a', -4.0, 4.0))
    a[0] = -3.0
    a[3] = 4.0
    ddh = synth.f32(synth.counter_uniform((9,), 'g1/ddh', -0.5, 0.5))
    ddh[1] = 0.5
    ddh[2] = -0.5
    vlen = synth.f32(synth.counter_uniform((9,), 'g1/len', 3.5, 6.0))
    tm, _ = ref_model(R)
    sim = tm.sim_traj(st.unsqueeze(1), a.view(9, 1, 1), ddh.view(9, 1, 1), vlen.view(9, 1))[:, 0, 0]
    out['bicycle'] = npy(sim)
    # MLP + GRU one step with the model's own weights
    x = synth.f32(synth.counter_uniform((6, 38), 'g1/mlp_in', -1.0, 1.0))
    out['mlp_past_encoder'] = npy(tm.past_encoder(x))
    xi = synth.f32(synth.counter_uniform((6, 1, 4), 'g1/gru_x', -1.0, 1.0))
    h0 = synth.f32(synth.counter_uniform((3, 6, 64), 'g1/gru_h', -1.0, 1.0))
    o, h1 = tm.decoder_memory(xi, h0)
    out['gru_out'] = npy(o[:, 0])
    out['gru_h'] = npy(h1)
    tr = synth.f32(synth.counter_uniform((5, 12, 4), 'g1/traj', -2.0, 2.0))
    out['interp'] = npy(R.adv_losses.interp_traj(tr, scale_factor=3))
    zz = synth.f32(synth.counter_uniform((5, 32), 'g1/z', -2.0, 2.0))
    mu = synth.f32(synth.counter_uniform((5, 32), 'g1/mu', -1.0, 1.0))
    var = synth.f32(synth.counter_uniform((5, 32), 'g1/var', 0.2, 2.0))
    var2 = synth.f32(synth.counter_uniform((5, 32), 'g1/var2', 0.2, 2.0))
    out['log_normal'] = npy(R.loss_common.log_normal(zz, mu, var))
    out['kl_normal'] = npy(R.loss_common.kl_normal(zz, var2, mu, var))
    save('g1_ops.npz', **out)


def g2_inputs():
    raster, dx = synth.make_raster(RASTER_HW, RASTER_HW, M=2)
    n = 12
    frame = np.zeros((n, 4))
    frame[:, 0] = synth.counter_uniform((n,), 'g2/x', 30.0, 226.0)
    frame[:, 1] = synth.counter_uniform((n,), 'g2/y', 30.0, 226.0)
    ang = synth.counter_uniform((n,), 'g2/h', -np.pi, np.pi)
    frame[:, 2], frame[:, 3] = np.cos(ang), np.sin(ang)
    frame[1, :2] = [3.0, 4.0]          # mostly out of bounds
    frame[2, :2] = [250.0, 252.0]      # off the far corner
    frame[3, 2:] = [1.0, 0.0]          # axis aligned
    frame[4, 2:] = [0.0, 1.0]
    frame = synth.f32(frame)
    frame[5, 0] = float('nan')         # NaN frame -> samples world (0,0)
    mapixes = torch.tensor([i % 2 for i in range(n)], dtype=torch.long)
    lw = synth.f32(np.stack([synth.counter_uniform((n,), 'g2/l', 3.5, 6.5),
                             synth.counter_uniform((n,), 'g2/w', 1.6, 2.5)], -1))
    return raster, dx, frame, mapixes, lw


def g2_crop(R):
    raster, dx, frame, mapixes, lw = g2_inputs()
    bounds = [-17.0, -38.5, 60.0, 38.5]
    crop = R.nutils.get_map_obs(raster, dx, frame, mapixes, bounds, L=256, W=256)
    out = {'crop_sum': npy(crop.long().sum(dim=(2, 3))),
           'crop_rowsum': npy(crop.long().sum(dim=3)).astype(np.int16),
           'crop_colsum': npy(crop.long().sum(dim=2)).astype(np.int16),
           'crop_full_0': np.packbits(npy(crop[0])), 'crop_full_7': np.packbits(npy(crop[7])),
           'crop_full_1': np.packbits(npy(crop[1]))}
    ok = ~torch.isnan(frame[:, 0])
    pt, frac = R.nutils.get_coll_point(raster[:, 0], dx, frame[ok], lw[ok], mapixes[ok], return_iou=True)
    out['coll_pt'] = npy(pt)
    out['coll_frac'] = npy(frac)
    out['on_layer'] = npy(R.nutils.check_on_layer(raster[:, 0], dx, frame[ok], lw[ok], mapixes[ok]))
    start = frame[ok][:, :2]
    end = start + synth.f32(synth.counter_uniform((int(ok.sum()), 2), 'g2/end', -25.0, 25.0))
    inb = (start.min(dim=1)[0] > 10) & (end.min(dim=1)[0] > 10) & (start.max(dim=1)[0] < 245) & (end.max(dim=1)[0] < 245)
    out['line_sel'] = npy(inb)
    out['line_hit'] = npy(R.nutils.check_line_layer(raster[:, 0], dx, start[inb], end[inb], mapixes[ok][inb]))
    save('g2_crop.npz', **out)


G3_SIZES = [3, 5, 1]


def g3_gnn(R):
    tm, _ = ref_model(R)
    batch, map_idx, raster, dx = build_inputs(G3_SIZES, 'g3')
    NA = batch.past.shape[0]
    out = {}
    for name, net, fin in (('decoder', tm.decoder_net, 164), ('prior', tm.prior_net, 130), ('posterior', tm.posterior_net, 194)):
        x = synth.f32(synth.counter_uniform((NA, fin), 'g3/x/' + name, -1.0, 1.0)).requires_grad_(True)
        pos = batch.past[:, -1, :4].clone()
        if name == 'prior':
            pos[1, 0] = float('nan')           # NaN pose on edges -> rel pose 0
        pos.requires_grad_(True)
        batch.x, batch.pos = x, pos
        y = net(batch)
        rw = synth.f32(synth.counter_uniform(tuple(y.shape), 'g3/r/' + name, -1.0, 1.0))
        gx, gp = torch.autograd.grad((y * rw).sum(), [x, pos])
        out[name + '_out'] = npy(y)
        out[name + '_gx'] = npy(gx)
        out[name + '_gpos'] = npy(gp)
    # multi-sample path (NS=2) of the decoder net
    x = synth.f32(synth.counter_uniform((NA, 2, 164), 'g3/x/ns', -1.0, 1.0))
    pos = batch.past[:, -1, :4].unsqueeze(1).expand(NA, 2, 4) + 0.01 * synth.f32(synth.counter_uniform((NA, 2, 4), 'g3/p/ns', -1, 1))
    batch.x, batch.pos = x, pos
    out['decoder_ns_out'] = npy(tm.decoder_net(batch))
    save('g3_gnn.npz', **out)


G4_SIZES = [3, 5, 1]


def g4_rollout(R):
    out = {}
    for FT in (12, 16):
        tm, _ = ref_model(R, FT=12)
        batch, map_idx, raster, dx = build_inputs(G4_SIZES, 'g4')
        env = ref_map_env(R, raster, dx)
        with torch.no_grad():
            emb = tm.embed(batch, map_idx, env)
        emb = R.scenario_gen.detach_embed_info(emb)
        if FT == 12:
            out['map_feat'] = npy(emb['map_feat'])
            out['past_feat'] = npy(emb['past_feat'])
            out['prior_mu'] = npy(emb['prior_out'][0])
            out['prior_var'] = npy(emb['prior_out'][1])
            out['post_mu'] = npy(emb['posterior_out'][0])
            out['post_var'] = npy(emb['posterior_out'][1])
        z = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g4/z').requires_grad_(True)
        pred = tm.decode_embedding(z, emb, batch, map_idx, env, nfuture=FT)['future_pred']
        rw = synth.f32(synth.counter_uniform(tuple(pred.shape), 'g4/r%d' % FT, -1.0, 1.0))
        gz, = torch.autograd.grad((pred * rw).sum(), [z])
        out['pred_ft%d' % FT] = npy(pred)
        out['gz_ft%d' % FT] = npy(gz)
    # ext_future (ego teacher forcing) -- FT = 12
    NA = batch.past.shape[0]
    ego = batch.ptr[:-1]
    ext = batch.future_gt[ego][:, :, :4]
    z = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g4/z').requires_grad_(True)
    pred = tm.decode_embedding(z, emb, batch, map_idx, env, ext_future=ext)['future_pred']
    rw = synth.f32(synth.counter_uniform(tuple(pred.shape), 'g4/rext', -1.0, 1.0))
    gz, = torch.autograd.grad((pred * rw).sum(), [z])
    out['pred_ext'] = npy(pred)
    out['gz_ext'] = npy(gz)
    # 3-D z, NS = 2
    z3 = torch.stack([synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g4/z'),
                      synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g4/z_b')], dim=1).requires_grad_(True)
    pred = tm.decode_embedding(z3, emb, batch, map_idx, env)['future_pred']
    rw = synth.f32(synth.counter_uniform(tuple(pred.shape), 'g4/rns', -1.0, 1.0))
    gz, = torch.autograd.grad((pred * rw).sum(), [z3])
    out['pred_ns'] = npy(pred)
    out['gz_ns'] = npy(gz)
    save('g4_rollout.npz', **out)


def g4u_inputs():
    """g4's scenes over a UNIFORM raster (layer 0 = 1 everywhere): every crop is the same image, so the reference's free-running
    rollout and the product's cannot part at a crop flip -- a tight reference-direct comparison of bare rollouts."""
    batch, map_idx, _, dx = build_inputs(G4_SIZES, 'g4')
    raster = torch.zeros((1, 4, RASTER_HW, RASTER_HW), dtype=torch.uint8)
    raster[:, 0] = 1
    return batch, map_idx, raster, dx


def g4u_rollout(R):
    """The reference's decode_embedding (src/models/traffic_model.py:405-414, 626-698) over the uniform raster: nfuture 12 / 16,
    ext_future, NS = 2, each with d(sum(pred * r))/dz."""
    out = {}
    tm, _ = ref_model(R, FT=12)
    batch, map_idx, raster, dx = g4u_inputs()
    env = ref_map_env(R, raster, dx)
    with torch.no_grad():
        emb = R.scenario_gen.detach_embed_info(tm.embed(batch, map_idx, env))
    out['map_feat'] = npy(emb['map_feat'])
    out['past_feat'] = npy(emb['past_feat'])
    out['prior_mu'] = npy(emb['prior_out'][0])
    out['prior_var'] = npy(emb['prior_out'][1])
    ego = batch.ptr[:-1]
    ext = batch.future_gt[ego][:, :, :4]
    z1 = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g4/z')
    z2 = torch.stack([z1, synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g4/z_b')], dim=1)
    for name, zz, kw, rk in (('ft12', z1, {'nfuture': 12}, 'g4u/r12'), ('ft16', z1, {'nfuture': 16}, 'g4u/r16'),
                             ('ext', z1, {'ext_future': ext}, 'g4u/rext'), ('ns', z2, {}, 'g4u/rns')):
        z = zz.clone().requires_grad_(True)
        pred = tm.decode_embedding(z, emb, batch, map_idx, env, **kw)['future_pred']
        rw = synth.f32(synth.counter_uniform(tuple(pred.shape), rk, -1.0, 1.0))
        gz, = torch.autograd.grad((pred * rw).sum(), [z])
        out['pred_' + name] = npy(pred)
        out['gz_' + name] = npy(gz)
    save('g4u_rollout.npz', **out)


G4B_SIZES = [4, 2, 1]


def g4b_nc5(R):
    """NC = 5 semantic classes (reduce_cats, reference configs/adv_gen_replay_cyclist.cfg:15-16): every sem-bearing input
    of the networks grows (decoder node input 167, edge input 142).  embed + FT = 12 rollout forward and d/dz."""
    out = {}
    tm, _ = ref_model(R, NC=5, key='weights5')
    batch, map_idx, raster, dx = build_inputs(G4B_SIZES, 'g4b', NC=5)
    env = ref_map_env(R, raster, dx)
    with torch.no_grad():
        emb = R.scenario_gen.detach_embed_info(tm.embed(batch, map_idx, env))
    out['map_feat'] = npy(emb['map_feat'])
    out['past_feat'] = npy(emb['past_feat'])
    out['prior_mu'] = npy(emb['prior_out'][0])
    out['prior_var'] = npy(emb['prior_out'][1])
    out['post_mu'] = npy(emb['posterior_out'][0])
    out['post_var'] = npy(emb['posterior_out'][1])
    z = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g4b/z').requires_grad_(True)
    pred = tm.decode_embedding(z, emb, batch, map_idx, env)['future_pred']
    rw = synth.f32(synth.counter_uniform(tuple(pred.shape), 'g4b/r', -1.0, 1.0))
    gz, = torch.autograd.grad((pred * rw).sum(), [z])
    out['pred'] = npy(pred)
    out['gz'] = npy(gz)
    # the same over the uniform raster (smooth chain: compared tightly on the GPU)
    uraster, udx = loop_rasters('u')
    uenv = ref_map_env(R, uraster, udx)
    with torch.no_grad():
        uemb = R.scenario_gen.detach_embed_info(tm.embed(batch, map_idx, uenv))
    out['u_map_feat'] = npy(uemb['map_feat'])
    out['u_past_feat'] = npy(uemb['past_feat'])
    z = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g4b/z').requires_grad_(True)
    pred = tm.decode_embedding(z, uemb, batch, map_idx, uenv)['future_pred']
    gz, = torch.autograd.grad((pred * rw).sum(), [z])
    out['u_pred'] = npy(pred)
    out['u_gz'] = npy(gz)
    save('g4b_nc5.npz', **out)


G5_SIZES = [4, 6, 2]
ADV_WEIGHTS = {'coll_veh': 20.0, 'coll_veh_plan': 20.0, 'coll_env': 20.0, 'init_z': 0.5, 'init_z_atk': 0.05,
               'motion_prior': 1.0, 'motion_prior_atk': 0.005, 'motion_prior_ext': 0.0001, 'match_ext': 10.0,
               'adv_crash': 2.0}
REFINE_WEIGHTS = {'coll_veh': 100.0, 'coll_env': 100.0, 'init_z': 0.01, 'motion_prior': 1.0}


def g5_inputs(R_or_none, model_builder):
    """Trajectories for the loss fixtures: a rollout from a dense (window 14 m) batch so vehicle
    and environment collisions actually occur."""
    batch, map_idx, raster, dx = build_inputs(G5_SIZES, 'g5', window=14.0)
    return batch, map_idx, raster, dx


def g5_losses(R):
    tm, _ = ref_model(R)
    batch, map_idx, raster, dx = g5_inputs(R, None)
    env = ref_map_env(R, raster, dx)
    with torch.no_grad():
        emb = R.scenario_gen.detach_embed_info(tm.embed(batch, map_idx, env))
    NA = batch.past.shape[0]
    B = len(G5_SIZES)
    ego_mask = torch.zeros((NA,), dtype=torch.bool)
    ego_mask[batch.ptr[:-1]] = True
    veh_att = tm.get_att_normalizer().unnormalize(batch.lw)
    mapixes = map_idx[batch.batch]
    out = {}

    def run_avoid(tag, buf, single, FT):
        z = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g5/z').requires_grad_(True)
        pred = tm.decode_embedding(z, emb, batch, map_idx, env, nfuture=FT)['future_pred']
        out['pred_' + tag] = npy(pred)
        predu = tm.get_normalizer().unnormalize(pred)
        if single is None:
            lf = R.adv_losses.AvoidCollLoss(REFINE_WEIGHTS, veh_att, mapixes, env, z.clone().detach() * 0.9, veh_coll_buffer=buf)
            ld = lf(predu, z, emb['prior_out'])
        else:
            lf = R.adv_losses.AvoidCollLoss(REFINE_WEIGHTS, veh_att, mapixes, env, z[ego_mask].clone().detach() * 0.9,
                                            veh_coll_buffer=buf, single_veh_idx=0, ptr=batch.ptr)
            ld = lf(predu, z[ego_mask], (emb['prior_out'][0][ego_mask], emb['prior_out'][1][ego_mask]))
        ld['loss'].backward()
        for k, v in ld.items():
            out['%s_%s' % (tag, k)] = npy(v)
        out[tag + '_gz'] = npy(z.grad)

    run_avoid('avoid02', 0.2, None, 16)
    run_avoid('avoid05s', 0.5, 0, 12)

    # adversarial loss, ego-planner mode (target trajectory = ego GT future), min time 2, infront 0.0
    z = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g5/z')
    other_z = z[~ego_mask].clone().requires_grad_(True)
    tgt_z = z[ego_mask].clone().requires_grad_(True)
    zc = R.adv_optim.collate_tgt_other_z(batch, tgt_z.detach(), other_z)
    planner = batch.future_gt[ego_mask][:, :, :4]
    pred = tm.decode_embedding(zc, emb, batch, map_idx, env, ext_future=planner)['future_pred']
    lf = R.adv_losses.AdvGenLoss(ADV_WEIGHTS, veh_att, mapixes, env, other_z.clone().detach() * 0.9, batch.ptr,
                                 veh_coll_buffer=0.1, crash_loss_min_time=2, crash_loss_min_infront=0.0)
    oprior = (emb['prior_out'][0][~ego_mask], emb['prior_out'][1][~ego_mask])
    ld = lf(tm.get_normalizer().unnormalize(pred), tm.get_normalizer().unnormalize(planner), other_z, oprior,
            return_mins=True)
    ld['loss'].backward()
    for k, v in ld.items():
        out['adv_%s' % k] = npy(v) if torch.is_tensor(v) else np.asarray(v)
    out['adv_gz'] = npy(other_z.grad)
    out['adv_pred'] = npy(pred)
    # same with a fixed attacker per scene and without the in-front test
    other_z2 = z[~ego_mask].clone().requires_grad_(True)
    zc = R.adv_optim.collate_tgt_other_z(batch, tgt_z.detach(), other_z2)
    pred = tm.decode_embedding(zc, emb, batch, map_idx, env, ext_future=planner)['future_pred']
    lf2 = R.adv_losses.AdvGenLoss(ADV_WEIGHTS, veh_att, mapixes, env, other_z2.clone().detach() * 0.9, batch.ptr,
                                  veh_coll_buffer=0.1, crash_loss_min_time=0, crash_loss_min_infront=None)
    atk = torch.tensor([1, 2, 1]) + batch.ptr[:-1]
    ld = lf2(tm.get_normalizer().unnormalize(pred), tm.get_normalizer().unnormalize(planner), other_z2, oprior,
             attack_agt_idx=atk)
    ld['loss'].backward()
    for k, v in ld.items():
        out['adv2_%s' % k] = npy(v)
    out['adv2_gz'] = npy(other_z2.grad)
    # all-behind case: put every attacker behind the target => fallback branch
    far = tm.get_normalizer().unnormalize(pred).detach().clone()
    tgt_far = tm.get_normalizer().unnormalize(planner).clone()
    tgt_far[:, :, 0] += 500.0
    tgt_far[:, :, 2] = 1.0
    tgt_far[:, :, 3] = 0.0
    ld = lf(far, tgt_far, other_z.detach(), oprior, return_mins=True)
    out['advbehind_crash'] = npy(ld['adv_crash_loss'])
    out['advbehind_min_agt'] = np.asarray(ld['min_agt'])
    out['advbehind_min_t'] = np.asarray(ld['min_t'])

    # TgtMatchingLoss (with its prior-term quirk)
    tl = R.adv_losses.TgtMatchingLoss(ADV_WEIGHTS)
    tz = z[ego_mask].clone().requires_grad_(True)
    tprior = (emb['prior_out'][0][ego_mask], emb['prior_out'][1][ego_mask])
    ld = tl(tm.get_normalizer().unnormalize(pred[ego_mask]), tm.get_normalizer().unnormalize(planner), tz, tprior)
    for k, v in ld.items():
        out['tgt_%s' % k] = npy(v)

    # raw vehicle-collision matrices + a no-collision case + training variants
    fine = R.adv_losses.interp_traj(tm.get_normalizer().unnormalize(pred).detach(), 3)
    vl = R.adv_losses.VehCollLoss(veh_att, buffer_dist=0.1, ptr=batch.ptr)
    pens, cmask = vl(fine, return_raw=True)
    valid = vl.off_diag_mask.view(1, NA, NA).expand_as(pens)
    out['veh_raw_pens_valid'] = npy(pens[valid])
    out['veh_raw_mask_valid'] = npy(cmask[valid])
    spread = fine.clone()
    spread[:, :, 0] += 40.0 * torch.arange(NA).view(NA, 1)
    out['veh_nocoll'] = npy(vl(spread))
    tvl = R.tm_losses.VehCollLoss(veh_att, batch.batch, batch.ptr)
    tp, npairs = tvl(tm.get_normalizer().unnormalize(pred).detach())
    out['train_veh_pens'] = npy(tp)
    out['train_veh_npairs'] = npy(npairs)
    ego = batch.ptr[:-1]
    tel = R.tm_losses.EnvCollLoss(tm.get_att_normalizer().unnormalize(batch.lw[ego]), map_idx, env, pred.shape[1])
    out['train_env_pens'] = npy(tel(tm.get_normalizer().unnormalize(pred[ego]).detach()))

    # full training loss (fwd values only; weight grads of a few tensors as checksums)
    tm.train()
    for p in tm.parameters():
        p.grad = None
    eps_post = synth.f32(synth.counter_normal((NA, 32), 'g5/eps_post'))
    eps_prior = synth.f32(synth.counter_normal((NA, 32), 'g5/eps_prior'))
    seq = [eps_post, eps_prior]
    tm.rsample = lambda mean, var: mean + seq.pop(0) * torch.sqrt(var)
    net_out = tm(batch, map_idx, env, future_sample=True)
    tw = {'recon': 1.0, 'kl': 0.004, 'coll_veh_prior': 0.05, 'coll_env_prior': 0.1}
    tloss = R.tm_losses.TrafficModelLoss(tw, tm.get_normalizer(), tm.get_att_normalizer())
    ld = tloss(batch, net_out, map_idx, env)
    ld['loss'].sum().backward()
    for k, v in ld.items():
        out['train_%s' % k] = npy(v)
    out['train_future_pred'] = npy(net_out['future_pred'])
    out['train_future_samp'] = npy(net_out['future_samp'])
    sdg = {n: p.grad for n, p in tm.named_parameters() if p.grad is not None}
    out['train_ngrads'] = np.asarray(len(sdg))
    for n in ('decoder_net.mlp_out.net.6.weight', 'decoder_memory.weight_hh_l0', 'map_conv.0.weight',
              'map_feature.weight', 'prior_net.msg.0.edge_mlp.net.0.weight', 'past_encoder.net.0.weight',
              'posterior_net.mlp_in.net.0.weight', 'future_encoder.net.9.bias'):
        out['train_grad/' + n] = npy(sdg[n]).astype(np.float32) if sdg[n].numel() < 20000 else npy(sdg[n].reshape(-1)[:20000])
    save('g5_losses.npz', **out)


G6_SIZES = [8]


def g6_loop(R):
    """config-1-like: 1 scene, 8 agents, refine closure, 10 Adam iterations at lr 0.05."""
    tm, _ = ref_model(R)
    batch, map_idx, raster, dx = build_inputs(G6_SIZES, 'g6', window=16.0)
    env = ref_map_env(R, raster, dx)
    with torch.no_grad():
        emb = R.scenario_gen.detach_embed_info(tm.embed(batch, map_idx, env))
    z0 = synth.make_latents(emb['prior_out'][0], emb['prior_out'][1], key='g6/z')
    z = z0.clone().detach().requires_grad_(True)
    opt = torch.optim.Adam([z], lr=0.05)
    lf = R.adv_losses.AvoidCollLoss(REFINE_WEIGHTS, tm.get_att_normalizer().unnormalize(batch.lw), map_idx[batch.batch],
                                    env, z.clone().detach(), veh_coll_buffer=0.2)
    zs, losses, grads = [], [], []
    keys = ['coll_veh_loss', 'coll_env_loss', 'motion_prior_loss', 'init_loss', 'loss']
    for it in range(10):
        opt.zero_grad()
        pred = tm.decode_embedding(z, emb, batch, map_idx, env, nfuture=16)['future_pred']
        ld = lf(tm.get_normalizer().unnormalize(pred), z, emb['prior_out'])
        ld['loss'].backward()
        grads.append(npy(z.grad).copy())
        losses.append([float(torch.mean(ld[k])) for k in keys])
        opt.step()
        zs.append(npy(z).copy())
    save('g6_loop.npz', z=np.stack(zs), grad=np.stack(grads), losses=np.asarray(losses), loss_keys=np.asarray(keys))

    # the reference's own refine_traffic_optim() draws its starting sample unseeded, so it cannot be a fixture; its
    # init / adv / sol loop functions are run as they are by g6_loops() below, and tests/test_dropin_reference.py
    # (build container only) executes the reference's loop functions on top of this package's modules.


LOOP_WEIGHTS = dict(ADV_WEIGHTS)
LOOP_WEIGHTS.update({'init_motion_prior_ext': 0.01, 'init_match_ext': 10.0, 'sol_motion_prior': 0.005, 'sol_coll_veh': 10.0,
                     'sol_coll_env': 10.0, 'sol_motion_prior_ext': 0.001, 'sol_match_ext': 10.0, 'sol_init_z': 0.0})
LOOP_ITERS = {'u': 10, 't': 5}      # uniform raster (smooth chain, compared tightly) / textured raster (chaotic, loose)
LOOP_LR = 0.05


def loop_rasters(kind):
    """'u': every pixel drivable, no dividers -> the crop (hence the CNN feature) does not depend on the pose and the whole
    optimisation chain is smooth; 't': the textured raster of the other fixtures."""
    if kind == 'u':
        raster = torch.zeros((1, 4, RASTER_HW, RASTER_HW), dtype=torch.uint8)
        raster[:, 0] = 1
        return raster, torch.tensor([[0.25, 0.25]], dtype=torch.float64)
    return synth.make_raster(RASTER_HW, RASTER_HW)


class _LoopRecorder(object):
    """Harness around the reference's OWN loop functions (they expose no hook): while active, every ``Adam.step`` call
    records the leaf latents and their gradients as they are BEFORE the update, together with the loss dicts the
    reference's loss modules returned during that closure."""

    def __init__(self, R, prefixes):
        self.R, self.prefixes, self.trace, self._cur = R, prefixes, [], {}

    def __enter__(self):
        rec = self
        self._orig_step = torch.optim.Adam.step
        self._orig_fwd = {}

        def step(opt, closure=None):
            ps = [p for g in opt.param_groups for p in g['params']]
            ent = {'z': [p.detach().clone() for p in ps], 'grad': [p.grad.detach().clone() for p in ps]}
            ent.update(rec._cur)
            rec._cur = {}
            rec.trace.append(ent)
            return rec._orig_step(opt, closure)
        torch.optim.Adam.step = step
        for cls_name, prefix in self.prefixes.items():
            cls = getattr(self.R.adv_losses, cls_name)
            orig = cls.forward
            self._orig_fwd[cls_name] = orig

            def fwd(mod, *a, _orig=orig, _prefix=prefix, **k):
                out = _orig(mod, *a, **k)
                for key, v in out.items():
                    if torch.is_tensor(v):
                        rec._cur[_prefix + key] = v.detach().clone()
                return out
            cls.forward = fwd
        return self

    def __exit__(self, *exc):
        torch.optim.Adam.step = self._orig_step
        for cls_name, orig in self._orig_fwd.items():
            getattr(self.R.adv_losses, cls_name).forward = orig


def _store_trace(out, tag, trace):
    keys = sorted(k for k in trace[0] if k not in ('z', 'grad'))
    out[tag + '/loss_keys'] = np.asarray(keys)
    out[tag + '/losses'] = np.asarray([[float(torch.mean(e[k])) for k in keys] for e in trace])
    for i in range(len(trace[0]['z'])):
        out['%s/z%d' % (tag, i)] = np.stack([npy(e['z'][i]) for e in trace])
        out['%s/grad%d' % (tag, i)] = np.stack([npy(e['grad'][i]) for e in trace])


def g6_loops(R):
    """The reference's own run_init_optim -> run_adv_gen_optim(planner 'ego') -> run_find_solution_optim on the G5 scene,
    over the uniform raster (10 iterations each) and the textured one (5 iterations): per iteration the leaf latents,
    their gradients and the mean of every loss-dict entry; plus each loop's inputs so a test can start any loop from
    the reference's own state."""
    import contextlib
    import io
    out = {}
    for kind in ('u', 't'):
        tm, _ = ref_model(R)
        batch, map_idx, _, _ = g5_inputs(R, None)
        raster, dx = loop_rasters(kind)
        env = ref_map_env(R, raster, dx)
        n = LOOP_ITERS[kind]
        with torch.no_grad():
            emb = R.scenario_gen.detach_embed_info(tm.embed(batch, map_idx, env))
        NA = batch.past.shape[0]
        ego = torch.zeros((NA,), dtype=torch.bool)
        ego[batch.ptr[:-1]] = True
        pm, pv = emb['prior_out']
        tp, op = (pm[ego], pv[ego]), (pm[~ego], pv[~ego])
        z0 = synth.make_latents(pm, pv, key='g6l/z')
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            with _LoopRecorder(R, {'TgtMatchingLoss': ''}) as rec:
                z1, traj1, _ = R.init_optim.run_init_optim(z0, batch.future_gt[:, :, :4], batch.future_vis, LOOP_LR, LOOP_WEIGHTS, tm,
                                                           batch, env, map_idx, n, emb, emb['prior_out'])
            _store_trace(out, kind + '/init', rec.trace)
            z1 = z1.detach()
            with _LoopRecorder(R, {'TgtMatchingLoss': 'tgt_match_', 'AdvGenLoss': 'adv_'}) as rec:
                z2, fin, _, agt, tt = R.adv_optim.run_adv_gen_optim(z1, LOOP_LR, LOOP_WEIGHTS, tm, batch, env, map_idx, n, emb, 'ego',
                                                                    tp, op, 2, 0.0)
            _store_trace(out, kind + '/adv', rec.trace)
            z2 = z2.detach()
            with _LoopRecorder(R, {'AvoidCollLoss': 'tgt_', 'TgtMatchingLoss': 'other_'}) as rec:
                z3, sol, _ = R.sol_optim.run_find_solution_optim(z2, fin, 16, LOOP_LR, LOOP_WEIGHTS, tm, batch, env, map_idx, n, emb,
                                                                 tp, op)
            _store_trace(out, kind + '/sol', rec.trace)
        out[kind + '/init/z_out'] = npy(z1)
        out[kind + '/init/traj_out'] = npy(traj1)
        out[kind + '/adv/z_out'] = npy(z2)
        out[kind + '/adv/final_result_traj'] = npy(fin)
        out[kind + '/adv/min_agt'] = np.asarray(agt)
        out[kind + '/adv/min_t'] = np.asarray(tt)
        out[kind + '/sol/z_out'] = npy(z3)
        out[kind + '/sol/traj_out'] = npy(sol)
    save('g6_loops.npz', **out)


G6H_SIZES = [4, 3]
G6H_ITERS = 5


def g6h_inputs():
    """Scenes whose agents sit on the synthetic lane graph (so that the rule-based planner has lanes to follow), uniform
    raster (smooth chain), one map."""
    lg = synth.make_lane_graph()
    scenes = []
    for b, n in enumerate(G6H_SIZES):
        poses = synth.lane_scene_poses(lg, n, 'g6h/%d' % b, radius=25.0, centre=(128.0 + 40.0 * b, 128.0))
        scenes.append(synth.make_scene(n, 'g6h/%d' % b, poses=poses))
    batch = Batch.from_data_list(scenes)
    map_idx = torch.zeros((len(G6H_SIZES),), dtype=torch.long)
    raster, dx = loop_rasters('u')
    return lg, batch, map_idx, raster, dx


def g6h_hardcode(R):
    """The reference's run_adv_gen_optim in CLOSED LOOP against its own rule-based planner (planner_name='hardcode',
    adv_gen_rule_based.cfg: planner 'hardcode'), 5 iterations: per-iteration latents / gradients / loss entries + results."""
    import contextlib
    import io
    tm, _ = ref_model(R)
    lg, batch, map_idx, raster, dx = g6h_inputs()
    env = ref_map_env(R, raster, dx)
    env.lane_graphs = {m: lg for m in env.map_list}
    with torch.no_grad():
        emb = R.scenario_gen.detach_embed_info(tm.embed(batch, map_idx, env))
    NA = batch.past.shape[0]
    ego = torch.zeros((NA,), dtype=torch.bool)
    ego[batch.ptr[:-1]] = True
    pm, pv = emb['prior_out']
    tp, op = (pm[ego], pv[ego]), (pm[~ego], pv[~ego])
    z0 = synth.make_latents(pm, pv, key='g6h/z')
    planner = R.planner.HardcodeNuscPlanner(env, R.planner_base.PlannerConfig(**R.planner.CONFIG_DICT['default']))
    out = {}
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
        with _LoopRecorder(R, {'TgtMatchingLoss': 'tgt_match_', 'AdvGenLoss': 'adv_'}) as rec:
            z2, fin, _, agt, tt = R.adv_optim.run_adv_gen_optim(z0, LOOP_LR, LOOP_WEIGHTS, tm, batch, env, map_idx, G6H_ITERS, emb,
                                                                'hardcode', tp, op, 2, 0.0, planner=planner)
    _store_trace(out, 'h/adv', rec.trace)
    out['h/adv/z_out'] = npy(z2)
    out['h/adv/final_result_traj'] = npy(fin)
    out['h/adv/min_agt'] = np.asarray(agt)
    out['h/adv/min_t'] = np.asarray(tt)
    save('g6h_hardcode.npz', **out)


G10_SIZES = [5, 3]


def g10_inputs():
    """Two scenes on the synthetic lane graph for the rule-based planner: agents on lane nodes (one moved off every lane,
    one standing still), unnormalised initial states, constant-velocity futures of the non-ego agents with a NaN tail."""
    lg = synth.make_lane_graph()
    states, atts, mask, obs = [], [], [], []
    t = np.linspace(0.5, 6.0, 12)
    for b, n in enumerate(G10_SIZES):
        px, py, h, s = synth.lane_scene_poses(lg, n, 'g10/%d' % b, centre=(128.0 + 40.0 * b, 128.0))
        if b == 0:
            px[3] += 9.0; py[3] += 7.0; h[3] += 0.9            # off every lane: constant-heading prediction
            s[2] = 0.0                                         # parked on its lane
        lw = np.stack([4.2 + 0.4 * synth.counter_uniform((n,), 'g10/l%d' % b), 1.9 + 0.2 * synth.counter_uniform((n,), 'g10/w%d' % b)], -1)
        st = np.stack([px, py, np.cos(h), np.sin(h), s, np.zeros(n)], -1)
        states.append(st); atts.append(lw); mask += [b] * n
        fut = np.stack([px[1:, None] + s[1:, None] * np.cos(h[1:, None]) * t[None], py[1:, None] + s[1:, None] * np.sin(h[1:, None]) * t[None],
                        np.broadcast_to(np.cos(h[1:, None]), (n - 1, 12)), np.broadcast_to(np.sin(h[1:, None]), (n - 1, 12))], -1)
        if b == 1:
            fut[1, 7:] = np.nan                                # this agent's observations end after 3.5 s
        obs.append(fut)
    ptr = np.concatenate([[0], np.cumsum([n - 1 for n in G10_SIZES])])
    return (lg, synth.f32(np.concatenate(states)), synth.f32(np.concatenate(atts)), torch.tensor(mask), np.concatenate(obs).astype(np.float64),
            t, ptr)


class _LaneEnv(object):
    def __init__(self, lg, n=1):
        self.map_list = ['synthetic-%d' % i for i in range(n)]
        self.lane_graphs = {m: lg for m in self.map_list}


def g10_planner(R):
    """HardcodeNuscPlanner.rollout of the reference (both shipped configurations) on the g10 scenes."""
    import contextlib
    import io
    lg, st, att, mask, obs, t, ptr = g10_inputs()
    out = {}
    for name in ('default', 'final_tuned_val_1'):
        pl = R.planner.HardcodeNuscPlanner(_LaneEnv(lg), R.planner_base.PlannerConfig(**R.planner.CONFIG_DICT[name]))
        pl.reset(st, att, mask, len(G10_SIZES), torch.zeros((len(G10_SIZES),), dtype=torch.long))
        with contextlib.redirect_stdout(io.StringIO()):
            plan = pl.rollout(obs.copy(), t, ptr, t, control_all=False)
            plan2 = pl.rollout(obs.copy(), t, ptr, t, control_all=False)        # rollouts do not mutate the reset state
        assert torch.equal(plan, plan2)
        out['plan_' + name] = plan.numpy()
    save('g10_planner.npz', **out)


def g7_sample(R):
    """sample_batched NS=3 with injected eps + feasibility-style outputs."""
    tm, _ = ref_model(R)
    batch, map_idx, raster, dx = build_inputs([4, 2], 'g7')
    env = ref_map_env(R, raster, dx)
    NA = batch.past.shape[0]
    eps = synth.f32(synth.counter_normal((3, NA, 32), 'g7/eps'))
    tm.rsample = lambda mean, var: mean + eps * torch.sqrt(var)
    with torch.no_grad():
        so = tm.sample_batched(batch, map_idx, env, 3, include_mean=True, nfuture=8)
    save('g7_sample.npz', future_pred=npy(so['future_pred']), z_samp=npy(so['z_samp']),
         z_logprob=npy(so['z_logprob']), z_mdist=npy(so['z_mdist']))


def g8_inputs():
    """One scene of 6 agents x 4 samples x 8 steps (UNNORMALISED world poses on the g2 raster) for the feasibility /
    drivable-area tests: the ego drives along +x through the map centre, the others start on a ring around it."""
    raster, dx, _, _, _ = g2_inputs()
    NA, NS, FT = 6, 4, 8
    t = torch.arange(FT, dtype=torch.float32).view(1, 1, FT)
    ang0 = synth.f32(synth.counter_uniform((NA, NS, 1), 'g8/ang', -3.1, 3.1))
    rad = synth.f32(synth.counter_uniform((NA, NS, 1), 'g8/rad', 2.0, 30.0))
    spd = synth.f32(synth.counter_uniform((NA, NS, 1), 'g8/spd', 0.0, 3.0))
    hd = synth.f32(synth.counter_uniform((NA, NS, 1), 'g8/hd', -3.1, 3.1))
    cx, cy = 128.0, 128.0
    x = cx + rad * torch.cos(ang0) + spd * t * torch.cos(hd)
    y = cy + rad * torch.sin(ang0) + spd * t * torch.sin(hd)
    world = torch.stack([x, y, torch.cos(hd).expand(NA, NS, FT), torch.sin(hd).expand(NA, NS, FT)], dim=-1).contiguous()
    world[0, :, :, 0] = cx - 6.0 + 2.0 * t[0]          # the ego: same in every sample
    world[0, :, :, 1] = cy
    world[0, :, :, 2] = 1.0
    world[0, :, :, 3] = 0.0
    world[5, 1, 3:5, :] = float('nan')                  # NaN frames never count in the drivable-area test
    lw = synth.f32(synth.counter_uniform((NA, 2), 'g8/lw', 0.0, 1.0)) * torch.tensor([1.5, 0.6]) + torch.tensor([4.0, 1.7])
    return raster, dx, world, lw


def g8_checks(R):
    """determine_feasibility_nusc and compute_coll_rate_env of the reference on the g8 scene."""
    tm, _ = ref_model(R)
    raster, dx, world, lw = g8_inputs()
    env = ref_map_env(R, raster, dx)
    nrm, att = tm.get_normalizer(), tm.get_att_normalizer()
    samples = nrm.normalize(torch.nan_to_num(world, nan=0.0))
    samples = torch.where(torch.isnan(world), world, samples)
    map_idx = torch.tensor([1])
    out = {}
    cases = [(15.0, 0, 0.0, None, False), (15.0, 2, 0.5, None, True), (25.0, 1, 0.0, 0.0, True), (8.0, 0, 1.0, -0.5, False)]
    clean = torch.nan_to_num(samples, nan=0.0)
    for ci, (th, t0, vel, front, sep) in enumerate(cases):
        f, st, ds = R.scenario_gen.determine_feasibility_nusc(clean.clone(), nrm, th, feasibility_time=t0, feasibility_vel=vel,
                                                              feasibility_infront_min=front, check_non_drivable_separation=sep,
                                                              map_env=env, map_idx=map_idx)
        out['feas_%d' % ci] = npy(f)
        out['step_%d' % ci] = npy(st)
        out['dist_%d' % ci] = npy(ds)
    batch, _, _, _ = build_inputs([6], 'g8')
    batch.lw = att.normalize(lw)
    for name, ego_only in (('all', False), ('ego', True)):
        cd = R.tm_losses.compute_coll_rate_env(batch, map_idx, samples.clone(), env, nrm, att, ego_only=ego_only)
        out['env_did_%s' % name] = npy(cd['did_collide'])
        out['env_num_%s' % name] = np.array([cd['num_coll_map'], cd['num_traj_map']])
    cd = R.tm_losses.compute_coll_rate_env_from_traj(world.clone(), lw, map_idx.expand(world.shape[0]), env)
    out['env_did_traj'] = npy(cd['did_collide'])
    out['env_num_traj'] = np.array([cd['num_coll_map'], cd['num_traj_map']])
    save('g8_checks.npz', **out)


def g9_inputs():
    """Scene of 3 agents for the scenario wire format: normalised futures / latents from the counter generator."""
    batch, map_idx, raster, dx = build_inputs([3], 'g9')
    f = lambda shape, key: synth.f32(synth.counter_uniform(shape, 'g9/' + key, -1.0, 1.0))
    return batch, raster, dx, dict(init=f((3, 12, 4), 'init'), adv=f((3, 12, 4), 'adv'), sol=f((3, 12, 4), 'sol'),
                                   internal=f((1, 12, 4), 'int'), adv_z=f((3, 32), 'zadv'), sol_z=f((3, 32), 'zsol'),
                                   pm=f((3, 32), 'pm'), pv=f((3, 32), 'pv').abs() + 0.1, bike=f((12, 2), 'bike'))


def g9_wire(R):
    """prepare_output_dict of the reference (full and minimal argument sets) as JSON fixtures."""
    import json
    tm, _ = ref_model(R)
    batch, raster, dx, t = g9_inputs()
    env = ref_map_env(R, raster, dx)
    env.map_list = ['synthetic-map-0']
    full = R.scenario_gen.prepare_output_dict(batch, 0, env, 0.5, tm, t['init'], t['adv'], sol_fut_traj=t['sol'], attack_agt=2,
                                              attack_t=7, adv_z=t['adv_z'], sol_z=t['sol_z'], prior_distrib=(t['pm'], t['pv']),
                                              attack_bike_params=t['bike'], internal_ego_traj=t['internal'])
    mini = R.scenario_gen.prepare_output_dict(batch, 0, env, 0.5, tm, t['init'], t['adv'])
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, 'g9_scenario_full.json'), 'w') as f:
        json.dump(full, f)
    with open(os.path.join(here, 'g9_scenario_min.json'), 'w') as f:
        json.dump(mini, f)
    print('wrote g9_scenario_{full,min}.json')


def _install_exact_shapely():
    """shapely is absent from the image: the reference's collision checks get a ``shapely.geometry.Polygon`` stand-in whose
    ``intersection(...).area`` / ``union(...).area`` are EXACT (rational arithmetic on the corners the reference hands over,
    tests/golden/make_iou_exact.py) -- everything around that call (pair ordering, skipping, thresholds, NaN frames, counts)
    is the reference's own code."""
    from fractions import Fraction
    import make_iou_exact as mx

    class _Area(object):
        def __init__(self, area):
            self.area = area

    class Polygon(object):
        def __init__(self, corners):
            self.p = mx._ccw(mx._frac_poly(np.asarray(corners, dtype=np.float64)))
            self.a = mx._area2(self.p) / 2

        def _inter(self, other):
            pts = [v for v in self.p if mx._inside(v, other.p)] + [v for v in other.p if mx._inside(v, self.p)]
            for i in range(4):
                p, p2 = self.p[i], self.p[(i + 1) % 4]
                r = (p2[0] - p[0], p2[1] - p[1])
                for j in range(4):
                    q, q2 = other.p[j], other.p[(j + 1) % 4]
                    sv = (q2[0] - q[0], q2[1] - q[1])
                    den = r[0] * sv[1] - r[1] * sv[0]
                    if den == 0:
                        continue
                    qp = (q[0] - p[0], q[1] - p[1])
                    t = (qp[0] * sv[1] - qp[1] * sv[0]) / den
                    u = (qp[0] * r[1] - qp[1] * r[0]) / den
                    if 0 <= t <= 1 and 0 <= u <= 1:
                        pts.append((p[0] + t * r[0], p[1] + t * r[1]))
            pts = list(dict.fromkeys(pts))
            if len(pts) < 3:
                return Fraction(0)
            import functools
            cx = sum(v[0] for v in pts) / len(pts)
            cy = sum(v[1] for v in pts) / len(pts)
            rel = [(v[0] - cx, v[1] - cy) for v in pts]

            def cmp(a, b):
                ha, hb = mx._half(a), mx._half(b)
                if ha != hb:
                    return -1 if ha < hb else 1
                c = a[0] * b[1] - a[1] * b[0]
                return -1 if c > 0 else (1 if c < 0 else 0)
            rel.sort(key=functools.cmp_to_key(cmp))
            return abs(mx._area2(rel)) / 2

        def intersection(self, other):
            return _Area(float(self._inter(other)))

        def union(self, other):
            return _Area(float(self.a + other.a - self._inter(other)))

    m = types.ModuleType('shapely')
    gm = types.ModuleType('shapely.geometry')
    gm.Polygon = Polygon
    m.geometry = gm
    sys.modules['shapely'] = m
    sys.modules['shapely.geometry'] = gm


G11_SIZES = [5, 3, 4]


def g11_inputs():
    """Three scenes, 4 samples x 8 steps of NORMALISED futures around the scenes' own ground truth: small per-sample offsets so
    that some pairs overlap at some step, one sample with NaN frames, one prediction horizon shorter than the data's."""
    batch, _, _, _ = build_inputs(G11_SIZES, 'g11')
    NA, NS, FT = batch.past.shape[0], 4, 8
    base = batch.future_gt[:, :FT, :4].unsqueeze(1).expand(NA, NS, FT, 4)
    off = synth.f32(synth.counter_uniform((NA, NS, 1, 2), 'g11/off', -0.06, 0.06))          # normalised units (15 m per unit)
    drift = synth.f32(synth.counter_uniform((NA, NS, 1, 2), 'g11/drift', -0.01, 0.01)) * torch.arange(FT).view(1, 1, FT, 1)
    pred = base.clone()
    pred[..., :2] = pred[..., :2] + off + drift
    ang = synth.f32(synth.counter_uniform((NA, NS, 1), 'g11/ang', -0.3, 0.3))
    c, s_ = torch.cos(ang), torch.sin(ang)
    hx, hy = pred[..., 2].clone(), pred[..., 3].clone()
    pred[..., 2], pred[..., 3] = c * hx - s_ * hy, s_ * hx + c * hy
    # agents 1 and 2 of scene 0 are pulled onto agent 0 in sample 2 (certain overlap); agent 6 loses frames in sample 1
    pred[1, 2, :, :2] = pred[0, 2, :, :2] + 0.02
    pred[2, 2, 3:, :2] = pred[0, 2, 3:, :2] - 0.03
    pred[6, 1, 2:5] = float('nan')
    return batch, pred.contiguous()


def g11_eval(R):
    """compute_disp_err and compute_coll_rate_veh of the reference (src/losses/traffic_model.py:297-364, 465-545); the latter
    with the exact shapely stand-in above."""
    _install_exact_shapely()
    tm, _ = ref_model(R)
    nrm, att = tm.get_normalizer(), tm.get_att_normalizer()
    batch, pred = g11_inputs()
    out = {}
    de = R.tm_losses.compute_disp_err(batch, {'future_pred': torch.nan_to_num(pred, nan=0.0)}, nrm)
    for k, v in de.items():
        out['disp/' + k] = npy(v)
    de6 = R.tm_losses.compute_disp_err(batch, {'future_pred': torch.nan_to_num(pred[:, :, :6], nan=0.0)}, nrm)
    for k, v in de6.items():
        out['disp6/' + k] = npy(v)
    cv = R.tm_losses.compute_coll_rate_veh(batch, {'future_pred': pred.clone()}, nrm, att)
    out['veh/did_collide'] = np.asarray(cv['did_collide'])
    out['veh/num'] = np.array([cv['num_coll_veh'], cv['num_traj_veh']])
    save('g11_eval.npz', **out)


G12_SIZES = [4]


def g12_inputs():
    batch, map_idx, _, _ = build_inputs(G12_SIZES, 'g12', window=14.0)
    raster, dx = loop_rasters('u')
    eps = synth.f32(synth.counter_normal((1, batch.past.shape[0], 32), 'g12/eps'))
    return batch, map_idx, raster, dx, eps


def g12_refine_fn(R):
    """The reference's OWN refine_traffic_optim() (src/refine_traffic_optim.py:146-226) -- its function body executed from its
    file -- with the unseeded prior sample replaced by a counter-generated one: Adam (3 iterations) and the LBFGS branch
    (max_iter 20, strong-Wolfe line search; 2 iterations), samp_future_len = save_future_len = 6, uniform raster."""
    import contextlib
    import io
    import tqdm
    src = open(os.path.join(REF_SRC, 'refine_traffic_optim.py')).read()
    body = src[src.index('def refine_traffic_optim('):src.index('def run_one_epoch(')]
    ns = {'torch': torch, 'optim': torch.optim, 'tqdm': tqdm, 'detach_embed_info': R.scenario_gen.detach_embed_info,
          'AvoidCollLoss': R.adv_losses.AvoidCollLoss}
    exec(compile(body, 'reference:refine_traffic_optim.py', 'exec'), ns)
    out = {}
    for name, use_adam, iters in (('adam', True, 3), ('lbfgs', False, 2)):
        tm, _ = ref_model(R)
        batch, map_idx, raster, dx, eps = g12_inputs()
        env = ref_map_env(R, raster, dx)
        tm.rsample = lambda mean, var: mean + eps * torch.sqrt(var)
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            init_pred, z, res, emb = ns['refine_traffic_optim'](batch, map_idx, env, tm, REFINE_WEIGHTS, iters, 6, 6, use_adam, 0.05)
        out[name + '/init_future_pred'] = npy(init_pred)
        out[name + '/z'] = npy(z)
        out[name + '/result_traj'] = npy(res)
    save('g12_refine_fn.npz', **out)


G8_CASES = [(15.0, 0, 0.0, None, False), (15.0, 2, 0.5, None, True), (25.0, 1, 0.0, 0.0, True), (8.0, 0, 1.0, -0.5, False)]


if __name__ == '__main__':
    torch.set_num_threads(8)
    R = import_reference()
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g4u', 'g4b', 'g5', 'g6', 'g6l', 'g6h', 'g7', 'g8', 'g9', 'g10', 'g11', 'g12']
    fns = {'g1': g1_ops, 'g2': g2_crop, 'g3': g3_gnn, 'g4': g4_rollout, 'g5': g5_losses, 'g6': g6_loop, 'g7': g7_sample,
           'g8': g8_checks, 'g9': g9_wire, 'g6l': g6_loops, 'g4b': g4b_nc5, 'g4u': g4u_rollout, 'g10': g10_planner, 'g6h': g6h_hardcode, 'g11': g11_eval, 'g12': g12_refine_fn}
    for w in which:
        fns[w](R)
