"""ORACLE (test infrastructure only) -- the CVAE traffic prior, restated functionally.

A CPU, plain-torch-fp32 restatement of the reference's ``TrafficModel`` hot path
(embed / decode_embedding / sample_batched / forward).  It is written as pure functions over a
``state_dict`` with the reference's 174 parameter names, so the same
weights drive the oracle, the reference (when generating golden vectors) and the HIP product.

The message passing follows the third-party semantics the reference relies on
(torch-geometric 1.7.1 ``MessagePassing(aggr='max', flow='source_to_target')`` +
torch-scatter 2.0.7 ``scatter(reduce='max')``, not vendored under /root/reference): per edge
(j -> i) gather ``x_i = x[edge_index[1]]``, ``x_j = x[edge_index[0]]``; aggregate with an
element-wise max grouped by target, 0 for nodes without incoming edges.  Because no reference
test pins that boundary, parity at it is anchored on golden vectors produced by running the
reference itself (tests/golden/make_golden.py) -- see the header of tests/golden/README.md.
"""
import torch
import torch.nn.functional as F

from .geometry import Normalizer, transform2frame, bicycle_step
from .mapenv import map_crop


# --------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------

def mlp(sd, prefix, x):
    """Linear -> (LayerNorm -> ReLU -> Linear)*, last layer linear.
    Restates MLP (reference src/models/common.py:8-44): Linear at net.0, then for k>=0
    LayerNorm at net.{3k+1}, ReLU at net.{3k+2}, Linear at net.{3k+3}."""
    x = F.linear(x, sd[prefix + '.net.0.weight'], sd[prefix + '.net.0.bias'])
    k = 0
    while (prefix + '.net.%d.weight' % (3 * k + 3)) in sd:
        g = sd[prefix + '.net.%d.weight' % (3 * k + 1)]
        b = sd[prefix + '.net.%d.bias' % (3 * k + 1)]
        x = F.relu(F.layer_norm(x, (g.shape[0],), g, b, 1e-5))
        x = F.linear(x, sd[prefix + '.net.%d.weight' % (3 * k + 3)], sd[prefix + '.net.%d.bias' % (3 * k + 3)])
        k += 1
    return x


def scatter_max_zero(msg, index, n):
    """Element-wise max of ``msg (E,C)`` rows grouped by ``index (E,)`` into ``(n,C)``; rows with
    no member are 0 (the reference relies on that, src/models/interaction_net.py:188)."""
    out = torch.zeros((n, msg.shape[1]), dtype=msg.dtype, device=msg.device)
    if msg.shape[0] == 0:
        return out
    idx = index.view(-1, 1).expand_as(msg)
    return out.scatter_reduce(0, idx, msg, reduce='amax', include_self=False)


def interaction_net(sd, prefix, x_in, pos, sem, edge_index):
    """One message-passing round + in/out MLPs.  ``x_in (N,F)`` or ``(N,NS,F)``, ``pos`` likewise
    with 4 channels.  Restates SceneInteractionNet.forward and AgentInteractionConv
    forward/message/update (reference src/models/interaction_net.py:52-77, 121-218) for the
    configuration TrafficModel uses (k=1, MLP update, max aggregation)."""
    x = mlp(sd, prefix + '.mlp_in', x_in)
    multi = x.dim() == 3
    N = x.shape[0]
    NS = x.shape[1] if multi else 1
    D = x.shape[-1]
    src, dst = edge_index[0], edge_index[1]
    E = src.shape[0]
    xs = x.reshape(N, NS, D)
    ps = pos.reshape(N, NS, 4)
    if E > 0:
        x_i, x_j = xs[dst], xs[src]                    # (E,NS,D)
        p_i, p_j = ps[dst].reshape(E * NS, 4), ps[src].reshape(E * NS, 4)
        rel = transform2frame(p_i, p_j.unsqueeze(1))[:, 0, :]
        rel = torch.where(torch.isnan(rel), torch.zeros_like(rel), rel).reshape(E, NS, 4)
        s_i = sem[dst].unsqueeze(1).expand(E, NS, sem.shape[1])
        s_j = sem[src].unsqueeze(1).expand(E, NS, sem.shape[1])
        m = mlp(sd, prefix + '.msg.0.edge_mlp', torch.cat([x_i, x_j, s_i, s_j, rel], dim=-1))
        C = m.shape[-1]
        aggr = scatter_max_zero(m.reshape(E, NS * C), dst, N).reshape(N, NS, C)
    else:
        aggr = torch.zeros((N, NS, D), dtype=x.dtype, device=x.device)
    s_n = sem.unsqueeze(1).expand(N, NS, sem.shape[1])
    upd = mlp(sd, prefix + '.msg.0.update_mlp', torch.cat([xs, aggr, s_n], dim=-1))
    out = mlp(sd, prefix + '.mlp_out', upd)
    return out if multi else out[:, 0]


def gru_step(sd, prefix, x, h, num_layers=3):
    """One time step of a stacked GRU: ``x (N,I)``, ``h (L,N,H)`` -> (top output, new h).
    Gate order r,z,n; n = tanh(W_in x + b_in + r*(W_hn h + b_hn)); h' = (1-z)*n + z*h
    (torch.nn.GRU semantics used at reference src/models/traffic_model.py:152-156, 686-688)."""
    if hasattr(torch, '_VF') and hasattr(torch._VF, 'gru'):
        # Same library primitive nn.GRU dispatches to, so the oracle reproduces the reference's GRU rounding
        # bit for bit (the rollout is chaotic through the per-step raster re-sampling: a 1e-7 difference in a
        # pose can flip a crop pixel and move the map feature by 1e-3).  The explicit gate arithmetic below is
        # the definition and is what the HIP kernel implements.
        flat = []
        for l in range(num_layers):
            flat += [sd['%s.weight_ih_l%d' % (prefix, l)], sd['%s.weight_hh_l%d' % (prefix, l)],
                     sd['%s.bias_ih_l%d' % (prefix, l)], sd['%s.bias_hh_l%d' % (prefix, l)]]
        out, hn = torch._VF.gru(x.unsqueeze(1), h.contiguous(), flat, True, num_layers, 0.0, False, False, True)
        return out[:, 0], hn
    new_h = []
    inp = x
    for l in range(num_layers):
        gi = F.linear(inp, sd['%s.weight_ih_l%d' % (prefix, l)], sd['%s.bias_ih_l%d' % (prefix, l)])
        gh = F.linear(h[l], sd['%s.weight_hh_l%d' % (prefix, l)], sd['%s.bias_hh_l%d' % (prefix, l)])
        H = h.shape[-1]
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        hn = (1.0 - z) * n + z * h[l]
        new_h.append(hn)
        inp = hn
    return inp, torch.stack(new_h, dim=0)


def map_cnn(sd, crop_f32, nlayers=6, strides=None):
    """6x[Conv2d(stride 2, no padding) -> GroupNorm(1) -> ReLU] -> flatten -> Linear
    (reference src/models/traffic_model.py:69-87, 437-440)."""
    x = crop_f32
    for l in range(nlayers):
        st = 2 if strides is None else strides[l]
        x = F.conv2d(x, sd['map_conv.%d.weight' % (3 * l)], sd['map_conv.%d.bias' % (3 * l)], stride=st)
        g = sd['map_conv.%d.weight' % (3 * l + 1)]
        x = F.relu(F.group_norm(x, 1, g, sd['map_conv.%d.bias' % (3 * l + 1)], 1e-5))
    return F.linear(x.reshape(x.shape[0], -1), sd['map_feature.weight'], sd['map_feature.bias'])


# --------------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------------

class OracleTrafficModel(object):
    """Functional restatement of TrafficModel (reference src/models/traffic_model.py:23-735) for
    the default configuration (MLP trajectory encoders, bicycle output)."""

    def __init__(self, sd, state_norm, att_norm, bike, PT=4, FT=12, NC=2, z_size=32, nconv=6):
        self.sd = sd
        self.normalizer = Normalizer(*state_norm) if isinstance(state_norm, tuple) else state_norm
        self.att_normalizer = Normalizer(*att_norm) if isinstance(att_norm, tuple) else att_norm
        self.bike = bike
        self.PT, self.FT, self.NC, self.z_size = PT, FT, NC, z_size
        self.nconv = nconv
        self.dt = 0.5

    def get_normalizer(self):
        return self.normalizer

    def get_att_normalizer(self):
        return self.att_normalizer

    # -- encoders ---------------------------------------------------------------------------
    def encode_map(self, pos_norm, batch_of_agent, map_idx, map_env, return_crop=False, crop_only=False):
        """pos_norm (NA,4) or (NA,NS,4) NORMALISED -> map feature (NA,[NS,]64).
        Restates encode_map (reference src/models/traffic_model.py:416-451) without the in-place
        graph (un)normalisation round trip (it has no observable effect)."""
        multi = pos_norm.dim() == 3
        NA = pos_norm.shape[0]
        mapixes = map_idx[batch_of_agent]
        pos = self.normalizer.unnormalize(pos_norm)
        if multi:
            NS = pos.shape[1]
            pos = pos.reshape(NA * NS, 4)
            mapixes = mapixes.unsqueeze(1).expand(NA, NS).reshape(-1)
        crop_u8 = map_crop(map_env.nusc_raster, map_env.nusc_dx, pos, mapixes, map_env.bounds, L=map_env.L, W=map_env.W)
        if crop_only:
            return None, crop_u8
        feat = map_cnn(self.sd, crop_u8.to(torch.float), self.nconv)
        feat = feat.reshape(NA, NS, -1) if multi else feat
        return (feat, crop_u8) if return_crop else feat

    def _encode_traj(self, prefix, g, traj, vis):
        NA, T, _ = traj.shape
        local = transform2frame(g.past[:, -1, :4], traj[:, :, :4])
        local = torch.cat([local, traj[:, :, 4:]], dim=2)
        local = torch.where((vis == 0.0).unsqueeze(-1), torch.zeros_like(local), local)
        local = torch.cat([local, vis.unsqueeze(-1)], dim=-1)
        att = g.lw.unsqueeze(1).expand(NA, T, 2)
        enc_in = torch.cat([torch.cat([local, att], dim=-1).reshape(NA, -1), g.sem], dim=1)
        return mlp(self.sd, prefix, enc_in)

    def encode_past(self, g):
        """(reference src/models/traffic_model.py:453-486)"""
        return self._encode_traj('past_encoder', g, g.past, g.past_vis)

    def encode_future(self, g):
        """(reference src/models/traffic_model.py:488-523)"""
        return self._encode_traj('future_encoder', g, g.future, g.future_vis)

    def prior(self, g, map_feat, past_feat):
        """(reference src/models/traffic_model.py:545-565)"""
        x = torch.cat([past_feat, map_feat, g.sem], dim=-1)
        out = interaction_net(self.sd, 'prior_net', x, g.past[:, -1, :4], g.sem, g.edge_index)
        return out[:, :self.z_size], torch.exp(out[:, self.z_size:])

    def posterior(self, g, map_feat, past_feat, future_feat):
        """(reference src/models/traffic_model.py:525-543)"""
        x = torch.cat([past_feat, future_feat, map_feat, g.sem], dim=-1)
        out = interaction_net(self.sd, 'posterior_net', x, g.past[:, -1, :4], g.sem, g.edge_index)
        return out[:, :self.z_size], torch.exp(out[:, self.z_size:])

    def embed(self, g, map_idx, map_env):
        """(reference src/models/traffic_model.py:372-403)"""
        map_feat = self.encode_map(g.past[:, -1, :4], g.batch, map_idx, map_env)
        past_feat = self.encode_past(g)
        out = {'prior_out': self.prior(g, map_feat, past_feat), 'map_feat': map_feat, 'past_feat': past_feat}
        if 'future' in g:
            future_feat = self.encode_future(g)
            out['posterior_out'] = self.posterior(g, map_feat, past_feat, future_feat)
        return out

    # -- decoder ----------------------------------------------------------------------------
    def decode(self, g, map_feat, past_feat, z, map_idx, map_env, ext_future=None, nfuture=None,
               return_trace=False, crop_poses=None):
        """Autoregressive rollout; ``z (NA,D)`` or ``(NA,NS,D)`` -> normalised global
        (x,y,hx,hy) of shape ``(NA,FT,4)`` / ``(NA,NS,FT,4)``.
        Restates autoregressive_decoder (reference src/models/traffic_model.py:589-704).

        ``crop_poses`` (test hook, not in the reference): ``(NA,[NS,]>=FT-1,4)`` normalised poses at which the raster is cropped
        for steps 1 .. FT-1 INSTEAD of this rollout's own (detached) poses.  The reference crops at ``pos.detach()`` (:694-695), so
        the crop is data, not part of the autograd graph: feeding the poses of ANOTHER evaluation of the same rollout (the HIP
        path's, 1e-7 away) removes the one discontinuous step of the chain -- a pose difference in the last bit can flip crop
        pixels and move a map feature by 4e-3 -- and leaves a smooth function that two fp32 implementations must agree on
        tightly, forward and backward.  ``self.last_crop_flips`` then holds, per (row, step), whether the crop at the forced pose
        differs from the crop at this rollout's own pose."""
        NA = map_feat.shape[0]
        FT = self.FT if nfuture is None else nfuture
        multi = z.dim() == 3
        NS = z.shape[1] if multi else 1
        R = NA * NS
        a_mu, a_sd = self.bike['a_stats']
        d_mu, d_sd = self.bike['ddh_stats']

        def rep(t):  # (NA,F) -> (NA,NS,F)
            return t.unsqueeze(1).expand(NA, NS, t.shape[-1])

        prev_state = rep(g.past[:, -1, :]).reshape(R, 6)
        pos = rep(g.past[:, -1, :4])
        cur_past = rep(past_feat)
        cur_map = rep(map_feat)
        sem_r, lw_r = rep(g.sem), rep(g.lw)
        veh_len = rep(self.att_normalizer.unnormalize(g.lw)[:, 0:1]).reshape(R)
        zz = z if multi else z.unsqueeze(1)
        mem = cur_past.reshape(R, -1).unsqueeze(0).expand(3, R, past_feat.shape[1]).contiguous()
        ego = g.ptr[:-1]
        if ext_future is not None:
            ego_rows = (ego.view(-1, 1) * NS + torch.arange(NS, device=ego.device).view(1, NS)).reshape(-1)
            ext = ext_future.unsqueeze(1).expand(ext_future.shape[0], NS, ext_future.shape[1], 4)
            ext = ext.reshape(-1, ext_future.shape[1], 4)
        traj = []
        trace = []
        flips = torch.zeros((R, FT), dtype=torch.bool)
        if crop_poses is not None:
            crop_poses = crop_poses.detach().reshape(NA, NS, crop_poses.shape[-2], 4)
            assert crop_poses.shape[2] >= FT - 1
        for t in range(FT):
            feat = torch.cat([cur_past, cur_map, sem_r, zz, lw_r], dim=-1)
            dec = interaction_net(self.sd, 'decoder_net', feat, pos, g.sem, g.edge_index).reshape(R, 2)
            acc = dec[:, 0] * a_sd + a_mu
            ddh = dec[:, 1] * d_sd + d_mu
            bike = bicycle_step(self.normalizer.unnormalize(prev_state), acc, ddh, veh_len,
                                self.bike['dt'], self.bike['maxhdot'], self.bike['maxs'])
            bike = self.normalizer.normalize(bike)
            glob = bike[:, :4]
            local = transform2frame(prev_state[:, :4], glob.unsqueeze(1))[:, 0]
            traj.append(glob)
            if return_trace:
                trace.append({'dec': dec, 'pos': pos.reshape(R, 4), 'past_feat': cur_past.reshape(R, -1),
                              'map_feat': cur_map.reshape(R, -1), 'local': local})
            if ext_future is not None:
                glob = glob.clone()
                glob[ego_rows] = ext[:, t]
                local = local.clone()
                local[ego_rows] = transform2frame(prev_state[ego_rows][:, :4],
                                                  glob[ego_rows].unsqueeze(1))[:, 0]
            prev_state = bike
            if t < FT - 1:
                top, mem = gru_step(self.sd, 'decoder_memory', local, mem)
                cur_past = top.reshape(NA, NS, -1)
                own = glob.detach().reshape(NA, NS, 4)
                if crop_poses is None:
                    cur_map = self.encode_map(own, g.batch, map_idx, map_env)
                else:
                    forced = crop_poses[:, :, t].clone()
                    if ext_future is not None:
                        forced.reshape(R, 4)[ego_rows] = ext[:, t]          # (the injected ego pose is the same data in both runs)
                    cur_map, crop_f = self.encode_map(forced, g.batch, map_idx, map_env, return_crop=True)
                    with torch.no_grad():
                        _, crop_o = self.encode_map(own, g.batch, map_idx, map_env, return_crop=True, crop_only=True)
                    flips[:, t + 1] = (crop_f != crop_o).flatten(1).any(1)
                pos = glob.reshape(NA, NS, 4)
        out = torch.stack(traj, dim=1)
        out = out.reshape(NA, NS, FT, 4) if multi else out
        self.last_crop_flips = flips
        return (out, trace) if return_trace else out

    def decode_embedding(self, z, embed_out, g, map_idx, map_env, ext_future=None, nfuture=None, crop_poses=None):
        """(reference src/models/traffic_model.py:405-414); ``crop_poses``: see decode (test hook)"""
        return {'future_pred': self.decode(g, embed_out['map_feat'], embed_out['past_feat'], z, map_idx,
                                           map_env, ext_future=ext_future, nfuture=nfuture, crop_poses=crop_poses)}

    def sample_batched(self, g, map_idx, map_env, eps, include_mean=False, nfuture=None):
        """NS prior samples rolled out jointly; ``eps (NS,NA,D)`` is injected because the reference
        draws it unseeded (reference src/models/traffic_model.py:319-370, 706-712)."""
        NS, NA, D = eps.shape
        map_feat = self.encode_map(g.past[:, -1, :4], g.batch, map_idx, map_env)
        past_feat = self.encode_past(g)
        mu, var = self.prior(g, map_feat, past_feat)
        smu = mu.view(1, NA, D).expand(NS, NA, D)
        svar = var.view(1, NA, D).expand(NS, NA, D)
        z = smu + eps * torch.sqrt(svar)
        if include_mean:
            z[-1] = mu
        pred = self.decode(g, map_feat, past_feat, z.transpose(0, 1), map_idx, map_env, nfuture=nfuture)
        dist = torch.distributions.Normal(smu, torch.sqrt(svar))
        return {
            'prior_out': (mu, var),
            'z_samp': z.transpose(0, 1),
            'future_pred': pred,
            'z_logprob': dist.log_prob(z).sum(dim=-1).transpose(0, 1),
            'z_mdist': torch.norm((z - smu) / torch.sqrt(svar), dim=-1).transpose(0, 1),
        }

    def forward(self, g, map_idx, map_env, eps_post=None, eps_prior=None, use_post_mean=False, crop_poses=None):
        """Training forward: posterior-sample rollout (+ prior-sample rollout when ``eps_prior``
        is given).  (reference src/models/traffic_model.py:178-225).  ``crop_poses`` = (poses of the posterior rollout, poses of
        the prior rollout): the test hook of ``decode``."""
        cp_pred, cp_samp = crop_poses if crop_poses is not None else (None, None)
        map_feat = self.encode_map(g.past[:, -1, :4], g.batch, map_idx, map_env)
        past_feat = self.encode_past(g)
        future_feat = self.encode_future(g)
        pmu, pvar = self.prior(g, map_feat, past_feat)
        qmu, qvar = self.posterior(g, map_feat, past_feat, future_feat)
        z = qmu if use_post_mean else qmu + eps_post * torch.sqrt(qvar)
        out = {'prior_out': (pmu, pvar), 'posterior_out': (qmu, qvar),
               'future_pred': self.decode(g, map_feat, past_feat, z, map_idx, map_env, crop_poses=cp_pred)}
        out['crop_flips_pred'] = self.last_crop_flips
        if eps_prior is not None:
            zp = pmu + eps_prior * torch.sqrt(pvar)
            out['future_samp'] = self.decode(g, map_feat, past_feat, zp, map_idx, map_env, crop_poses=cp_samp)
            out['crop_flips_samp'] = self.last_crop_flips
        return out
