"""TrafficModel: the CVAE traffic prior with the reference's API, executed by libstrive_hip.

Host-side mirror of reference src/models/traffic_model.py: same constructor signature, attributes
(``FT, PT, dt, NC, z_size, normalizer, att_normalizer, bicycle_params``), methods (``embed``,
``decode_embedding``, ``sample_batched``, ``sample``, ``reconstruct``, ``forward``, ``encode_map`` ...)
and the same 174-tensor ``state_dict`` layout, so the reference's optimisation
drivers and checkpoints work unchanged.  The arithmetic is not torch: every method hands tensors to
the C ABI (include/strive_hip.h) through strive_amd.ops -- the decoder rollout is ONE call that
enqueues the whole FT-step kernel sequence, and its backward is an explicit reverse-time sweep that
yields d/dz only when the loops call ``decode_embedding`` (what the latent optimisation needs, SURVEY.md
Appendix A) and, inside ``forward()`` (training), also the gradients of every parameter.

Not supported (raise NotImplementedError): ``traj_encoder='gru'``, ``output_bicycle=False`` and
non-default map-CNN shapes -- no shipped config uses them (last paragraph).
"""
import os

import torch
from torch import nn

from .common import MLP
from .interaction_net import SceneInteractionNet
from .. import ops

TRAJ_ENCODER_CHOICES = ['mlp', 'gru']


def calc_conv_out(in_size, kernel_size, stride, padding_size=0):
    """(reference src/utils/torch.py:62-63)"""
    return int(((in_size - kernel_size - 2 * padding_size) // stride) + 1)


class TrafficModel(nn.Module):
    def __init__(self, npast, nfuture, map_obs_size_pix, nclasses,
                 map_feat_size=64, past_feat_size=64, future_feat_size=64, latent_size=32,
                 output_bicycle=True, traj_encoder='mlp', conv_channel_in=4,
                 conv_kernel_list=[7, 5, 5, 3, 3, 3],
                 conv_stride_list=[2, 2, 2, 2, 2, 2],
                 conv_filter_list=[16, 32, 64, 64, 128, 128]):
        super(TrafficModel, self).__init__()
        if traj_encoder != 'mlp':
            raise NotImplementedError("strive_amd implements traj_encoder='mlp' (the only one shipped configs use)")
        if not output_bicycle:
            raise NotImplementedError('strive_amd implements the bicycle output parameterisation only')
        if (list(conv_kernel_list), list(conv_stride_list), list(conv_filter_list), conv_channel_in, map_obs_size_pix) != \
                ([7, 5, 5, 3, 3, 3], [2] * 6, [16, 32, 64, 64, 128, 128], 4, 256):
            raise NotImplementedError('strive_amd implements the default map CNN (4x256x256 crop, 6 stride-2 convs) only')
        if (map_feat_size, past_feat_size, future_feat_size, latent_size) != (64, 64, 64, 32):
            raise NotImplementedError('strive_amd kernels are built for feature size 64 and latent size 32')
        self.normalizer = self.att_normalizer = None
        self.PT, self.FT, self.NC = npast, nfuture, nclasses
        self.dt = 0.5
        self.output_bicycle = True
        self.bicycle_params = None
        # forward(future_sample=True): both decodes as one rollout over the batch stacked twice (STRIVE_STACK_ROLLOUTS=0: one after
        # the other, the round-4 form)
        self.stack_rollouts = os.environ.get('STRIVE_STACK_ROLLOUTS', '1') != '0'
        self.state_size, self.att_feat_size = 6, 2
        self.traj_encoder_type = traj_encoder
        self.mapH = self.mapW = self.map_obs_size_pix = map_obs_size_pix

        # map encoder: parameter containers only, the forward is the fused crop+CNN HIP path
        chans = [conv_channel_in] + list(conv_filter_list)
        layers = []
        size = map_obs_size_pix
        for l in range(6):
            layers += [nn.Conv2d(chans[l], chans[l + 1], kernel_size=conv_kernel_list[l], stride=conv_stride_list[l]),
                       nn.GroupNorm(1, chans[l + 1]), nn.ReLU()]
            size = calc_conv_out(size, conv_kernel_list[l], conv_stride_list[l])
        self.map_conv = nn.Sequential(*layers)
        self.map_feat_in_size = chans[-1] * size * size
        self.map_feat_out_size = map_feat_size
        self.map_feature = nn.Linear(self.map_feat_in_size, map_feat_size)

        self.past_feat_size = past_feat_size
        self.past_in_size = self.NC + self.PT * (self.state_size + self.att_feat_size + 1)
        self.past_encoder = MLP([self.past_in_size, 128, 128, 128, past_feat_size])
        self.future_feat_size = future_feat_size
        self.future_in_size = self.NC + self.FT * (self.state_size + self.att_feat_size + 1)
        self.future_encoder = MLP([self.future_in_size, 128, 128, 128, future_feat_size])

        self.z_size = latent_size
        self.prior_net = SceneInteractionNet(past_feat_size + map_feat_size + self.NC, self.NC, 4,
                                             2 * past_feat_size, 2 * latent_size)
        self.posterior_net = SceneInteractionNet(future_feat_size + past_feat_size + map_feat_size + self.NC, self.NC, 4,
                                                 2 * past_feat_size, 2 * latent_size)
        self.traj_out_size = 2
        self.decoder_net = SceneInteractionNet(latent_size + past_feat_size + map_feat_size + self.NC + self.att_feat_size,
                                               self.NC, 4, 64, self.traj_out_size)
        self.num_memory_layers = 3
        self.decoder_memory = nn.GRU(4, past_feat_size, self.num_memory_layers, batch_first=True)
        self._packs = {}

    # -- reference accessors ------------------------------------------------------------------
    def set_normalizer(self, normalizer):
        self.normalizer = normalizer

    def get_normalizer(self):
        return self.normalizer

    def set_att_normalizer(self, normalizer):
        self.att_normalizer = normalizer

    def get_att_normalizer(self):
        return self.att_normalizer

    def set_bicycle_params(self, bicycle_params):
        self.bicycle_params = bicycle_params

    # -- encoders -------------------------------------------------------------------------------
    def encode_map(self, scene_graph, map_idx, map_env):
        """Map feature (NA,64) / (NA,NS,64) at ``scene_graph.pos`` (NORMALISED), no gradient.
        (reference src/models/traffic_model.py:416-451; the graph is not mutated)"""
        return ops.encode_map(self, scene_graph.pos, scene_graph.batch, map_idx, map_env)

    def _encode_traj(self, encoder, scene_graph, traj, vis):
        return ops.encode_traj(self, encoder, scene_graph, traj, vis)

    def encode_past(self, scene_graph):
        """(reference src/models/traffic_model.py:453-486)"""
        return self._encode_traj(self.past_encoder, scene_graph, scene_graph.past, scene_graph.past_vis)

    def encode_future(self, scene_graph):
        """(reference src/models/traffic_model.py:488-523)"""
        return self._encode_traj(self.future_encoder, scene_graph, scene_graph.future, scene_graph.future_vis)

    def _latent_net(self, net, scene_graph, feats):
        scene_graph.x = torch.cat(feats + [scene_graph.sem], dim=-1)
        scene_graph.pos = scene_graph.past[:, -1, :4]
        out = net(scene_graph)
        return out[:, :self.z_size], torch.exp(out[:, self.z_size:])

    def prior(self, scene_graph, map_feat, past_feat):
        """(reference src/models/traffic_model.py:545-565)"""
        return self._latent_net(self.prior_net, scene_graph, [past_feat, map_feat])

    def encoder(self, scene_graph, map_feat, past_feat, future_feat):
        """(reference src/models/traffic_model.py:525-543)"""
        return self._latent_net(self.posterior_net, scene_graph, [past_feat, future_feat, map_feat])

    def embed(self, scene_graph, map_idx, map_env):
        """(reference src/models/traffic_model.py:372-403)"""
        scene_graph.pos = scene_graph.past[:, -1, :4]
        map_feat = self.encode_map(scene_graph, map_idx, map_env)
        past_feat = self.encode_past(scene_graph)
        out = {'prior_out': self.prior(scene_graph, map_feat, past_feat), 'map_feat': map_feat, 'past_feat': past_feat}
        if 'future' in scene_graph:
            future_feat = self.encode_future(scene_graph)
            out['posterior_out'] = self.encoder(scene_graph, map_feat, past_feat, future_feat)
        return out

    # -- decoder --------------------------------------------------------------------------------
    def decoder(self, scene_graph, map_feat, past_feat, z, map_idx, map_env, ext_future=None, nfuture=None):
        """Autoregressive rollout -> (NA,FT,4) / (NA,NS,FT,4) normalised global poses.
        (reference src/models/traffic_model.py:567-704)"""
        FT = self.FT if nfuture is None else nfuture
        return ops.decoder_rollout(self, scene_graph, map_feat, past_feat, z, map_idx, map_env, ext_future, FT)

    autoregressive_decoder = decoder

    def decode_embedding(self, z, embed_out, scene_graph, map_idx, map_env, ext_future=None, nfuture=None):
        """(reference src/models/traffic_model.py:405-414)"""
        return {'future_pred': self.decoder(scene_graph, embed_out['map_feat'], embed_out['past_feat'], z, map_idx,
                                            map_env, ext_future=ext_future, nfuture=nfuture)}

    def decode_embedding_pair(self, z_a, z_b, embed_out, scene_graph, map_idx, map_env, ext_future=None, nfuture_a=None, nfuture_b=None):
        """``decode_embedding(z_a, ..., nfuture=nfuture_a)`` and ``decode_embedding(z_b, ..., nfuture=nfuture_b)`` for latents of
        EQUAL VALUES that differ only in which leaves they are differentiated for (the complementary-detach pair of the
        adversarial and solution loops, reference src/utils/adv_gen_optim.py:120-131, src/utils/sol_optim.py:73-80): one forward
        rollout serves both, each keeps its own backward (ops._RolloutPairFn).  Not part of the reference's API."""
        fa = self.FT if nfuture_a is None else nfuture_a
        fb = self.FT if nfuture_b is None else nfuture_b
        ta, tb = ops.decoder_rollout_pair(self, scene_graph, embed_out['map_feat'], embed_out['past_feat'], z_a, z_b, map_idx, map_env,
                                          ext_future, fa, fb)
        return {'future_pred': ta}, {'future_pred': tb}

    def rsample(self, mean, var):
        """(reference src/models/traffic_model.py:706-712)"""
        return mean + torch.randn_like(mean) * torch.sqrt(var)

    def sample_batched(self, scene_graph, map_idx, map_env, num_samples, include_mean=False, nfuture=None):
        """(reference src/models/traffic_model.py:319-370)"""
        NA, NS, D = scene_graph.past.size(0), num_samples, self.z_size
        scene_graph.pos = scene_graph.past[:, -1, :4]
        map_feat = self.encode_map(scene_graph, map_idx, map_env)
        past_feat = self.encode_past(scene_graph)
        mu, var = self.prior(scene_graph, map_feat, past_feat)
        smu = mu.view(1, NA, D).expand(NS, NA, D)
        svar = var.view(1, NA, D).expand(NS, NA, D)
        z = self.rsample(smu, svar)
        if include_mean:
            z[-1, :, :] = mu
        pred = self.decoder(scene_graph, map_feat, past_feat, z.transpose(0, 1), map_idx, map_env, nfuture=nfuture)
        # (validate_args=False: the argument checks of torch.distributions read `(scale > 0).all()` and the support test back on
        #  the host -- two synchronisations per call; the reference's arithmetic is the same without them)
        dist = torch.distributions.Normal(smu, torch.sqrt(svar), validate_args=False)
        return {
            'prior_out': (mu, var),
            'z_samp': z.transpose(0, 1),
            'future_pred': pred,
            'z_logprob': dist.log_prob(z).sum(dim=-1).transpose(0, 1),
            'z_mdist': torch.norm((z - smu) / torch.sqrt(svar), dim=-1).transpose(0, 1),
        }

    def sample(self, scene_graph, map_idx, map_env, num_samples, include_mean=False, nfuture=None):
        """Serial sampler (reference src/models/traffic_model.py:259-317)."""
        scene_graph.pos = scene_graph.past[:, -1, :4]
        map_feat = self.encode_map(scene_graph, map_idx, map_env)
        past_feat = self.encode_past(scene_graph)
        mu, var = self.prior(scene_graph, map_feat, past_feat)
        dist = torch.distributions.Normal(mu, torch.sqrt(var), validate_args=False)
        out = {'prior_out': (mu, var), 'z_samp': [], 'z_logprob': [], 'z_mdist': [], 'future_pred': []}
        for sidx in range(num_samples):
            z = mu if (include_mean and sidx == num_samples - 1) else self.rsample(mu, var)
            out['z_samp'].append(z)
            out['z_logprob'].append(dist.log_prob(z).sum(dim=-1))
            out['z_mdist'].append(torch.norm((z - mu) / torch.sqrt(var), dim=-1))
            out['future_pred'].append(self.decoder(scene_graph, map_feat, past_feat, z, map_idx, map_env, nfuture=nfuture))
        for k in ('z_samp', 'z_logprob', 'z_mdist', 'future_pred'):
            out[k] = torch.stack(out[k], dim=1)
        return out

    def reconstruct(self, scene_graph, map_idx, map_env):
        """(reference src/models/traffic_model.py:227-257)"""
        emb = self.embed(scene_graph, map_idx, map_env)
        mu, var = emb['posterior_out']
        return {'posterior_out': (mu, var),
                'future_pred': self.decoder(scene_graph, emb['map_feat'], emb['past_feat'], mu, map_idx, map_env)}

    def forward(self, scene_graph, map_idx, map_env, use_post_mean=False, future_sample=False):
        """Training-time forward (reference src/models/traffic_model.py:178-225): posterior-sample rollout (+ prior-sample
        rollout with ``future_sample``).  This is the one entry point that builds an autograd graph with PARAMETER
        gradients: while it runs (with grad enabled and trainable parameters) every HIP operator -- map CNN, trajectory
        encoders, prior / posterior networks, decoder rollout -- is an autograd Function whose backward is the matching
        ``strive_*_bwd`` call, so ``loss.backward()`` fills ``p.grad`` of all 174 tensors like the reference."""
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if train:
            # every weight pack is rebuilt after an optimiser step and each needs max |w| of its tensors on the host (operand
            # scales are kernel arguments): fetch them all with one read-back instead of one per pack
            from .. import params as _params
            _params.prefetch_absmax(list(self.parameters()))
        with ops.weight_grad_mode(train):
            emb = self.embed(scene_graph, map_idx, map_env)
            pmu, pvar = emb['prior_out']
            qmu, qvar = emb['posterior_out']
            z = qmu if use_post_mean else self.rsample(qmu, qvar)
            out = {'prior_out': (pmu, pvar), 'posterior_out': (qmu, qvar)}
            if future_sample and self.stack_rollouts:
                # the two decodes of the reference (posterior sample, then prior sample) as ONE rollout over the batch stacked
                # twice (ops.decoder_rollout_stacked): scenes do not interact, every kernel of a step runs once instead of twice
                zp = self.rsample(pmu, pvar)
                out['future_pred'], out['future_samp'] = ops.decoder_rollout_stacked(self, scene_graph, emb['map_feat'], emb['past_feat'],
                                                                                     [z, zp], map_idx, map_env, self.FT)
                return out
            out['future_pred'] = self.decoder(scene_graph, emb['map_feat'], emb['past_feat'], z, map_idx, map_env)
            if future_sample:
                zp = self.rsample(pmu, pvar)
                out['future_samp'] = self.decoder(scene_graph, emb['map_feat'], emb['past_feat'], zp, map_idx, map_env)
        return out
