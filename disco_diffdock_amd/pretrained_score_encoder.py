"""Host mirror of the reference's AR latent model at inference:
``PretrainedScoreEncoder`` (models/pretrained_score_encoder.py:9-89) and ``GenericEncoder.encode_ar``
(models/model_classes.py:9-49), latent_vocab == 1.  Everything runs in libddk.so: ``score_model.embed()`` at t = 1 with
unconditional = 1 and the partially decoded input latents is the ordinary score-model forward of the AR checkpoint's own
score-model copy (its own ddk context), the two predictor MLPs with BatchNorm1d(eval) are ``ddk_ar_logits`` and the per-graph
pick + one-hot write is ``ddk_ar_decode`` (csrc/k_ar.hip).  ``encode_ar`` never reads the device back."""
import torch
from torch import nn

from .score_model import complex_for_batch


class GenericEncoder(nn.Module):
    def encode_ar(self, data, sampling_temperature=1.0, choice_fn=None, uniforms=None):
        """assumes graphs of the same complex as input (model_classes.py:10).  Returns the one-hot latents
        ``(latent_l [B*n_lig, D], latent_r [B*n_rec, D])`` on the device; ``self.last_choices`` [B, D] (int32, device) holds the
        picked node of every graph (ligand atoms first).
        Extras for the parity tests: ``uniforms`` [D, B] replaces the device draws of the inverse-CDF pick;
        ``choice_fn(idx, logits)`` replaces the pick altogether (host round trip)."""
        if self.latent_vocab != 1:
            raise RuntimeError('ddk: AR decoding is implemented for latent_vocab == 1')
        sm = self.pretrained_score_model
        pos = data['ligand'].pos
        if not pos.is_cuda:
            raise RuntimeError('ddk AR model runs on the GPU only (no CPU fallback)')
        dev = pos.device
        cx, B = complex_for_batch(data, dev, ctx=sm.ctx)
        D = self.input_latent_dim
        latent_l = torch.zeros(B * cx.n_lig, D, device=dev)
        latent_r = torch.zeros(B * cx.n_rec, D, device=dev)
        choices = torch.full((B, D), -1, dtype=torch.int32, device=dev)
        T = float(sampling_temperature)
        p3 = pos.reshape(B, -1, 3)
        cx.keep_receptor_features(True)        # embed(): the predictors read the receptor rows of the last conv layer too
        try:
            for idx in range(D):
                logits = self._logits(cx, p3, latent_l, latent_r)
                if choice_fn is not None:
                    c = choice_fn(idx, logits * T)[:, 0].to(dev)
                    rows = torch.arange(B, device=dev)
                    in_lig = c < cx.n_lig
                    latent_l[(rows * cx.n_lig + c)[in_lig], idx] = 1
                    latent_r[(rows * cx.n_rec + c - cx.n_lig)[~in_lig], idx] = 1
                    choices[:, idx] = c.int()
                    continue
                u = None
                if T < 100:
                    u = uniforms[idx].to(dev).float().contiguous() if uniforms is not None else torch.rand(B, device=dev)
                cx.ar_decode(logits, T, u, idx, latent_l, latent_r, choices)
        finally:
            cx.keep_receptor_features(False)
        self.last_choices = choices
        return latent_l, latent_r

    def _logits(self, cx, pos, latent_l, latent_r):
        """one embed() pass + the predictors: [B, n_lig + n_rec]"""
        cx.set_latents(latent_l, latent_r, 1.0)            # unconditional = 1 (pretrained_score_encoder.py:62-64)
        cx.score_forward(pos, 1.0, 1.0, 1.0)               # set_time(data, 1, 1, 1, ...) (:59-61)
        return cx.ar_logits(pos.shape[0])


class PretrainedScoreEncoder(GenericEncoder):
    def __init__(self, pretrained_score_model, ns, latent_dim, latent_vocab, latent_no_batchnorm=False, latent_dropout=0.0,
                 latent_hidden_dim=128, input_latent_dim=0, apply_gumbel_softmax=True):
        super().__init__()
        assert input_latent_dim > 0
        if latent_dim != 1:
            raise RuntimeError('ddk: the AR predictors emit one logit per node (latent_dim = 1, utils/model_utils.py:133-139)')
        if pretrained_score_model.cfg['num_conv_layers'] < 3:
            raise RuntimeError('ddk: AR predictors need the full irreps (num_conv_layers >= 3: inputs [x[:, :ns] | x[:, -ns:]])')
        self.ns, self.latent_dim, self.latent_vocab = ns, latent_dim, latent_vocab
        self.latent_temperature = 1.0
        self.input_latent_dim = input_latent_dim
        self.apply_gumbel_softmax = apply_gumbel_softmax
        self.pretrained_score_model = pretrained_score_model      # ddk-backed TensorProductScoreModel
        self.hidden, self.no_bn = latent_hidden_dim, bool(latent_no_batchnorm)
        self.last_choices = None

    def predictor_spec(self):
        """name -> shape of the predictor part of the AR checkpoint (models/pretrained_score_encoder.py:24-45)."""
        H, n_in, spec = self.hidden, 2 * self.ns, {}
        for name in ('latent_s_predictor', 'latent_r_predictor'):
            for i, shape in ((0, (H, n_in)), (4, (H, H)), (8, (self.latent_dim, H))):
                spec[f'{name}.{i}.weight'], spec[f'{name}.{i}.bias'] = shape, (shape[0],)
            if not self.no_bn:
                for i in (1, 5):
                    for k in ('weight', 'bias', 'running_mean', 'running_var'):
                        spec[f'{name}.{i}.{k}'] = (H,)
                    spec[f'{name}.{i}.num_batches_tracked'] = ()
        return spec

    def load_state_dict(self, state_dict, strict=True):
        pre = 'pretrained_score_model.'
        own = {k: v for k, v in state_dict.items() if not k.startswith(pre)}
        spec = self.predictor_spec()
        missing = [k for k in spec if k not in own and not k.endswith('num_batches_tracked')]
        unexpected = [k for k in own if k not in spec]
        bad = [k for k in spec if k in own and tuple(own[k].shape) != tuple(spec[k])]
        if missing or bad or (strict and unexpected):
            raise RuntimeError(f'AR state_dict mismatch: missing {missing}, unexpected {unexpected}, mis-shaped {bad}')
        # the predictors live in the SAME ddk context as the AR checkpoint's score-model copy (ddk_finalize_weights folds the BatchNorms)
        self.pretrained_score_model.load_state_dict({k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)}, strict=strict,
                                                    extra={k: v for k, v in own.items() if k in spec})
        return torch.nn.modules.module._IncompatibleKeys([], unexpected)

    def logits(self, data):
        """[B, latent_dim, n_lig + n_rec] logits of PretrainedScoreEncoder.forward with apply_gumbel_softmax=False, for the input
        latents ``data[...].input_latent`` (the batch is left untouched)."""
        if self.training:
            raise RuntimeError('ddk: inference (eval mode) only')
        pos = data['ligand'].pos
        if not pos.is_cuda:
            raise RuntimeError('ddk AR model runs on the GPU only (no CPU fallback)')
        cx, B = complex_for_batch(data, pos.device, ctx=self.pretrained_score_model.ctx)
        cx.keep_receptor_features(True)
        try:
            lg = self._logits(cx, pos.reshape(B, -1, 3), data['ligand'].input_latent.to(pos.device), data['receptor'].input_latent.to(pos.device))
        finally:
            cx.keep_receptor_features(False)
        return lg[:, None, :]

    def forward(self, data):
        raise RuntimeError('ddk: use encode_ar() / logits(); the Gumbel-softmax training path is outside the hot path')
