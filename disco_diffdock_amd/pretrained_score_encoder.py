"""Host mirror of the reference's AR latent model at inference:
``PretrainedScoreEncoder`` (models/pretrained_score_encoder.py:9-89) and ``GenericEncoder.encode_ar``
(models/model_classes.py:9-49), latent_vocab == 1.  The expensive part - ``score_model.embed()`` at t = 1 with
unconditional = 1 and the partially decoded input latents - runs in libddk.so (the AR checkpoint carries its own copy
of the score model, loaded into its own ddk context); the two 32->128->128->1 predictor MLPs with BatchNorm1d(eval) are
plain dense layers on the [N, 32] scalar channels (torch, rocBLAS)."""
import copy

import torch
from torch import nn

from .diffusion_utils import set_time


class GenericEncoder(nn.Module):
    def encode_ar(self, data, sampling_temperature=1.0, choice_fn=None):
        """assumes graphs of the same complex as input (model_classes.py:10).  ``choice_fn(idx, logits)`` (extra, optional)
        replaces the multinomial draw - used by the parity tests."""
        if self.latent_vocab != 1:
            raise RuntimeError('ddk: AR decoding is implemented for latent_vocab == 1')
        B = data.num_graphs
        dev = data['ligand'].pos.device
        len_lig, len_rec = len(data['ligand'].pos) // B, len(data['receptor'].pos) // B
        latent_l = torch.zeros(len(data['ligand'].pos), self.input_latent_dim, device=dev)
        latent_r = torch.zeros(len(data['receptor'].pos), self.input_latent_dim, device=dev)
        for decoding_idx in range(self.input_latent_dim):
            data['ligand'].input_latent, data['receptor'].input_latent = latent_l, latent_r
            data.decoding_idx = torch.zeros(B, device=dev).long() + decoding_idx
            lat = self.logits(data)[:, 0, :] * sampling_temperature
            assert lat.shape == (B, len_lig + len_rec)
            if sampling_temperature >= 100:
                lat_choice = torch.argmax(lat, 1, keepdim=True)
            elif choice_fn is not None:
                lat_choice = choice_fn(decoding_idx, lat)
            else:
                p = torch.exp(lat)
                if torch.any(torch.isnan(p)) or torch.any(torch.isinf(p)):
                    print("Warning: NaNs or INF in AR setting them to 0")
                    p = torch.nan_to_num(p)
                lat_choice = torch.multinomial(p, 1)
            rows = torch.arange(B, device=dev)
            c = lat_choice[:, 0]
            in_lig = c < len_lig
            latent_l[(rows * len_lig + c)[in_lig], decoding_idx] = 1
            latent_r[(rows * len_rec + c - len_lig)[~in_lig], decoding_idx] = 1
        return latent_l, latent_r


class PretrainedScoreEncoder(GenericEncoder):
    def __init__(self, pretrained_score_model, ns, latent_dim, latent_vocab, latent_no_batchnorm=False, latent_dropout=0.0,
                 latent_hidden_dim=128, input_latent_dim=0, apply_gumbel_softmax=True):
        super().__init__()
        assert input_latent_dim > 0
        self.ns, self.latent_dim, self.latent_vocab = ns, latent_dim, latent_vocab
        self.latent_temperature = 1.0
        self.input_latent_dim = input_latent_dim
        self.apply_gumbel_softmax = apply_gumbel_softmax
        self.pretrained_score_model = pretrained_score_model      # ddk-backed TensorProductScoreModel
        n_in = 2 * ns if pretrained_score_model.cfg['num_conv_layers'] >= 3 else ns

        def predictor():
            bn = (lambda: nn.Identity()) if latent_no_batchnorm else (lambda: nn.BatchNorm1d(latent_hidden_dim))
            return nn.Sequential(nn.Linear(n_in, latent_hidden_dim), bn(), nn.ReLU(), nn.Dropout(latent_dropout),
                                 nn.Linear(latent_hidden_dim, latent_hidden_dim), bn(), nn.ReLU(), nn.Dropout(latent_dropout),
                                 nn.Linear(latent_hidden_dim, latent_dim))
        self.latent_s_predictor = predictor()
        self.latent_r_predictor = predictor()

    def load_state_dict(self, state_dict, strict=True):
        pre = 'pretrained_score_model.'
        self.pretrained_score_model.load_state_dict({k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)}, strict=strict)
        own = {k: v for k, v in state_dict.items() if not k.startswith(pre)}
        missing, unexpected = [], []
        sd = nn.Module.state_dict(self)
        for k in sd:
            if k not in own:
                missing.append(k)
        for k in own:
            if k not in sd:
                unexpected.append(k)
        if strict and (missing or unexpected):
            raise RuntimeError(f'AR state_dict mismatch: missing {missing}, unexpected {unexpected}')
        with torch.no_grad():
            for k, v in own.items():
                if k in sd:
                    sd[k].copy_(v)
        return self

    def logits(self, data):
        """[B, latent_dim, n_lig + n_rec] logits of PretrainedScoreEncoder.forward with apply_gumbel_softmax=False."""
        if self.training:
            raise RuntimeError('ddk: inference (eval mode) only')
        data = copy.copy(data)          # shallow: the forward only rebinds attributes (reference deep-copies, model_classes.py:34)
        lig, rec = data['ligand'], data['receptor']
        dev = lig.pos.device
        lig.latent_h, rec.latent_h = lig.input_latent, rec.input_latent
        B = data.num_graphs
        set_time(data, 1, 1, 1, B, False, dev)
        lig.unconditional = torch.ones((len(lig.pos), 1), device=dev)
        rec.unconditional = torch.ones((len(rec.pos), 1), device=dev)
        lig_h, rec_h = self.pretrained_score_model.embed(data)[:2]
        ns = self.ns
        deep = self.pretrained_score_model.cfg['num_conv_layers'] >= 3
        sl = torch.cat([lig_h[:, :ns], lig_h[:, -ns:]], 1) if deep else lig_h[:, :ns]
        sr = torch.cat([rec_h[:, :ns], rec_h[:, -ns:]], 1) if deep else rec_h[:, :ns]
        sl, sr = self.latent_s_predictor(sl), self.latent_r_predictor(sr)
        n_l, n_r = sl.shape[0] // B, sr.shape[0] // B
        return torch.cat([sl.reshape(B, n_l, -1), sr.reshape(B, n_r, -1)], 1).transpose(1, 2)

    def forward(self, data):
        raise RuntimeError('ddk: use encode_ar() / logits(); the Gumbel-softmax training path is outside the hot path')
