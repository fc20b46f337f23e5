"""CPU restatement of the all-atom confidence model forward.  TEST INFRASTRUCTURE.

Follows models/all_atom_score_model.py (paths relative to /root/reference) in ``confidence_mode`` with the constructor
arguments ``get_model`` maps from workdir/paper_confidence_model/model_parameters.yml (utils/model_utils.py:25-68:
all_atoms -> AAScoreModel, sh_lmax absent -> 2, use_old_atom_encoder absent -> True, num_confidence_outputs =
len(rmsd_classification_cutoff)+1 = 2).  Parameters are read from a dict with the reference's ``state_dict()`` key names.

The e3nn / torch_cluster / torch_scatter arithmetic comes from the ``*_lite`` restatements -> PARITY UNPINNED
(oracle/__init__.py); the golden vector for this file is produced by the reference's own module running on the same
stand-ins (tests/golden/make_golden.py), which pins the model code, the state_dict layout and the op order, not e3nn."""
from dataclasses import dataclass

import math
import torch
import torch.nn.functional as F

from . import e3nn_lite as o3
from .cluster_lite import radius, radius_graph
from .scatter_lite import scatter
from .score_model_ref import (LIG_FEATURE_DIMS, REC_FEATURE_DIMS, gaussian_smearing, mlp2, sinusoidal_embedding, irreps_dim,
                              _bn_spec)

REC_ATOM_FEATURE_DIMS = (38, 119, 23, 38)     # datasets_utils/process_mols.py:81-86


@dataclass
class ConfidenceModelConfig:
    ns: int = 24
    nv: int = 6
    num_conv_layers: int = 5
    sigma_embed_dim: int = 32
    distance_embed_dim: int = 32
    cross_distance_embed_dim: int = 32
    lig_max_radius: float = 5.0        # args.max_radius
    rec_max_radius: float = 30.0       # ctor default (all_atom_score_model.py:55)
    cross_max_distance: float = 80.0
    dynamic_max_cross: bool = True
    embedding_scale: float = 10000.0   # paper_confidence_model yml
    batch_norm: bool = True
    sh_lmax: int = 2
    num_confidence_outputs: int = 2
    confidence_no_batchnorm: bool = False
    lm_embedding_dim: int = 1280
    in_lig_edge_features: int = 4

    @property
    def irrep_seq(self):   # all_atom_score_model.py:117-122 (use_second_order_repr = False)
        ns, nv = self.ns, self.nv
        return [f'{ns}x0e', f'{ns}x0e + {nv}x1o', f'{ns}x0e + {nv}x1o + {nv}x1e', f'{ns}x0e + {nv}x1o + {nv}x1e + {ns}x0o']

    def conv_irreps(self, l):
        s = self.irrep_seq
        return s[min(l, len(s) - 1)], s[min(l + 1, len(s) - 1)]

    @property
    def sh_irreps(self):
        return o3.Irreps.spherical_harmonics(self.sh_lmax)


def old_atom_encoder(x, P, prefix, n_cat, n_scalar, lm_dim):
    """models/layers.py:103-116 (OldAtomEncoder.forward): the 'scalar features' slice is x[:, n_cat:n_cat+n_scalar] and the
    language-model slice is the LAST lm_dim columns - with x = [ids | ESM | sigma_emb] that is ESM[:32] and [ESM[32:] | sigma_emb]."""
    emb = 0
    for i in range(n_cat):
        emb = emb + P[f'{prefix}.atom_embedding_list.{i}.weight'][x[:, i].long()]
    emb = emb + F.linear(x[:, n_cat:n_cat + n_scalar], P[f'{prefix}.linear.weight'], P[f'{prefix}.linear.bias'])
    if lm_dim:
        emb = F.linear(torch.cat([emb, x[:, -lm_dim:]], dim=1), P[f'{prefix}.lm_embedding_layer.weight'], P[f'{prefix}.lm_embedding_layer.bias'])
    return emb


def conv_layer(P, prefix, cfg, l, node_attr, edge_index, edge_attr, edge_sh, out_nodes=None, eps=1e-5):
    """all_atom_score_model.py:37-50 with residual=False: BN(scatter_mean(tp(x[dst], sh, fc(edge_attr)), src))."""
    in_irreps, out_irreps = cfg.conv_irreps(l)
    out_nodes = out_nodes or node_attr.shape[0]
    edge_src, edge_dst = edge_index
    w = mlp2(edge_attr, P, f'{prefix}.fc', 0, 3)
    tp = o3.FullyConnectedTensorProduct(in_irreps, cfg.sh_irreps, out_irreps)(node_attr[edge_dst], edge_sh, w)
    out = scatter(tp, edge_src, dim=0, dim_size=out_nodes, reduce='mean')
    if cfg.batch_norm:
        out = o3.batch_norm_eval(out, out_irreps, P[f'{prefix}.batch_norm.weight'], P[f'{prefix}.batch_norm.bias'],
                                 P[f'{prefix}.batch_norm.running_mean'], P[f'{prefix}.batch_norm.running_var'], eps)
    return out


def _sh(vec, cfg):
    return o3.spherical_harmonics(cfg.sh_irreps, vec, normalize=True, normalization='component')


def _intra_graph(pos, edge_index, sigma_emb, stop, cfg, dtype, extra=None):
    src, dst = edge_index
    vec = pos[dst] - pos[src]
    attr = [sigma_emb[src], gaussian_smearing(vec.norm(dim=-1), stop, cfg.distance_embed_dim, dtype)]
    if extra is not None:
        attr = [extra] + attr
    return torch.cat(attr, 1), _sh(vec, cfg)


def confidence_forward(P, cfg, data, dtype=torch.float32, return_intermediates=False, stop_after=None, group_mask=0x1ff, mask_from=0):
    """all_atom_score_model.py:203-284 in confidence_mode (complex_t is used as sigma directly, :205-207)."""
    ns = cfg.ns
    lig, rec, atom = data['ligand'], data['receptor'], data['atom']
    tr_sigma = data.complex_t['tr'].to(dtype)
    emb = lambda t: sinusoidal_embedding(cfg.embedding_scale * t, cfg.sigma_embed_dim).to(dtype)
    lig_sig, rec_sig, atom_sig = emb(lig.node_t['tr']), emb(rec.node_t['tr']), emb(atom.node_t['tr'])

    # ligand graph (:330-353)
    radius_edges = radius_graph(lig.pos, cfg.lig_max_radius, lig.batch)
    ll_index = torch.cat([data['ligand', 'ligand'].edge_index, radius_edges], 1).long()
    bond_attr = torch.cat([data['ligand', 'ligand'].edge_attr.to(dtype), torch.zeros(radius_edges.shape[-1], cfg.in_lig_edge_features, dtype=dtype)], 0)
    ll_attr, ll_sh = _intra_graph(lig.pos.to(dtype), ll_index, lig_sig, cfg.lig_max_radius, cfg, dtype, extra=bond_attr)
    lig_x = old_atom_encoder(torch.cat([lig.x.to(dtype), lig_sig], 1), P, 'lig_node_embedding', len(LIG_FEATURE_DIMS), cfg.sigma_embed_dim, 0)
    ll_attr = mlp2(ll_attr, P, 'lig_edge_embedding', 0, 3)
    # receptor graph (:355-371)
    rr_index = data['receptor', 'receptor'].edge_index
    rr_attr, rr_sh = _intra_graph(rec.pos.to(dtype), rr_index, rec_sig, cfg.rec_max_radius, cfg, dtype)
    rec_x = old_atom_encoder(torch.cat([rec.x.to(dtype), rec_sig], 1), P, 'rec_node_embedding', len(REC_FEATURE_DIMS), cfg.sigma_embed_dim, cfg.lm_embedding_dim)
    rr_attr = mlp2(rr_attr, P, 'rec_edge_embedding', 0, 3)
    # atom graph (:373-388): distances expanded with the LIGAND expansion
    aa_index = data['atom', 'atom'].edge_index
    aa_attr, aa_sh = _intra_graph(atom.pos.to(dtype), aa_index, atom_sig, cfg.lig_max_radius, cfg, dtype)
    atom_x = old_atom_encoder(torch.cat([atom.x.to(dtype), atom_sig], 1), P, 'atom_node_embedding', len(REC_ATOM_FEATURE_DIMS), cfg.sigma_embed_dim, 0)
    aa_attr = mlp2(aa_attr, P, 'atom_edge_embedding', 0, 3)

    # cross graphs (:390-433)
    if cfg.dynamic_max_cross:
        cut = (tr_sigma * 3 + 20).unsqueeze(1)
        lr_index = radius(rec.pos / cut[rec.batch], lig.pos / cut[lig.batch], 1, rec.batch, lig.batch, max_num_neighbors=10000)
    else:
        lr_index = radius(rec.pos, lig.pos, cfg.cross_max_distance, rec.batch, lig.batch, max_num_neighbors=10000)
    lr_vec = rec.pos[lr_index[1]] - lig.pos[lr_index[0]]
    lr_attr = torch.cat([lig_sig[lr_index[0]], gaussian_smearing(lr_vec.norm(dim=-1), cfg.cross_max_distance, cfg.cross_distance_embed_dim, dtype)], 1)
    lr_sh = _sh(lr_vec.to(dtype), cfg)
    la_index = radius(atom.pos, lig.pos, cfg.lig_max_radius, atom.batch, lig.batch, max_num_neighbors=10000)
    la_vec = atom.pos[la_index[1]] - lig.pos[la_index[0]]
    la_attr = torch.cat([lig_sig[la_index[0]], gaussian_smearing(la_vec.norm(dim=-1), cfg.cross_max_distance, cfg.cross_distance_embed_dim, dtype)], 1)
    la_sh = _sh(la_vec.to(dtype), cfg)
    ar_index = data['atom', 'receptor'].edge_index
    ar_vec = rec.pos[ar_index[1]] - atom.pos[ar_index[0]]
    ar_attr = torch.cat([atom_sig[ar_index[0]], gaussian_smearing(ar_vec.norm(dim=-1), cfg.rec_max_radius, cfg.distance_embed_dim, dtype)], 1)
    ar_sh = _sh(ar_vec.to(dtype), cfg)
    lr_attr = mlp2(lr_attr, P, 'lr_edge_embedding', 0, 3)
    la_attr = mlp2(la_attr, P, 'la_edge_embedding', 0, 3)
    ar_attr = mlp2(ar_attr, P, 'ar_edge_embedding', 0, 3)

    edge_sets = dict(ll=(ll_index, ll_attr, ll_sh), lr=(lr_index, lr_attr, lr_sh), la=(la_index, la_attr, la_sh), aa=(aa_index, aa_attr, aa_sh),
                     ar=(ar_index, ar_attr, ar_sh), rr=(rr_index, rr_attr, rr_sh))
    x0 = dict(lig=lig_x, atom=atom_x, rec=rec_x)
    flip = lambda ei: torch.flip(ei, dims=[0])
    cat3 = lambda e, a, b: torch.cat([e, a[:, :ns], b[:, :ns]], -1)
    L = cfg.num_conv_layers
    for l in range(L):
        if stop_after is not None and l >= stop_after:
            break
        def cv(k, x, ei, ea, sh_, **kw):
            if l >= mask_from and not (group_mask >> k) & 1:      # development aid: conv k sees no edges -> BatchNorm of zeros
                n_out = kw.get('out_nodes') or x.shape[0]
                o_irr = cfg.conv_irreps(l)[1]
                z0 = torch.zeros((n_out, irreps_dim(o_irr)), dtype=x.dtype)
                pre = f'conv_layers.{9 * l + k}.batch_norm'
                return o3.batch_norm_eval(z0, o_irr, P[f'{pre}.weight'], P[f'{pre}.bias'], P[f'{pre}.running_mean'], P[f'{pre}.running_var'], 1e-5) if cfg.batch_norm else z0
            return conv_layer(P, f'conv_layers.{9 * l + k}', cfg, l, x, ei, ea, sh_, **kw)
        n_l, n_a, n_r = lig_x.shape[0], atom_x.shape[0], rec_x.shape[0]
        lig_update = cv(0, lig_x, ll_index, cat3(ll_attr, lig_x[ll_index[0]], lig_x[ll_index[1]]), ll_sh)
        lr_update = cv(1, rec_x, lr_index, cat3(lr_attr, lig_x[lr_index[0]], rec_x[lr_index[1]]), lr_sh, out_nodes=n_l)
        la_update = cv(2, atom_x, la_index, cat3(la_attr, lig_x[la_index[0]], atom_x[la_index[1]]), la_sh, out_nodes=n_l)
        if l != L - 1:
            atom_update = cv(3, atom_x, aa_index, cat3(aa_attr, atom_x[aa_index[0]], atom_x[aa_index[1]]), aa_sh)
            al_update = cv(4, lig_x, flip(la_index), cat3(la_attr, atom_x[la_index[1]], lig_x[la_index[0]]), la_sh, out_nodes=n_a)
            ar_update = cv(5, rec_x, ar_index, cat3(ar_attr, atom_x[ar_index[0]], rec_x[ar_index[1]]), ar_sh, out_nodes=n_a)
            rec_update = cv(6, rec_x, rr_index, cat3(rr_attr, rec_x[rr_index[0]], rec_x[rr_index[1]]), rr_sh)
            rl_update = cv(7, lig_x, flip(lr_index), cat3(lr_attr, rec_x[lr_index[1]], lig_x[lr_index[0]]), lr_sh, out_nodes=n_r)
            ra_update = cv(8, atom_x, flip(ar_index), cat3(ar_attr, rec_x[ar_index[1]], atom_x[ar_index[0]]), ar_sh, out_nodes=n_r)
        lig_x = F.pad(lig_x, (0, lig_update.shape[-1] - lig_x.shape[-1])) + lig_update + la_update + lr_update
        if l != L - 1:
            atom_x = F.pad(atom_x, (0, atom_update.shape[-1] - atom_x.shape[-1])) + atom_update + al_update + ar_update
            rec_x = F.pad(rec_x, (0, rec_update.shape[-1] - rec_x.shape[-1])) + rec_update + ra_update + rl_update

    scalar = torch.cat([lig_x[:, :ns], lig_x[:, -ns:]], dim=1) if L >= 3 else lig_x[:, :ns]
    pooled = scatter(scalar, lig.batch, dim=0, dim_size=data.num_graphs, reduce='mean')
    h = pooled
    for i_lin, i_bn in ((0, 1), (4, 5)):      # Sequential(Linear, BN1d, ReLU, Dropout, Linear, BN1d, ReLU, Dropout, Linear)  :143-153
        h = F.linear(h, P[f'confidence_predictor.{i_lin}.weight'], P[f'confidence_predictor.{i_lin}.bias'])
        if not cfg.confidence_no_batchnorm:
            k = f'confidence_predictor.{i_bn}'
            h = (h - P[f'{k}.running_mean']) / torch.sqrt(P[f'{k}.running_var'] + 1e-5) * P[f'{k}.weight'] + P[f'{k}.bias']
        h = torch.relu(h)
    conf = F.linear(h, P['confidence_predictor.8.weight'], P['confidence_predictor.8.bias']).squeeze(dim=-1)
    if return_intermediates:
        return conf, dict(lig_node_attr=lig_x, atom_node_attr=atom_x, rec_node_attr=rec_x, pooled=pooled, edge_sets=edge_sets, x0=x0,
                          counts=dict(ll=ll_index.shape[1], lr=lr_index.shape[1], la=la_index.shape[1], aa=aa_index.shape[1],
                                      ar=ar_index.shape[1], rr=rr_index.shape[1]))
    return conf


def state_dict_spec(cfg):
    ns, sd, dd, cd = cfg.ns, cfg.sigma_embed_dim, cfg.distance_embed_dim, cfg.cross_distance_embed_dim
    spec = {}

    def lin(name, o, i):
        spec[f'{name}.weight'] = (o, i)
        spec[f'{name}.bias'] = (o,)

    for pre, dims, lm in (('lig_node_embedding', LIG_FEATURE_DIMS, 0), ('rec_node_embedding', REC_FEATURE_DIMS, cfg.lm_embedding_dim),
                          ('atom_node_embedding', REC_ATOM_FEATURE_DIMS, 0)):
        for i, d in enumerate(dims):
            spec[f'{pre}.atom_embedding_list.{i}.weight'] = (d, ns)
        lin(f'{pre}.linear', ns, sd)
        if lm:
            lin(f'{pre}.lm_embedding_layer', ns, lm + ns)
    lin('lig_edge_embedding.0', ns, cfg.in_lig_edge_features + sd + dd)
    lin('lig_edge_embedding.3', ns, ns)
    for k, n in (('rec', dd), ('atom', dd), ('lr', cd), ('ar', dd), ('la', cd)):
        lin(f'{k}_edge_embedding.0', ns, sd + n)
        lin(f'{k}_edge_embedding.3', ns, ns)
    for k, n in (('lig', dd), ('rec', dd), ('cross', cd)):
        spec[f'{k}_distance_expansion.offset'] = (n,)
    for l in range(cfg.num_conv_layers):
        i_irr, o_irr = cfg.conv_irreps(l)
        W = o3.FullyConnectedTensorProduct(i_irr, cfg.sh_irreps, o_irr).weight_numel
        for k in range(9):
            lin(f'conv_layers.{9 * l + k}.fc.0', 3 * ns, 3 * ns)
            lin(f'conv_layers.{9 * l + k}.fc.3', W, 3 * ns)
            if cfg.batch_norm:
                _bn_spec(spec, f'conv_layers.{9 * l + k}.batch_norm', o_irr)
    lin('confidence_predictor.0', ns, 2 * ns if cfg.num_conv_layers >= 3 else ns)
    lin('confidence_predictor.4', ns, ns)
    lin('confidence_predictor.8', cfg.num_confidence_outputs, ns)
    if not cfg.confidence_no_batchnorm:
        for i in (1, 5):
            for k in ('weight', 'bias', 'running_mean', 'running_var'):
                spec[f'confidence_predictor.{i}.{k}'] = (ns,)
    return spec


def random_state_dict(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    spec, P = state_dict_spec(cfg), {}
    for name, shape in spec.items():
        if name.endswith('distance_expansion.offset'):
            stop = {'lig': cfg.lig_max_radius, 'rec': cfg.rec_max_radius, 'cross': cfg.cross_max_distance}[name.split('_')[0]]
            P[name] = torch.linspace(0.0, stop, shape[0])
        elif 'atom_embedding_list' in name:
            a = math.sqrt(6.0 / (shape[0] + shape[1]))
            P[name] = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif '.batch_norm.' in name or (name.startswith('confidence_predictor') and name.split('.')[1] in ('1', '5')):
            if name.endswith('running_mean') or name.endswith('bias'):
                P[name] = torch.randn(shape, generator=g) * 0.1
            else:
                P[name] = torch.rand(shape, generator=g) + 0.5
        else:
            fan_in = shape[1] if name.endswith('weight') else spec[name[:-4] + 'weight'][1]
            P[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
    return P
