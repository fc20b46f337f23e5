"""CPU restatement of the DiffDock-S / DisCo-DiffDock-S score model forward.  TEST INFRASTRUCTURE.

Plain PyTorch-CPU ops in the reference's op order (including the materialised per-edge weight
tensor [E, W]); works in fp32 (default, == the reference's CPU path) or fp64.  Every function
cites the reference lines it follows (paths relative to /root/reference).  Parameters are read
from a dict with the reference's ``score_model.state_dict()`` key names.

Third-party arithmetic (e3nn / torch_cluster / torch_scatter) comes from the ``*_lite``
restatements in this package -> those parts are PARITY UNPINNED (oracle/__init__.py).
"""
import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

from . import e3nn_lite as o3
from .cluster_lite import radius, radius_graph
from .scatter_lite import scatter

LIG_FEATURE_DIMS = (119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2)   # process_mols.py:62-79
REC_FEATURE_DIMS = (38,)                                                      # process_mols.py:88-90


@dataclass
class ScoreModelConfig:
    """Constructor arguments as ``get_model`` maps them from model_parameters.yml
    (utils/model_utils.py:39-68) plus the ctor defaults it does not override (score_model.py:15-24)."""
    ns: int = 24
    nv: int = 6
    num_conv_layers: int = 5
    sigma_embed_dim: int = 32
    distance_embed_dim: int = 32
    cross_distance_embed_dim: int = 32
    lig_max_radius: float = 5.0        # args.max_radius
    rec_max_radius: float = 30.0       # ctor default
    cross_max_distance: float = 80.0
    center_max_distance: float = 30.0  # ctor default
    dynamic_max_cross: bool = True
    embedding_scale: float = 1000.0
    scale_by_sigma: bool = True
    no_torsion: bool = False
    batch_norm: bool = True
    sh_lmax: int = 1
    latent_dim: int = 0
    latent_vocab: int = 0
    latent_droprate: float = 0.0
    lm_embedding_dim: int = 1280
    in_lig_edge_features: int = 4
    tr_sigma_min: float = 0.1
    tr_sigma_max: float = 19.0
    rot_sigma_min: float = 0.03
    rot_sigma_max: float = 1.55
    tor_sigma_min: float = 0.03
    tor_sigma_max: float = 3.14
    confidence_mode: bool = False      # score_model.py:110-121, 186-189, 263-266: confidence_predictor instead of the score heads
    num_confidence_outputs: int = 1
    confidence_no_batchnorm: bool = False

    @staticmethod
    def from_namespace(args):
        g = lambda k, d: getattr(args, k, d)
        return ScoreModelConfig(
            ns=args.ns, nv=args.nv, num_conv_layers=args.num_conv_layers, sigma_embed_dim=args.sigma_embed_dim,
            distance_embed_dim=args.distance_embed_dim, cross_distance_embed_dim=args.cross_distance_embed_dim,
            lig_max_radius=args.max_radius, cross_max_distance=args.cross_max_distance,
            dynamic_max_cross=args.dynamic_max_cross, embedding_scale=args.embedding_scale,
            scale_by_sigma=args.scale_by_sigma, no_torsion=args.no_torsion, batch_norm=not args.no_batch_norm,
            sh_lmax=g('sh_lmax', 2), latent_dim=g('latent_dim', 0), latent_vocab=g('latent_vocab', 0),
            latent_droprate=g('latent_droprate', 0.0),
            lm_embedding_dim=1280 if args.esm_embeddings_path is not None else 0,
            tr_sigma_min=args.tr_sigma_min, tr_sigma_max=args.tr_sigma_max, rot_sigma_min=args.rot_sigma_min,
            rot_sigma_max=args.rot_sigma_max, tor_sigma_min=args.tor_sigma_min, tor_sigma_max=args.tor_sigma_max)

    def irrep_seq(self):
        ns, nv = self.ns, self.nv   # tensor_layers.py:20-26 (use_second_order_repr=False)
        return [f'{ns}x0e', f'{ns}x0e + {nv}x1o', f'{ns}x0e + {nv}x1o + {nv}x1e',
                f'{ns}x0e + {nv}x1o + {nv}x1e + {ns}x0o']

    def conv_irreps(self, l):
        seq = self.irrep_seq()
        return seq[min(l, len(seq) - 1)], seq[min(l + 1, len(seq) - 1)]


# ---------------------------------------------------------------------------------------------
# small pieces
# ---------------------------------------------------------------------------------------------
def t_to_sigma(t_tr, t_rot, t_tor, cfg):
    """utils/diffusion_utils.py:12-16"""
    return (cfg.tr_sigma_min ** (1 - t_tr) * cfg.tr_sigma_max ** t_tr,
            cfg.rot_sigma_min ** (1 - t_rot) * cfg.rot_sigma_max ** t_rot,
            cfg.tor_sigma_min ** (1 - t_tor) * cfg.tor_sigma_max ** t_tor)


def sinusoidal_embedding(timesteps, embedding_dim, max_positions=10000):
    """utils/diffusion_utils.py:58-69 (frequencies and the product are formed in fp32 there)."""
    half = embedding_dim // 2
    emb = math.log(max_positions) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = timesteps.float()[:, None] * emb[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)


def timestep_embedding(t, cfg, dtype):
    """get_timestep_embedding('sinusoidal'): x -> sinusoidal_embedding(scale * x, dim)  (diffusion_utils.py:87-94)"""
    return sinusoidal_embedding(cfg.embedding_scale * t, cfg.sigma_embed_dim).to(dtype)


def gaussian_smearing(dist, stop, num, dtype, start=0.0):
    """models/tensor_layers.py:171-181 (offset / coeff are built in fp32 like the module's buffers)."""
    offset = torch.linspace(start, stop, num)
    coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
    d = dist.reshape(-1, 1) - offset.to(dtype).reshape(1, -1)
    return torch.exp(coeff * torch.pow(d, 2))


def mlp2(x, P, prefix, i0, i1, act=torch.relu, bias=True):
    """nn.Sequential(Linear, act, Dropout(eval: identity), Linear) addressed as prefix.{i0,i1}"""
    h = F.linear(x, P[f'{prefix}.{i0}.weight'], P.get(f'{prefix}.{i0}.bias') if bias else None)
    h = act(h)
    return F.linear(h, P[f'{prefix}.{i1}.weight'], P.get(f'{prefix}.{i1}.bias') if bias else None)


def atom_encoder(x, P, prefix, n_cat):
    """models/layers.py:140-149 (AtomEncoder.forward)"""
    emb = 0
    for i in range(n_cat):
        emb = emb + P[f'{prefix}.atom_embedding_list.{i}.weight'][x[:, i].long()]
    return F.linear(torch.cat([emb, x[:, n_cat:]], dim=1),
                    P[f'{prefix}.additional_features_embedder.weight'], P[f'{prefix}.additional_features_embedder.bias'])


def _muls(irreps):
    m = {'0e': 0, '1o': 0, '1e': 0, '0o': 0}
    for mul, ir in o3.Irreps(irreps):
        m[str(ir)] = mul
    return m


def faster_tp_weight_shapes(in_irreps, out_irreps):
    """models/tensor_layers.py:58-63"""
    i, o = _muls(in_irreps), _muls(out_irreps)
    return {'0e': (i['0e'] + i['1o'], o['0e']), '1o': (i['0e'] + i['1o'] + i['1e'], o['1o']),
            '1e': (i['1o'] + i['1e'] + i['0o'], o['1e']), '0o': (i['1e'] + i['0o'], o['0o'])}


def faster_tp_weight_numel(in_irreps, out_irreps):
    return sum(a * b for a, b in faster_tp_weight_shapes(in_irreps, out_irreps).values())


def faster_tensor_product(x, sh, weight, in_irreps, out_irreps):
    """models/tensor_layers.py:65-116 (FasterTensorProduct.forward), restated.

    rows('0e') = [a*s0 ; (p.v)/sqrt3]      rows('1o') = [a (x) v ; p*s0 ; (q x v)/sqrt2]
    rows('1e') = [(p x v)/sqrt2 ; q*s0 ; c (x) v]      rows('0o') = [(q.v)/sqrt3 ; c*s0]
    out_key[k(,xyz)] = sum_i rows_key[i(,xyz)] * w_key[i,k] / sqrt(n_rows_key)
    """
    irr_in, irr_out = o3.Irreps(in_irreps), o3.Irreps(out_irreps)
    parts = {}
    for (mul, ir), sl in zip(irr_in, irr_in.slices()):
        v = x[..., sl]
        parts[str(ir)] = v.reshape(v.shape[:-1] + (mul, 3)) if ir.l == 1 else v
    s0, v = sh[..., 0:1], sh[..., 1:4]
    vv = v.unsqueeze(-2)
    rows = {'0e': [], '1o': [], '1e': [], '0o': []}
    if '0e' in parts:
        a = parts['0e']
        rows['0e'].append(a * s0)
        rows['1o'].append(a.unsqueeze(-1) * vv)
    if '1o' in parts:
        p = parts['1o']
        rows['0e'].append((p * vv).sum(-1) / np.sqrt(3))
        rows['1o'].append(p * s0.unsqueeze(-1))
        rows['1e'].append(torch.linalg.cross(p, vv.expand_as(p), dim=-1) / np.sqrt(2))
    if '1e' in parts:
        q = parts['1e']
        rows['1o'].append(torch.linalg.cross(q, vv.expand_as(q), dim=-1) / np.sqrt(2))
        rows['1e'].append(q * s0.unsqueeze(-1))
        rows['0o'].append((q * vv).sum(-1) / np.sqrt(3))
    if '0o' in parts:
        c = parts['0o']
        rows['1e'].append(c.unsqueeze(-1) * vv)
        rows['0o'].append(c * s0)
    shapes = faster_tp_weight_shapes(in_irreps, out_irreps)
    out, start = {}, 0
    for key in ('0e', '1o', '1e', '0o'):
        n_in, n_out = shapes[key]
        w = weight[..., start:start + n_in * n_out].reshape(weight.shape[:-1] + (n_in, n_out)) / np.sqrt(n_in)
        start += n_in * n_out
        if not rows[key]:
            continue
        if key in ('0e', '0o'):
            r = torch.cat(rows[key], dim=-1)
            out[key] = torch.matmul(r.unsqueeze(-2), w).squeeze(-2)
        else:
            r = torch.cat(rows[key], dim=-2)                              # [..., n_in, 3]
            o = (r.unsqueeze(-2) * w.unsqueeze(-1)).sum(-3)              # [..., n_out, 3]
            out[key] = o.reshape(o.shape[:-2] + (-1,))
    return torch.cat([out[str(ir)] for _, ir in irr_out], dim=-1)


def irreps_dim(irreps):
    return o3.Irreps(irreps).dim


def tp_conv_layer(P, prefix, node_attr, edge_index, edge_attr, edge_sh, in_irreps, sh_irreps, out_irreps,
                  residual=True, batch_norm=True, faster=False, edge_groups=1, out_nodes=None, eps=1e-5):
    """models/tensor_layers.py:147-168 (TensorProductConvLayer.forward, reduce='mean').

    edge_attr is a list of ``edge_groups`` tensors when edge_groups > 1 (one radial MLP per group,
    tensor_layers.py:140-143,154-155)."""
    out_size = irreps_dim(out_irreps)
    if edge_index.shape[1] == 0:
        out = torch.zeros((node_attr.shape[0], out_size), dtype=node_attr.dtype)
    else:
        edge_src, edge_dst = edge_index
        if edge_groups == 1:
            w = mlp2(edge_attr, P, f'{prefix}.fc', 0, 4)
        else:
            w = torch.cat([mlp2(edge_attr[g], P, f'{prefix}.fc.{g}', 0, 4) for g in range(edge_groups)], dim=0)
        if faster:
            tp = faster_tensor_product(node_attr[edge_dst], edge_sh, w, in_irreps, out_irreps)
        else:
            tp = o3.FullyConnectedTensorProduct(in_irreps, sh_irreps, out_irreps)(node_attr[edge_dst], edge_sh, w)
        out_nodes = out_nodes or node_attr.shape[0]
        out = scatter(tp, edge_src, dim=0, dim_size=out_nodes, reduce='mean')
        if batch_norm:
            out = o3.batch_norm_eval(out, out_irreps, P[f'{prefix}.batch_norm.weight'], P[f'{prefix}.batch_norm.bias'],
                                     P[f'{prefix}.batch_norm.running_mean'], P[f'{prefix}.batch_norm.running_var'], eps)
    if residual:
        out = out + F.pad(node_attr, (0, out.shape[-1] - node_attr.shape[-1]))
    return out


# ---------------------------------------------------------------------------------------------
# score-norm table lookups (utils/so3.py:91-95, utils/torus.py:79-83)
# ---------------------------------------------------------------------------------------------
def so3_score_norm(eps, exp_score_norms):
    """utils/so3.py:91-95; eps is the fp32 rot_sigma tensor -> numpy float32 arithmetic like the reference."""
    eps = eps.float().numpy() if torch.is_tensor(eps) else np.asarray(eps, dtype=np.float32)
    lo, hi, n = 0.01, 2.0, 1000
    idx = (np.log10(eps) - np.log10(lo)) / (np.log10(hi) - np.log10(lo)) * n
    idx = np.clip(np.around(idx).astype(int), a_min=0, a_max=n - 1)
    return torch.from_numpy(np.asarray(exp_score_norms)[idx]).float()


def torus_score_norm(sigma, score_norm_table):
    """utils/torus.py:79-83; sigma arrives as a float32 numpy array (score_model.py:306)."""
    sigma = np.asarray(sigma)
    lo, hi, n = 3e-3, 2.0, 5000
    s = np.log(sigma / np.pi)
    s = (s - np.log(lo)) / (np.log(hi) - np.log(lo)) * n
    s = np.round(np.clip(s, 0, n)).astype(int)
    return np.asarray(score_norm_table)[s]


# ---------------------------------------------------------------------------------------------
# graph builders + forward (models/score_model.py)
# ---------------------------------------------------------------------------------------------
def _sh(vec, lmax_or_irreps):
    return o3.spherical_harmonics(lmax_or_irreps, vec, normalize=True, normalization='component')


def build_lig_conv_graph(data, cfg, latent_h, dtype):
    """models/score_model.py:310-344"""
    lig = data['ligand']
    node_sigma_emb = timestep_embedding(lig.node_t['tr'], cfg, dtype)
    radius_edges = radius_graph(lig.pos, cfg.lig_max_radius, lig.batch)
    bonds = data['ligand', 'ligand']
    edge_index = torch.cat([bonds.edge_index, radius_edges], 1).long()
    edge_attr = torch.cat([bonds.edge_attr.to(dtype),
                           torch.zeros(radius_edges.shape[-1], cfg.in_lig_edge_features, dtype=dtype)], 0)
    edge_sigma_emb = node_sigma_emb[edge_index[0]]
    src, dst = edge_index
    edge_vec = lig.pos[dst] - lig.pos[src]
    edge_length_emb = gaussian_smearing(edge_vec.norm(dim=-1), cfg.lig_max_radius, cfg.distance_embed_dim, dtype)
    if latent_h is not None:
        node_latent = latent_h[0]
        edge_latent = torch.cat([node_latent[src], node_latent[dst]], 1)
        edge_attr = torch.cat([edge_attr, edge_sigma_emb, edge_length_emb, edge_latent], 1)
        node_attr = torch.cat([lig.x.to(dtype), node_sigma_emb, node_latent], 1)
    else:
        edge_attr = torch.cat([edge_attr, edge_sigma_emb, edge_length_emb], 1)
        node_attr = torch.cat([lig.x.to(dtype), node_sigma_emb], 1)
    return node_attr, edge_index, edge_attr, _sh(edge_vec, o3.Irreps.spherical_harmonics(cfg.sh_lmax)), node_sigma_emb


def build_rec_conv_graph(data, cfg, latent_h, dtype):
    """models/score_model.py:346-373"""
    rec = data['receptor']
    node_sigma_emb = timestep_embedding(rec.node_t['tr'], cfg, dtype)
    edge_index = data['receptor', 'receptor'].edge_index.long()
    src, dst = edge_index
    edge_vec = rec.pos[dst] - rec.pos[src]
    edge_length_emb = gaussian_smearing(edge_vec.norm(dim=-1), cfg.rec_max_radius, cfg.distance_embed_dim, dtype)
    edge_sigma_emb = node_sigma_emb[src]
    if latent_h is not None:
        node_latent = latent_h[1]
        edge_latent = torch.cat([node_latent[src], node_latent[dst]], 1)
        node_attr = torch.cat([rec.x.to(dtype), node_sigma_emb, node_latent], 1)
        edge_attr = torch.cat([edge_sigma_emb, edge_length_emb, edge_latent], 1)
    else:
        node_attr = torch.cat([rec.x.to(dtype), node_sigma_emb], 1)
        edge_attr = torch.cat([edge_sigma_emb, edge_length_emb], 1)
    return node_attr, edge_index, edge_attr, _sh(edge_vec, o3.Irreps.spherical_harmonics(cfg.sh_lmax))


def build_cross_conv_graph(data, cfg, cross_cutoff, lig_node_sigma_emb, latent_h, dtype):
    """models/score_model.py:375-408"""
    lig, rec = data['ligand'], data['receptor']
    if torch.is_tensor(cross_cutoff):
        edge_index = radius(rec.pos / cross_cutoff[rec.batch], lig.pos / cross_cutoff[lig.batch], 1,
                            rec.batch, lig.batch, max_num_neighbors=10000)
    else:
        edge_index = radius(rec.pos, lig.pos, cross_cutoff, rec.batch, lig.batch, max_num_neighbors=10000)
    src, dst = edge_index
    edge_vec = rec.pos[dst] - lig.pos[src]
    edge_length_emb = gaussian_smearing(edge_vec.norm(dim=-1), cfg.cross_max_distance, cfg.cross_distance_embed_dim, dtype)
    edge_sigma_emb = lig_node_sigma_emb[src]
    if latent_h is not None:
        edge_latent = torch.zeros(len(src), 2 * latent_h[0].shape[1], dtype=dtype)    # score_model.py:401
        edge_attr = torch.cat([edge_sigma_emb, edge_length_emb, edge_latent], 1)
    else:
        edge_attr = torch.cat([edge_sigma_emb, edge_length_emb], 1)
    return edge_index, edge_attr, _sh(edge_vec, o3.Irreps.spherical_harmonics(cfg.sh_lmax))


def embed(P, cfg, data, dtype=torch.float32, return_graph=False):
    """models/score_model.py:169-257 (TensorProductScoreModel.embed), latent_cross_attention=False."""
    ns = cfg.ns
    if cfg.latent_dim > 0:
        assert cfg.latent_vocab == 1, "oracle restates the equivariant-latent (vocab=1) branch only"
        latent_h = (data['ligand'].latent_h.to(dtype), data['receptor'].latent_h.to(dtype))
    else:
        latent_h = None
    if not cfg.confidence_mode:
        tr_sigma, rot_sigma, tor_sigma = t_to_sigma(*[data.complex_t[k] for k in ('tr', 'rot', 'tor')], cfg)
    else:      # score_model.py:186-189: complex_t IS the sigma in confidence_mode
        tr_sigma, rot_sigma, tor_sigma = [data.complex_t[k] for k in ('tr', 'rot', 'tor')]

    lig_node_attr, lig_edge_index, lig_edge_attr, lig_edge_sh, lig_sig = build_lig_conv_graph(data, cfg, latent_h, dtype)
    lig_node_attr = atom_encoder(lig_node_attr, P, 'lig_node_embedding', len(LIG_FEATURE_DIMS))
    lig_edge_attr = mlp2(lig_edge_attr, P, 'lig_edge_embedding', 0, 3)

    rec_node_attr, rec_edge_index, rec_edge_attr, rec_edge_sh = build_rec_conv_graph(data, cfg, latent_h, dtype)
    rec_node_attr = atom_encoder(rec_node_attr, P, 'rec_node_embedding', len(REC_FEATURE_DIMS))
    rec_edge_attr = mlp2(rec_edge_attr, P, 'rec_edge_embedding', 0, 3)

    cross_cutoff = (tr_sigma * 3 + 20).unsqueeze(1).to(dtype) if cfg.dynamic_max_cross else cfg.cross_max_distance
    lr_edge_index, lr_edge_attr, lr_edge_sh = build_cross_conv_graph(data, cfg, cross_cutoff, lig_sig, latent_h, dtype)
    lr_edge_attr = mlp2(lr_edge_attr, P, 'cross_edge_embedding', 0, 3)

    if cfg.latent_droprate > 0:   # score_model.py:209-215
        ul, ur = data['ligand'].unconditional.to(dtype), data['receptor'].unconditional.to(dtype)
        lig_node_attr = lig_node_attr + ul * P['lig_node_unconditional_embedding']
        rec_node_attr = rec_node_attr + ur * P['rec_node_unconditional_embedding']
        lig_edge_attr = lig_edge_attr + ul[lig_edge_index[0]] * P['lig_edge_unconditional_embedding']
        rec_edge_attr = rec_edge_attr + ur[rec_edge_index[0]] * P['rec_edge_unconditional_embedding']
        lr_edge_attr = lr_edge_attr + ul[lr_edge_index[0]] * P['cross_edge_unconditional_embedding']

    n_lig = len(lig_node_attr)
    node_attr = torch.cat([lig_node_attr, rec_node_attr], dim=0)
    lr_edge_index = torch.stack([lr_edge_index[0], lr_edge_index[1] + n_lig])
    edge_index = torch.cat([lig_edge_index, lr_edge_index, rec_edge_index + n_lig, torch.flip(lr_edge_index, dims=[0])], dim=1)
    edge_attr = torch.cat([lig_edge_attr, lr_edge_attr, rec_edge_attr, lr_edge_attr], dim=0)
    edge_sh = torch.cat([lig_edge_sh, lr_edge_sh, rec_edge_sh, lr_edge_sh], dim=0)
    s1 = lig_edge_index.shape[1]
    s2 = s1 + lr_edge_index.shape[1]
    s3 = s2 + rec_edge_index.shape[1]
    sh_irreps = o3.Irreps.spherical_harmonics(cfg.sh_lmax)
    graph = dict(edge_index=edge_index, edge_emb=edge_attr, edge_sh=edge_sh, splits=(s1, s2, s3), x0=node_attr, layers=[])
    for l in range(cfg.num_conv_layers):
        ea = torch.cat([edge_attr, node_attr[edge_index[0], :ns], node_attr[edge_index[1], :ns]], -1)
        ea = [ea[:s1], ea[s1:s2], ea[s2:s3], ea[s3:]]
        in_irreps, out_irreps = cfg.conv_irreps(l)
        node_attr = tp_conv_layer(P, f'conv_layers.{l}', node_attr, edge_index, ea, edge_sh, in_irreps, sh_irreps,
                                  out_irreps, residual=True, batch_norm=cfg.batch_norm,
                                  faster=(cfg.sh_lmax == 1), edge_groups=4)
        graph['layers'].append(node_attr)
    out = (node_attr[:n_lig], node_attr[n_lig:], tr_sigma, rot_sigma, tor_sigma, lig_sig)
    return out + (graph,) if return_graph else out


def build_center_conv_graph(data, cfg, lig_node_sigma_emb, dtype):
    """models/score_model.py:410-423"""
    lig = data['ligand']
    n = len(lig.batch)
    edge_index = torch.stack([lig.batch, torch.arange(n)], 0)
    center = torch.zeros((data.num_graphs, 3), dtype=dtype)
    center.index_add_(0, lig.batch, lig.pos)
    center = center / torch.bincount(lig.batch, minlength=data.num_graphs).unsqueeze(1)
    edge_vec = lig.pos[edge_index[1]] - center[edge_index[0]]
    edge_attr = gaussian_smearing(edge_vec.norm(dim=-1), cfg.center_max_distance, cfg.distance_embed_dim, dtype)
    edge_attr = torch.cat([edge_attr, lig_node_sigma_emb[edge_index[1]]], 1)
    return edge_index, edge_attr, _sh(edge_vec, o3.Irreps.spherical_harmonics(cfg.sh_lmax))


def build_bond_conv_graph(P, data, cfg, dtype):
    """models/score_model.py:425-438"""
    lig = data['ligand']
    bonds = data['ligand', 'ligand'].edge_index[:, lig.edge_mask].long()
    bond_pos = (lig.pos[bonds[0]] + lig.pos[bonds[1]]) / 2
    bond_batch = lig.batch[bonds[0]]
    edge_index = radius(lig.pos, bond_pos, cfg.lig_max_radius, batch_x=lig.batch, batch_y=bond_batch)
    edge_vec = lig.pos[edge_index[1]] - bond_pos[edge_index[0]]
    edge_attr = gaussian_smearing(edge_vec.norm(dim=-1), cfg.lig_max_radius, cfg.distance_embed_dim, dtype)
    edge_attr = mlp2(edge_attr, P, 'final_edge_embedding', 0, 3)
    return bonds, edge_index, edge_attr, _sh(edge_vec, o3.Irreps.spherical_harmonics(cfg.sh_lmax))


def score_model_forward(P, cfg, data, so3_table, torus_table, dtype=torch.float32, return_intermediates=False):
    """models/score_model.py:259-308 (TensorProductScoreModel.forward, confidence_mode=False).

    ``data`` is a graph_lite batch with node_t / complex_t set (utils/diffusion_utils.py:101-117).
    Returns (tr_pred [B,3], rot_pred [B,3], tor_pred [sum R])."""
    P = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in P.items()}
    for nt in ('ligand', 'receptor'):
        data[nt].pos = data[nt].pos.to(dtype)
    ns = cfg.ns
    lig_node_attr, rec_node_attr, tr_sigma, rot_sigma, tor_sigma, lig_sig, graph = embed(P, cfg, data, dtype, True)
    sh_irreps = o3.Irreps.spherical_harmonics(cfg.sh_lmax)
    conv_out = cfg.conv_irreps(cfg.num_conv_layers - 1)[1]

    c_edge_index, c_edge_attr, c_edge_sh = build_center_conv_graph(data, cfg, lig_sig, dtype)
    c_edge_attr = mlp2(c_edge_attr, P, 'center_edge_embedding', 0, 3)
    c_edge_attr = torch.cat([c_edge_attr, lig_node_attr[c_edge_index[1], :ns]], -1)
    global_pred = tp_conv_layer(P, 'final_conv', lig_node_attr, c_edge_index, c_edge_attr, c_edge_sh, conv_out, sh_irreps,
                                '2x1o + 2x1e', residual=False, batch_norm=cfg.batch_norm, faster=False,
                                out_nodes=data.num_graphs)
    tr_pred = global_pred[:, :3] + global_pred[:, 6:9]
    rot_pred = global_pred[:, 3:6] + global_pred[:, 9:]
    graph_sigma_emb = timestep_embedding(data.complex_t['tr'], cfg, dtype)

    def final_layer(prefix, v):   # Sequential(Linear, Dropout, ReLU, Linear)  score_model.py:142-143
        return mlp2(v, P, prefix, 0, 3)

    tr_norm = torch.linalg.vector_norm(tr_pred, dim=1).unsqueeze(1)
    tr_pred = tr_pred / tr_norm * final_layer('tr_final_layer', torch.cat([tr_norm, graph_sigma_emb], dim=1))
    rot_norm = torch.linalg.vector_norm(rot_pred, dim=1).unsqueeze(1)
    rot_pred = rot_pred / rot_norm * final_layer('rot_final_layer', torch.cat([rot_norm, graph_sigma_emb], dim=1))
    if cfg.scale_by_sigma:
        tr_pred = tr_pred / tr_sigma.to(dtype).unsqueeze(1)
        rot_pred = rot_pred * so3_score_norm(rot_sigma, so3_table).to(dtype).unsqueeze(1)
    inter = dict(lig_node_attr=lig_node_attr, rec_node_attr=rec_node_attr, global_pred=global_pred, graph=graph)
    if cfg.no_torsion or data['ligand'].edge_mask.sum() == 0:
        out = (tr_pred, rot_pred, torch.empty(0, dtype=dtype))
        return out + (inter,) if return_intermediates else out

    lig = data['ligand']
    tor_bonds, tor_edge_index, tor_edge_attr, tor_edge_sh = build_bond_conv_graph(P, data, cfg, dtype)
    tor_bond_vec = lig.pos[tor_bonds[1]] - lig.pos[tor_bonds[0]]
    tor_bond_attr = lig_node_attr[tor_bonds[0]] + lig_node_attr[tor_bonds[1]]
    tor_bonds_sh = _sh(tor_bond_vec, '2e')
    tp_tor = o3.FullTensorProduct(sh_irreps, '2e')
    tor_edge_sh = tp_tor(tor_edge_sh, tor_bonds_sh[tor_edge_index[0]])
    tor_edge_attr = torch.cat([tor_edge_attr, lig_node_attr[tor_edge_index[1], :ns],
                               tor_bond_attr[tor_edge_index[0], :ns]], -1)
    tor_pred = tp_conv_layer(P, 'tor_bond_conv', lig_node_attr, tor_edge_index, tor_edge_attr, tor_edge_sh, conv_out,
                             tp_tor.irreps_out, f'{ns}x0o + {ns}x0e', residual=False, batch_norm=cfg.batch_norm,
                             faster=False, out_nodes=int(lig.edge_mask.sum()))
    inter['tor_conv'] = tor_pred
    h = torch.tanh(F.linear(tor_pred, P['tor_final_layer.0.weight']))
    tor_pred = F.linear(h, P['tor_final_layer.3.weight']).squeeze(1)
    edge_sigma = tor_sigma[lig.batch][data['ligand', 'ligand'].edge_index[0]][lig.edge_mask]
    if cfg.scale_by_sigma:
        tor_pred = tor_pred * torch.sqrt(torch.tensor(torus_score_norm(edge_sigma.float().numpy(), torus_table)).float()).to(dtype)
    out = (tr_pred, rot_pred, tor_pred)
    return out + (inter,) if return_intermediates else out


# ---------------------------------------------------------------------------------------------
# seeded synthetic weights in the reference state_dict layout (SURVEY.md §8b, §8d "Weights")
# ---------------------------------------------------------------------------------------------
def confidence_forward(P, cfg, data, dtype=torch.float32):
    """models/score_model.py:259-266 (TensorProductScoreModel.forward, confidence_mode=True): the pooled ligand scalars through the
    confidence_predictor (:110-121: Linear, BatchNorm1d, ReLU, Dropout, Linear, BatchNorm1d, ReLU, Dropout, Linear; eval mode)."""
    assert cfg.confidence_mode
    ns = cfg.ns
    lig_node_attr = embed(P, cfg, data, dtype)[0]
    scalar = torch.cat([lig_node_attr[:, :ns], lig_node_attr[:, -ns:]], dim=1) if cfg.num_conv_layers >= 3 else lig_node_attr[:, :ns]
    h = scatter(scalar, data['ligand'].batch, dim=0, dim_size=data.num_graphs, reduce='mean')
    for i_lin, i_bn in ((0, 1), (4, 5)):
        h = torch.nn.functional.linear(h, P[f'confidence_predictor.{i_lin}.weight'].to(dtype), P[f'confidence_predictor.{i_lin}.bias'].to(dtype))
        if not cfg.confidence_no_batchnorm:
            k = f'confidence_predictor.{i_bn}'
            h = (h - P[f'{k}.running_mean'].to(dtype)) / torch.sqrt(P[f'{k}.running_var'].to(dtype) + 1e-5) * P[f'{k}.weight'].to(dtype) + P[f'{k}.bias'].to(dtype)
        h = torch.relu(h)
    return torch.nn.functional.linear(h, P['confidence_predictor.8.weight'].to(dtype), P['confidence_predictor.8.bias'].to(dtype)).squeeze(dim=-1)


def state_dict_spec(cfg):
    """name -> shape for ``score_model.state_dict()`` (parameters + BN buffers, no e3nn-internal buffers)."""
    ns, sd, dd, cd = cfg.ns, cfg.sigma_embed_dim, cfg.distance_embed_dim, cfg.cross_distance_embed_dim
    lat_n, lat_e = cfg.latent_dim * cfg.latent_vocab, cfg.latent_dim * max(cfg.latent_vocab, 2)
    spec = {}

    def lin(name, o, i, bias=True):
        spec[f'{name}.weight'] = (o, i)
        if bias:
            spec[f'{name}.bias'] = (o,)

    for i, d in enumerate(LIG_FEATURE_DIMS):
        spec[f'lig_node_embedding.atom_embedding_list.{i}.weight'] = (d, ns)
    lin('lig_node_embedding.additional_features_embedder', ns, ns + sd + lat_n)
    lin('lig_edge_embedding.0', ns, cfg.in_lig_edge_features + sd + dd + lat_e)
    lin('lig_edge_embedding.3', ns, ns)
    spec['rec_node_embedding.atom_embedding_list.0.weight'] = (REC_FEATURE_DIMS[0], ns)
    lin('rec_node_embedding.additional_features_embedder', ns, ns + sd + cfg.lm_embedding_dim + lat_n)
    lin('rec_edge_embedding.0', ns, sd + dd + lat_e)
    lin('rec_edge_embedding.3', ns, ns)
    lin('cross_edge_embedding.0', ns, sd + cd + lat_e)
    lin('cross_edge_embedding.3', ns, ns)
    if cfg.latent_droprate > 0:
        for k in ('lig_node', 'rec_node', 'lig_edge', 'rec_edge', 'cross_edge'):
            spec[f'{k}_unconditional_embedding'] = (1, ns)
    for k, n in (('lig', dd), ('rec', dd), ('cross', cd)) + (() if cfg.confidence_mode else (('center', dd),)):
        spec[f'{k}_distance_expansion.offset'] = (n,)
    for l in range(cfg.num_conv_layers):
        i_irr, o_irr = cfg.conv_irreps(l)
        W = faster_tp_weight_numel(i_irr, o_irr)
        for g in range(4):
            lin(f'conv_layers.{l}.fc.{g}.0', 3 * ns, 3 * ns)
            lin(f'conv_layers.{l}.fc.{g}.4', W, 3 * ns)
        _bn_spec(spec, f'conv_layers.{l}.batch_norm', o_irr)
    if cfg.confidence_mode:      # score_model.py:110-121
        lin('confidence_predictor.0', ns, 2 * ns if cfg.num_conv_layers >= 3 else ns)
        lin('confidence_predictor.4', ns, ns)
        lin('confidence_predictor.8', cfg.num_confidence_outputs, ns)
        if not cfg.confidence_no_batchnorm:
            for i in (1, 5):
                for k in ('weight', 'bias', 'running_mean', 'running_var'):
                    spec[f'confidence_predictor.{i}.{k}'] = (ns,)
        return spec
    lin('center_edge_embedding.0', ns, dd + sd)
    lin('center_edge_embedding.3', ns, ns)
    conv_out = cfg.conv_irreps(cfg.num_conv_layers - 1)[1]
    sh = o3.Irreps.spherical_harmonics(cfg.sh_lmax)
    lin('final_conv.fc.0', 2 * ns, 2 * ns)
    lin('final_conv.fc.4', o3.FullyConnectedTensorProduct(conv_out, sh, '2x1o + 2x1e').weight_numel, 2 * ns)
    _bn_spec(spec, 'final_conv.batch_norm', '2x1o + 2x1e')
    lin('tr_final_layer.0', ns, 1 + sd)
    lin('tr_final_layer.3', 1, ns)
    lin('rot_final_layer.0', ns, 1 + sd)
    lin('rot_final_layer.3', 1, ns)
    if not cfg.no_torsion:
        lin('final_edge_embedding.0', ns, dd)
        lin('final_edge_embedding.3', ns, ns)
        tor_sh = o3.FullTensorProduct(sh, '2e').irreps_out
        lin('tor_bond_conv.fc.0', 3 * ns, 3 * ns)
        lin('tor_bond_conv.fc.4', o3.FullyConnectedTensorProduct(conv_out, tor_sh, f'{ns}x0o + {ns}x0e').weight_numel, 3 * ns)
        _bn_spec(spec, 'tor_bond_conv.batch_norm', f'{ns}x0o + {ns}x0e')
        lin('tor_final_layer.0', ns, 2 * ns, bias=False)
        lin('tor_final_layer.3', 1, ns, bias=False)
    return spec


def _bn_spec(spec, name, irreps):
    irr = o3.Irreps(irreps)
    nf = irr.num_irreps
    nsc = sum(mul for mul, ir in irr if ir.is_scalar())
    spec[f'{name}.weight'] = (nf,)
    spec[f'{name}.bias'] = (nsc,)
    spec[f'{name}.running_mean'] = (nsc,)
    spec[f'{name}.running_var'] = (nf,)


def random_state_dict(cfg, seed=0):
    """PyTorch-default-style init (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for Linear, xavier-uniform
    embeddings) with randomised BatchNorm statistics, under a private generator."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    spec = state_dict_spec(cfg)
    for name, shape in spec.items():
        if name.endswith('distance_expansion.offset'):
            stop = {'lig': cfg.lig_max_radius, 'rec': cfg.rec_max_radius, 'cross': cfg.cross_max_distance,
                    'center': cfg.center_max_distance}[name.split('_')[0]]
            P[name] = torch.linspace(0.0, stop, shape[0])
        elif 'atom_embedding_list' in name:
            a = math.sqrt(6.0 / (shape[0] + shape[1]))
            P[name] = (torch.rand(shape, generator=g) * 2 - 1) * a
        elif name.endswith('unconditional_embedding'):
            P[name] = torch.randn(shape, generator=g) * 0.1
        elif '.batch_norm.' in name or (name.startswith('confidence_predictor') and name.split('.')[1] in ('1', '5')):
            if name.endswith('running_mean'):
                P[name] = torch.randn(shape, generator=g) * 0.1
            elif name.endswith('running_var') or name.endswith('weight'):
                P[name] = torch.rand(shape, generator=g) + 0.5
            else:
                P[name] = torch.randn(shape, generator=g) * 0.1
        else:
            fan_in = shape[1] if name.endswith('weight') else spec[name[:-4] + 'weight'][1]
            b = 1.0 / math.sqrt(fan_in)
            P[name] = (torch.rand(shape, generator=g) * 2 - 1) * b
    return P


def random_conv_layer_params(cfg, l, seed, batch_norm=True):
    """Seeded parameters for ONE TensorProductConvLayer (keys as in its own state_dict:
    fc.{g}.{0,4}.{weight,bias}, batch_norm.*) — lets the conv-layer goldens store a seed instead of weights."""
    g = torch.Generator().manual_seed(seed)
    i_irr, o_irr = cfg.conv_irreps(l)
    W = faster_tp_weight_numel(i_irr, o_irr)
    ne = 3 * cfg.ns
    P = {}
    for grp in range(4):
        P[f'fc.{grp}.0.weight'] = torch.randn(ne, ne, generator=g) / math.sqrt(ne)
        P[f'fc.{grp}.0.bias'] = torch.randn(ne, generator=g) * 0.2
        P[f'fc.{grp}.4.weight'] = torch.randn(W, ne, generator=g) / math.sqrt(ne)
        P[f'fc.{grp}.4.bias'] = torch.randn(W, generator=g) * 0.2
    if batch_norm:
        spec = {}
        _bn_spec(spec, 'batch_norm', o_irr)
        P['batch_norm.weight'] = torch.rand(spec['batch_norm.weight'], generator=g) + 0.5
        P['batch_norm.bias'] = torch.randn(spec['batch_norm.bias'], generator=g) * 0.1
        P['batch_norm.running_mean'] = torch.randn(spec['batch_norm.running_mean'], generator=g) * 0.1
        P['batch_norm.running_var'] = torch.rand(spec['batch_norm.running_var'], generator=g) + 0.5
    return P
