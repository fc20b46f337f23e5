"""Synthetic PDBBind-shaped complexes (no dataset / checkpoints are available offline).

Emits plain numpy arrays in the layout ``datasets_utils/process_mols.py`` produces for one
complex, following the recipe of SURVEY.md §8(d):

* receptor: N_r C-alpha points rejection-sampled in a ball at protein density
  (135 A^3 / residue, min separation 3.8 A), centred; kNN contact edges with the rule of
  ``get_calpha_graph`` (reference datasets_utils/process_mols.py:337-353: all residues
  closer than ``cutoff``; the ``max_neighbor`` nearest if there are more; at least one);
  features [residue id in 0..37 | ESM-like N(0,1)^1280].
* ligand: two six-rings joined by a linker plus a tail and single-atom substituents,
  1.5 A bonds, N_l in 20..40 atoms, 4..8 rotatable bonds; 16 categorical atom features
  within ``LIG_FEATURE_DIMS``; bond-type one-hots; ``edge_mask`` / ``mask_rotate`` by the
  rule of ``get_transformation_mask`` (reference utils/torsion.py:15-45).
"""
import numpy as np

# lengths of the categorical feature vocabularies (reference datasets_utils/process_mols.py:62-79, 88-90)
LIG_FEATURE_DIMS = (119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2)
REC_RESIDUE_FEATURE_DIMS = (38,)
ESM_DIM = 1280


def _rand_unit(rng):
    v = rng.normal(size=3)
    return v / np.linalg.norm(v)


def make_receptor(rng, n_res, cutoff=15.0, max_neighbor=24, esm_dim=ESM_DIM):
    radius = (3.0 * 135.0 * n_res / (4.0 * np.pi)) ** (1.0 / 3.0)
    pts = np.zeros((0, 3))
    cell = 3.8
    grid = {}

    def key(p):
        return tuple(np.floor(p / cell).astype(int))

    tries = 0
    out = []
    while len(out) < n_res:
        tries += 1
        if tries > 400 * n_res:
            radius *= 1.05
            tries = 0
        p = rng.uniform(-radius, radius, size=3)
        if p @ p > radius * radius:
            continue
        k = key(p)
        ok = True
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    for q in grid.get((k[0] + dx, k[1] + dy, k[2] + dz), ()):
                        if np.sum((out[q] - p) ** 2) < 3.8 ** 2:
                            ok = False
        if ok:
            grid.setdefault(k, []).append(len(out))
            out.append(p)
    pos = np.asarray(out)
    pos = pos - pos.mean(0, keepdims=True)
    d = np.linalg.norm(pos[:, None] - pos[None], axis=-1)
    src, dst = [], []
    for i in range(n_res):
        nb = list(np.where(d[i] < cutoff)[0])
        nb.remove(i)
        if max_neighbor is not None and len(nb) > max_neighbor:
            nb = list(np.argsort(d[i]))[1:max_neighbor + 1]
        if len(nb) == 0:
            nb = list(np.argsort(d[i]))[1:2]
        src += [i] * len(nb)
        dst += [int(j) for j in nb]
    x = np.concatenate([rng.integers(0, REC_RESIDUE_FEATURE_DIMS[0], size=(n_res, 1)).astype(np.float32),
                        rng.normal(size=(n_res, esm_dim)).astype(np.float32)], axis=1)
    return dict(rec_x=x, rec_pos=pos.astype(np.float32), rec_edge_index=np.asarray([src, dst], dtype=np.int64))


def _components_without(n, bonds, skip):
    adj = [[] for _ in range(n)]
    for bi, (a, b) in enumerate(bonds):
        if bi == skip:
            continue
        adj[a].append(b)
        adj[b].append(a)
    seen = [-1] * n
    comps = []
    for s in range(n):
        if seen[s] >= 0:
            continue
        stack, comp = [s], []
        seen[s] = len(comps)
        while stack:
            u = stack.pop()
            comp.append(u)
            for v in adj[u]:
                if seen[v] < 0:
                    seen[v] = len(comps)
                    stack.append(v)
        comps.append(comp)
    return comps


def transformation_mask(n_atoms, bonds):
    """edge_mask over the 2*len(bonds) directed bonds and mask_rotate [R, n_atoms]
    (rule of reference utils/torsion.py:15-45: directed pair (u->v, v->u) per bond; the direction
    whose head v lies on the smaller, >1 atom, side is the rotatable one)."""
    to_rotate = []
    for bi, (u, v) in enumerate(bonds):
        comps = _components_without(n_atoms, bonds, bi)
        if len(comps) > 1:
            l = sorted(comps, key=len)[0]
            if len(l) > 1:
                if u in l:
                    to_rotate += [[], l]
                else:
                    to_rotate += [l, []]
                continue
        to_rotate += [[], []]
    mask_edges = np.asarray([len(l) > 0 for l in to_rotate], dtype=bool)
    mask_rotate = np.zeros((int(mask_edges.sum()), n_atoms), dtype=bool)
    idx = 0
    for i, l in enumerate(to_rotate):
        if mask_edges[i]:
            mask_rotate[idx][np.asarray(l, dtype=int)] = True
            idx += 1
    return mask_edges, mask_rotate


def make_ligand(rng, n_atoms=None):
    if n_atoms is None:
        n_atoms = int(rng.integers(20, 41))
    k1, k2 = int(rng.integers(2, 5)), int(rng.integers(2, 5))
    n_atoms = max(n_atoms, 12 + k1 + k2)
    pos, bonds = [], []

    def clash(p, exclude=()):
        if not pos:
            return False
        d = np.asarray(pos) - p                       # all atoms placed so far at once (a Python loop over them made an 80-atom ligand take 8 s)
        near = np.sqrt((d * d).sum(axis=1)) < 2.0
        for j in exclude:
            if 0 <= j < len(near):
                near[j] = False
        return bool(near.any())

    def grow(parent, prev_dir):
        for _ in range(2000):
            d = _rand_unit(rng)
            if prev_dir is not None and d @ prev_dir < -0.2:  # avoid folding straight back
                continue
            p = pos[parent] + 1.5 * d
            if not clash(p, exclude=(parent,)):
                pos.append(p)
                bonds.append((parent, len(pos) - 1))
                return len(pos) - 1, d
        raise RuntimeError('ligand growth failed')

    def ring(parent, prev_dir):
        """planar hexagon a0..a5 (side 1.5) attached to `parent` through a0; returns (a3, out dir)."""
        for attempt in range(2000):
            d = prev_dir if (attempt == 0 and prev_dir is not None) else _rand_unit(rng)
            n = np.cross(d, _rand_unit(rng))
            n /= np.linalg.norm(n)
            e2 = np.cross(n, d)
            a0 = (pos[parent] + 1.5 * d) if parent is not None else np.zeros(3)
            centre = a0 + 1.5 * d
            verts = [centre + 1.5 * (np.cos(np.pi + k * np.pi / 3) * d + np.sin(np.pi + k * np.pi / 3) * e2)
                     for k in range(6)]
            if any(clash(v, exclude=(parent,) if parent is not None else ()) for v in verts):
                continue
            base = len(pos)
            pos.extend(verts)
            if parent is not None:
                bonds.append((parent, base))
            for k in range(6):
                bonds.append((base + k, base + (k + 1) % 6))
            return base + 3, d
        raise RuntimeError('ring placement failed')

    cur, d = ring(None, None)
    for _ in range(k1):
        cur, d = grow(cur, d)
    cur, d = ring(cur, d)
    for _ in range(k2):
        cur, d = grow(cur, d)
    heavy = list(range(len(pos)))
    failed = 0
    while len(pos) < n_atoms:
        parent = int(rng.choice(heavy))
        try:
            grow(parent, None)
            failed = 0
        except RuntimeError:
            failed += 1
            if failed >= 8:       # the backbone is saturated (large n_atoms): let the side chains grow on (never reached for 20-40 atoms)
                heavy = list(range(len(pos)))
                failed = 0
            continue
    n = len(pos)
    pos = np.asarray(pos)
    # randomise the atom order so that topology is not index-sorted
    perm = rng.permutation(n)
    inv = np.argsort(perm)
    pos = pos[perm]
    bonds = [(int(inv[a]), int(inv[b])) for a, b in bonds]
    edge_mask, mask_rotate = transformation_mask(n, bonds)
    ei = np.zeros((2, 2 * len(bonds)), dtype=np.int64)
    ea = np.zeros((2 * len(bonds), 4), dtype=np.float32)
    for bi, (a, b) in enumerate(bonds):
        ei[:, 2 * bi] = (a, b)
        ei[:, 2 * bi + 1] = (b, a)
        t = int(rng.integers(0, 4))
        ea[2 * bi, t] = 1.0
        ea[2 * bi + 1, t] = 1.0
    x = np.stack([rng.integers(0, dmax, size=n) for dmax in LIG_FEATURE_DIMS], axis=1).astype(np.int64)
    return dict(lig_x=x, lig_pos=pos.astype(np.float32), bond_index=ei, bond_attr=ea,
                edge_mask=edge_mask, mask_rotate=mask_rotate)


REC_ATOM_FEATURE_DIMS = (38, 119, 23, 38)     # process_mols.py:81-86 (residue type, atomic number, atom type 2, atom type 3)


def add_receptor_atoms(c, rng, atom_radius=5.0, atom_max_neighbors=8, atoms_per_residue=(4, 12)):
    """All-atom receptor level of the confidence model's graphs (datasets_utils/process_mols.py:383-477 layout):
    ``atom_x`` [n_atom,4] categorical ids, ``atom_pos`` [n_atom,3] (heavy atoms within ~3.5 A of their C-alpha, 1.2 A apart),
    ``atom_edge_index`` [2,E_aa] = radius_graph(atom_pos, atom_radius, max_num_neighbors) (directed j->i pairs, per target
    the first ``max`` neighbours in index order) and ``atom_rec_index`` [2,n_atom] = (atom, its residue)."""
    rec_pos, res_id = c['rec_pos'], c['rec_x'][:, 0].astype(np.int64)
    pos, owner = [], []
    for r in range(rec_pos.shape[0]):
        k = int(rng.integers(atoms_per_residue[0], atoms_per_residue[1] + 1))
        mine = [rec_pos[r].astype(np.float64)]              # the C-alpha itself is the first atom
        tries = 0
        while len(mine) < k and tries < 200:
            tries += 1
            p = rec_pos[r] + _rand_unit(rng) * rng.uniform(1.3, 3.5)
            if all(np.linalg.norm(p - q) > 1.2 for q in mine):
                mine.append(p)
        pos += mine
        owner += [r] * len(mine)
    pos, owner = np.asarray(pos, np.float32), np.asarray(owner, np.int64)
    n = pos.shape[0]
    x = np.stack([res_id[owner], rng.integers(0, REC_ATOM_FEATURE_DIMS[1], n), rng.integers(0, REC_ATOM_FEATURE_DIMS[2], n),
                  rng.integers(0, REC_ATOM_FEATURE_DIMS[3], n)], axis=1).astype(np.int64)
    # radius_graph: for every target i its neighbours j (ascending index, first max_num_neighbors), edge (j -> i) stored as [j; i]
    src, dst = [], []
    order = np.argsort(pos[:, 0], kind='stable')
    xs = pos[order, 0]
    for i in range(n):
        lo, hi = np.searchsorted(xs, pos[i, 0] - atom_radius), np.searchsorted(xs, pos[i, 0] + atom_radius)
        cand = np.sort(order[lo:hi])
        d = np.linalg.norm(pos[cand] - pos[i], axis=1)
        nb = cand[(d < atom_radius) & (cand != i)][:atom_max_neighbors]
        src += [int(j) for j in nb]
        dst += [i] * len(nb)
    c['atom_x'], c['atom_pos'] = x, pos
    c['atom_edge_index'] = np.asarray([src, dst], dtype=np.int64)
    c['atom_rec_index'] = np.asarray([np.arange(n), owner], dtype=np.int64)
    return c


def timesplit_shape(seed):
    """(n_res, n_lig) of complex `seed` of the timesplit-SHAPED synthetic set (bench.py --complexes 363).  The reference's test split
    (data/splits/timesplit_test, 363 PDB ids, README.md:20) mixes receptors from under a hundred to a few thousand residues - every chain within reach of
    the ligand is kept, datasets_utils/process_mols.py:411-429 - and ligands of ~10-80 heavy atoms.  Without the structures here the sizes are DRAWN:
    residues log-normal with median 350 and sigma 0.65 (mean ~430, 5 % above ~1 000, 0.1 % above ~2 600), clipped to [60, 3 000]; ligand atoms uniform
    in [10, 80] (the same draw as rounds 4 / 5, so a seed keeps its ligand).  The choice is stated next to every number that rests on it."""
    n_lig = int(np.random.default_rng(7000 + seed).integers(10, 81))
    n_res = int(np.clip(np.rint(np.exp(np.random.default_rng(9000 + seed).normal(np.log(350.0), 0.65))), 60, 3000))
    return n_res, n_lig


def make_complex(seed, n_res=300, n_lig=None, cutoff=15.0, max_neighbor=24, esm_dim=ESM_DIM):
    """One synthetic complex as a dict of numpy arrays; ligand centred on a random pocket point
    inside the receptor ball (coordinates are receptor-centred like pdbbind.py:341-347)."""
    rng = np.random.default_rng(seed)
    rec = make_receptor(rng, n_res, cutoff, max_neighbor, esm_dim)
    lig = make_ligand(rng, n_lig)
    rad = np.linalg.norm(rec['rec_pos'], axis=1).max()
    pocket = _rand_unit(rng) * rng.uniform(0.3, 0.8) * rad
    lig['lig_pos'] = (lig['lig_pos'] - lig['lig_pos'].mean(0, keepdims=True) + pocket).astype(np.float32)
    out = dict(rec)
    out.update(lig)
    out['original_center'] = np.zeros((1, 3), dtype=np.float32)
    out['name'] = f'synthetic_{seed}'
    return out


def score_model_state_dict_spec(ns=24, nv=6, num_conv_layers=5, sigma=32, dist=32, lm=1280, latent_dim=0, latent_droprate=0.0,
                                confidence_mode=False, num_confidence_outputs=1, confidence_no_batchnorm=False):
    """name -> shape of the DiffDock-S ``score_model.state_dict()`` (171 tensors / 2 107 134 elements); with
    latent_dim = 2, latent_droprate > 0 the DisCo-DiffDock-S layout (176 tensors / 2 107 638 elements: + latent_dim node columns,
    + 2 latent_dim edge columns, five unconditional embeddings; models/score_model.py:46-62)."""
    spec = {}
    ld = latent_dim

    def lin(name, o, i, bias=True):
        spec[f'{name}.weight'] = (o, i)
        if bias:
            spec[f'{name}.bias'] = (o,)

    for i, d in enumerate(LIG_FEATURE_DIMS):
        spec[f'lig_node_embedding.atom_embedding_list.{i}.weight'] = (d, ns)
    lin('lig_node_embedding.additional_features_embedder', ns, ns + sigma + ld)
    lin('lig_edge_embedding.0', ns, 4 + sigma + dist + 2 * ld)
    lin('lig_edge_embedding.3', ns, ns)
    spec['rec_node_embedding.atom_embedding_list.0.weight'] = (REC_RESIDUE_FEATURE_DIMS[0], ns)
    lin('rec_node_embedding.additional_features_embedder', ns, ns + sigma + lm + ld)
    lin('rec_edge_embedding.0', ns, sigma + dist + 2 * ld)
    lin('rec_edge_embedding.3', ns, ns)
    lin('cross_edge_embedding.0', ns, sigma + dist + 2 * ld)
    lin('cross_edge_embedding.3', ns, ns)
    for k in ('lig', 'rec', 'cross') + (() if confidence_mode else ('center',)):
        spec[f'{k}_distance_expansion.offset'] = (dist,)
    seq = [(ns, 0, 0, 0), (ns, nv, 0, 0), (ns, nv, nv, 0), (ns, nv, nv, ns)]
    for l in range(num_conv_layers):
        i, o = seq[min(l, 3)], seq[min(l + 1, 3)]
        W = (i[0] + i[1]) * o[0] + (i[0] + i[1] + i[2]) * o[1] + (i[1] + i[2] + i[3]) * o[2] + (i[2] + i[3]) * o[3]
        for g in range(4):
            lin(f'conv_layers.{l}.fc.{g}.0', 3 * ns, 3 * ns)
            lin(f'conv_layers.{l}.fc.{g}.4', W, 3 * ns)
        spec[f'conv_layers.{l}.batch_norm.weight'] = (sum(o),)
        spec[f'conv_layers.{l}.batch_norm.bias'] = (o[0],)
        spec[f'conv_layers.{l}.batch_norm.running_mean'] = (o[0],)
        spec[f'conv_layers.{l}.batch_norm.running_var'] = (sum(o),)
    if confidence_mode:      # models/score_model.py:110-121: the confidence_predictor replaces the heads
        lin('confidence_predictor.0', ns, 2 * ns if num_conv_layers >= 3 else ns)
        lin('confidence_predictor.4', ns, ns)
        lin('confidence_predictor.8', num_confidence_outputs, ns)
        if not confidence_no_batchnorm:
            for i in (1, 5):
                spec.update({f'confidence_predictor.{i}.weight': (ns,), f'confidence_predictor.{i}.bias': (ns,),
                             f'confidence_predictor.{i}.running_mean': (ns,), f'confidence_predictor.{i}.running_var': (ns,),
                             f'confidence_predictor.{i}.num_batches_tracked': ()})
        return spec
    lin('center_edge_embedding.0', ns, dist + sigma)
    lin('center_edge_embedding.3', ns, ns)
    lin('final_conv.fc.0', 2 * ns, 2 * ns)
    lin('final_conv.fc.4', 2 * (2 * ns + 4 * nv), 2 * ns)
    spec.update({'final_conv.batch_norm.weight': (4,), 'final_conv.batch_norm.bias': (0,),
                 'final_conv.batch_norm.running_mean': (0,), 'final_conv.batch_norm.running_var': (4,)})
    lin('tr_final_layer.0', ns, 1 + sigma)
    lin('tr_final_layer.3', 1, ns)
    lin('rot_final_layer.0', ns, 1 + sigma)
    lin('rot_final_layer.3', 1, ns)
    lin('final_edge_embedding.0', ns, dist)
    lin('final_edge_embedding.3', ns, ns)
    lin('tor_bond_conv.fc.0', 3 * ns, 3 * ns)
    lin('tor_bond_conv.fc.4', 2 * nv * ns, 3 * ns)
    spec.update({'tor_bond_conv.batch_norm.weight': (2 * ns,), 'tor_bond_conv.batch_norm.bias': (ns,),
                 'tor_bond_conv.batch_norm.running_mean': (ns,), 'tor_bond_conv.batch_norm.running_var': (2 * ns,)})
    lin('tor_final_layer.0', ns, 2 * ns, bias=False)
    lin('tor_final_layer.3', 1, ns, bias=False)
    if ld > 0 and latent_droprate > 0:
        for k in ('lig_node', 'rec_node', 'lig_edge', 'rec_edge', 'cross_edge'):
            spec[f'{k}_unconditional_embedding'] = (1, ns)
    return spec


def random_score_model_state_dict(seed=0, latent_dim=0, latent_droprate=0.0, **spec_kw):
    """Random-init DiffDock-S (or, with latent_dim = 2 / latent_droprate = 0.1, DisCo-DiffDock-S) weights (PyTorch-default style,
    randomised BatchNorm statistics) - no checkpoints exist offline."""
    import math
    import torch
    g = torch.Generator().manual_seed(seed)
    stops = {'lig': 5.0, 'rec': 30.0, 'cross': 80.0, 'center': 30.0}
    spec = score_model_state_dict_spec(latent_dim=latent_dim, latent_droprate=latent_droprate, **spec_kw)
    P = {}
    for name, shape in spec.items():
        if name.endswith('distance_expansion.offset'):
            P[name] = torch.linspace(0.0, stops[name.split('_')[0]], shape[0])
        elif name.endswith('num_batches_tracked'):
            P[name] = torch.zeros((), dtype=torch.long)
        elif name.startswith('confidence_predictor') and name.split('.')[1] in ('1', '5'):      # BatchNorm1d with randomised statistics
            P[name] = torch.randn(shape, generator=g) * 0.1 if name.endswith(('running_mean', 'bias')) else torch.rand(shape, generator=g) + 0.5
        elif name.endswith('unconditional_embedding'):
            P[name] = torch.randn(shape, generator=g) * 0.1
        elif 'atom_embedding_list' in name:
            P[name] = (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(6.0 / (shape[0] + shape[1]))
        elif '.batch_norm.' in name:
            P[name] = torch.rand(shape, generator=g) + 0.5 if name.endswith(('running_var', 'weight')) else torch.randn(shape, generator=g) * 0.1
        else:
            fan_in = shape[1] if name.endswith('weight') else spec[name[:-4] + 'weight'][1]
            P[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
    return P


def random_ar_state_dict(seed=0, ar_ns=16, hidden=128, latent_dim=2, latent_droprate=0.1):
    """Random-init checkpoint of the AR latent model (workdir/disco_diffdockS_ar_model layout): ``pretrained_score_model.*`` (its own
    copy of the DisCo score model) + the two predictor MLPs 2*ar_ns -> hidden -> hidden -> 1 with BatchNorm1d
    (models/pretrained_score_encoder.py:24-45)."""
    import math
    import torch
    g = torch.Generator().manual_seed(seed)
    P = {'pretrained_score_model.' + k: v for k, v in random_score_model_state_dict(seed + 1, latent_dim, latent_droprate).items()}
    for name in ('latent_s_predictor', 'latent_r_predictor'):
        for i, (o, n_in) in ((0, (hidden, 2 * ar_ns)), (4, (hidden, hidden)), (8, (1, hidden))):
            P[f'{name}.{i}.weight'] = (torch.rand(o, n_in, generator=g) * 2 - 1) / math.sqrt(n_in)
            P[f'{name}.{i}.bias'] = (torch.rand(o, generator=g) * 2 - 1) / math.sqrt(n_in)
        for i in (1, 5):
            P[f'{name}.{i}.weight'] = torch.rand(hidden, generator=g) + 0.5
            P[f'{name}.{i}.bias'] = torch.randn(hidden, generator=g) * 0.1
            P[f'{name}.{i}.running_mean'] = torch.randn(hidden, generator=g) * 0.1
            P[f'{name}.{i}.running_var'] = torch.rand(hidden, generator=g) + 0.5
            P[f'{name}.{i}.num_batches_tracked'] = torch.tensor(7)
    return P


def confidence_state_dict_spec(ns=24, nv=6, num_conv_layers=5, sigma=32, dist=32, lm=1280, n_out=2):
    """name -> shape of the all-atom confidence model's state_dict (models/all_atom_score_model.py in confidence_mode as
    get_model builds it from workdir/paper_confidence_model; 430 tensors / 4 773 122 elements).  Used for random-init benches."""
    spec = {}

    def lin(name, o, i):
        spec[f'{name}.weight'] = (o, i)
        spec[f'{name}.bias'] = (o,)

    for pre, dims, has_lm in (('lig_node_embedding', LIG_FEATURE_DIMS, False), ('rec_node_embedding', REC_RESIDUE_FEATURE_DIMS, True),
                              ('atom_node_embedding', REC_ATOM_FEATURE_DIMS, False)):
        for i, d in enumerate(dims):
            spec[f'{pre}.atom_embedding_list.{i}.weight'] = (d, ns)
        lin(f'{pre}.linear', ns, sigma)
        if has_lm and lm:
            lin(f'{pre}.lm_embedding_layer', ns, lm + ns)
    lin('lig_edge_embedding.0', ns, 4 + sigma + dist)
    lin('lig_edge_embedding.3', ns, ns)
    for k in ('rec', 'atom', 'lr', 'ar', 'la'):
        lin(f'{k}_edge_embedding.0', ns, sigma + dist)
        lin(f'{k}_edge_embedding.3', ns, ns)
    for k in ('lig', 'rec', 'cross'):
        spec[f'{k}_distance_expansion.offset'] = (dist,)
    seq = [(ns, 0, 0, 0), (ns, nv, 0, 0), (ns, nv, nv, 0), (ns, nv, nv, ns)]
    for l in range(num_conv_layers):
        i_, o_ = seq[min(l, 3)], seq[min(l + 1, 3)]
        # e3nn FullyConnectedTensorProduct(in, 0e+1o+2e, out) weight count: paths (in irrep, sh l, out irrep) allowed by parity / triangle
        irr = ((0, 1), (1, -1), (1, 1), (0, -1))
        W = sum(i_[a] * o_[c] for a in range(4) for (sl, sp) in ((0, 1), (1, -1), (2, 1)) for c in range(4)
                if i_[a] and o_[c] and irr[a][1] * sp == irr[c][1] and abs(irr[a][0] - sl) <= irr[c][0] <= irr[a][0] + sl)
        nf = sum(o_)
        for k in range(9):
            lin(f'conv_layers.{9 * l + k}.fc.0', 3 * ns, 3 * ns)
            lin(f'conv_layers.{9 * l + k}.fc.3', W, 3 * ns)
            for nm, sh in (('weight', nf), ('bias', o_[0]), ('running_mean', o_[0]), ('running_var', nf)):
                spec[f'conv_layers.{9 * l + k}.batch_norm.{nm}'] = (sh,)
    lin('confidence_predictor.0', ns, 2 * ns)
    lin('confidence_predictor.4', ns, ns)
    lin('confidence_predictor.8', n_out, ns)
    for i in (1, 5):
        for k in ('weight', 'bias', 'running_mean', 'running_var'):
            spec[f'confidence_predictor.{i}.{k}'] = (ns,)
    return spec


def random_confidence_state_dict(seed=0, **kw):
    import math
    import torch
    g = torch.Generator().manual_seed(seed)
    spec, P = confidence_state_dict_spec(**kw), {}
    stops = {'lig': 5.0, 'rec': 30.0, 'cross': 80.0}
    for name, shape in spec.items():
        if name.endswith('distance_expansion.offset'):
            P[name] = torch.linspace(0.0, stops[name.split('_')[0]], shape[0])
        elif 'atom_embedding_list' in name:
            P[name] = (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(6.0 / (shape[0] + shape[1]))
        elif '.batch_norm.' in name or (name.startswith('confidence_predictor') and name.split('.')[1] in ('1', '5')):
            P[name] = torch.randn(shape, generator=g) * 0.1 if name.endswith(('running_mean', 'bias')) else torch.rand(shape, generator=g) + 0.5
        else:
            fan_in = shape[1] if name.endswith('weight') else spec[name[:-4] + 'weight'][1]
            P[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
    return P
