"""CPU tests of the host-side logic above the C ABI: step coefficients vs the oracle, graph containers,
config mapping / loud failures, and the multi-process sharding + gather (gloo, world_size 2)."""
import os
import sys
from argparse import Namespace
from functools import partial

import numpy as np
import pytest
import torch

from oracle import score_model_ref as smr
from oracle import sampler_ref as spr

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
ARGS = Namespace(tr_sigma_min=0.1, tr_sigma_max=19.0, rot_sigma_min=0.03, rot_sigma_max=1.55, tor_sigma_min=0.03,
                 tor_sigma_max=3.14, no_torsion=False)
README_S = dict(temp_sampling=[1.886430780895051, 5.659562317960644, 2.8888668488630156],
                temp_psi=[0.07085125444659945, 2.686505606141324, 4.089493860493927],
                temp_sigma_data=[0.3617563913086843, 0.7437588205919711, 0.08897393057297842])


@pytest.mark.parametrize('kw', [{}, README_S, dict(ode=True)])
def test_step_coefficients_match_oracle(kw):
    from disco_diffdock_amd.sampling import step_coefficients
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    steps = 20
    sched = get_t_schedule(steps)
    assert np.array_equal(sched, spr.get_t_schedule(steps))
    kw = dict(kw)
    ode = kw.pop('ode', False)
    t_arr, sc, nc = step_coefficients(steps, sched, sched, sched, partial(t_to_sigma, args=ARGS), ARGS, ode, False, True,
                                      kw.get('temp_sampling', 1.0), kw.get('temp_psi', 0.0), kw.get('temp_sigma_data', 0.5))
    cfg = smr.ScoreModelConfig()
    for k in range(steps):
        ref = spr.sde_step_coefficients(k, steps, (sched, sched, sched), cfg, ode, kw.get('temp_sampling', 1.0),
                                        kw.get('temp_psi', 0.0), kw.get('temp_sigma_data', 0.5))
        for j in range(3):
            assert abs(sc[k, j] - float(ref[j][1])) <= 1e-6 * abs(float(ref[j][1]))
            if k < steps - 1:
                assert abs(nc[k, j] - float(ref[j][2])) <= 1e-6 * abs(float(ref[j][2])) + 1e-12
            else:
                assert nc[k, j] == 0.0     # no_final_step_noise


def test_collate_matches_oracle_container():
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.data import from_arrays, collate
    from helpers import batch_of
    c = synthetic.make_complex(1, n_res=12, n_lig=20, esm_dim=8)
    a = collate([from_arrays(c) for _ in range(3)])
    b = batch_of(c, 3)
    assert a.num_graphs == 3
    for nt in ('ligand', 'receptor'):
        assert torch.equal(a[nt].batch, b[nt].batch) and torch.equal(a[nt].pos, b[nt].pos)
    for et in (('ligand', 'ligand'), ('receptor', 'receptor')):
        assert torch.equal(a[et].edge_index, b[et].edge_index)


def test_get_model_refuses_unsupported_configs():
    from disco_diffdock_amd.model_utils import get_model
    base = dict(ns=24, nv=6, num_conv_layers=5, sigma_embed_dim=32, distance_embed_dim=32, cross_distance_embed_dim=32,
                max_radius=5.0, cross_max_distance=80, dynamic_max_cross=True, embedding_scale=1000, embedding_type='sinusoidal',
                scale_by_sigma=True, no_torsion=False, no_batch_norm=False, dropout=0.1, sh_lmax=1, use_second_order_repr=False,
                use_old_atom_encoder=False, esm_embeddings_path='x',
                **{k: v for k, v in vars(ARGS).items() if k != 'no_torsion'})
    with pytest.raises(RuntimeError, match='latent'):
        get_model(Namespace(**dict(base, latent_dim=2, latent_vocab=64)), torch.device('cpu'), None)
    with pytest.raises(RuntimeError, match='sh_lmax=1'):
        get_model(Namespace(**dict(base, sh_lmax=2)), torch.device('cpu'), None)
    with pytest.raises(RuntimeError, match='all-atom'):
        get_model(Namespace(**dict(base, all_atoms=True)), torch.device('cpu'), None)


def test_shard_assignment_is_a_balanced_partition():
    from disco_diffdock_amd.distributed import shard_indices
    costs = [300 * 30, 2000 * 40, 100 * 20, 500 * 25, 800 * 33, 120 * 22, 640 * 28, 50 * 20, 900 * 31, 310 * 30, 15 * 20]
    for world in (1, 2, 4, 8):
        parts = [shard_indices(costs, r, world) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(len(costs)))
        loads = [sum(costs[i] for i in p) for p in parts]
        if world <= 4:
            assert max(loads) <= 1.6 * (sum(costs) / world) + max(costs)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from disco_diffdock_amd.distributed import shard_indices, gather_poses
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n_lig = [20, 33, 25, 40, 22]
    S = 3
    mine = shard_indices([n * 100 for n in n_lig], rank, world)
    poses = {i: torch.full((S, n_lig[i], 3), float(i)) + torch.arange(S).reshape(S, 1, 1) for i in mine}
    full = gather_poses(poses, n_lig, S, device=torch.device('cpu'))
    # sample-level sharding of one complex (large-pocket layout): 7 poses over the ranks
    from disco_diffdock_amd.distributed import shard_samples, gather_samples
    lo, hi = shard_samples(7, rank, world)
    whole = torch.arange(7 * 4 * 3, dtype=torch.float32).reshape(7, 4, 3)
    got = gather_samples(whole[lo:hi], 7, rank, world, torch.device('cpu'))
    from disco_diffdock_amd.distributed import gather_confidences
    conf = gather_confidences({i: torch.tensor([[i + 0.5 * s, -i] for s in range(S)]) for i in mine}, len(n_lig), torch.device('cpu'))
    conf1 = gather_confidences({i: torch.arange(S) * 1.0 + 10 * i for i in mine}, len(n_lig), torch.device('cpu'))
    okc = all(torch.equal(conf[i], torch.tensor([[i + 0.5 * s, -i] for s in range(S)])) and torch.equal(conf1[i], torch.arange(S) * 1.0 + 10 * i)
              for i in range(len(n_lig)))
    if rank == 0:
        assert okc
        ok = all(torch.equal(full[i], torch.full((S, n_lig[i], 3), float(i)) + torch.arange(S).reshape(S, 1, 1)) for i in range(len(n_lig)))
        ok = ok and torch.equal(got, whole)
        open(os.path.join(tmp, 'ok'), 'w').write(str(ok))
    dist.destroy_process_group()


def test_shard_samples_partition():
    from disco_diffdock_amd.distributed import shard_samples
    for n, world in ((40, 8), (7, 2), (3, 8), (1, 1)):
        parts = [shard_samples(n, r, world) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == n and all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
        assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1


def test_gloo_world2_shard_and_gather(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / 'ok').read() == 'True'


def _worker_empty_rank(rank, world, port, tmp):
    """a rank that owns NO complex (more ranks than complexes): it still takes part in every gather and receives everything"""
    import torch.distributed as dist
    from disco_diffdock_amd.distributed import shard_indices, gather_poses, gather_confidences
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n_lig, S = [9], 2
    mine = shard_indices([n * 300 for n in n_lig], rank, world)
    assert mine == ([0] if rank == 0 else [])
    full = gather_poses({i: torch.full((S, n_lig[i], 3), 7.0) for i in mine}, n_lig, S, device=torch.device('cpu'))
    conf = gather_confidences({i: torch.tensor([1.5, -2.0]) for i in mine}, 1, torch.device('cpu'))
    ok = torch.equal(full[0], torch.full((S, 9, 3), 7.0)) and torch.equal(conf[0], torch.tensor([1.5, -2.0]))
    open(os.path.join(tmp, f'ok{rank}'), 'w').write(str(ok))
    dist.destroy_process_group()


def test_gloo_gather_with_a_rank_that_owns_nothing(tmp_path):
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_empty_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / 'ok0').read() == 'True' and open(tmp_path / 'ok1').read() == 'True'


def test_shard_indices_balances_the_timesplit_sized_set():
    """VERDICT r04 #5b / r05 #2: the LPT partition of bench.py --config 4 --complexes 363's cost vector - round 6: the timesplit-SHAPED set
    (synthetic.timesplit_shape: log-normal receptor sizes in [60, 3000] residues around a median of 350, 10-80-atom ligands; distributed.complex_cost)
    - over 8 ranks: max / mean load <= 1.05, every complex owned exactly once, the same answer on every rank.  The heavy tail is what makes this a test:
    the largest receptor costs ~4.2 x the median complex."""
    from disco_diffdock_amd.distributed import shard_indices, complex_cost
    from disco_diffdock_amd import synthetic
    shapes = [synthetic.timesplit_shape(i) for i in range(363)]
    n_res = np.array([r for r, _ in shapes])
    assert n_res.min() >= 60 and n_res.max() <= 3000 and 300 <= np.median(n_res) <= 400 and (n_res > 1000).sum() >= 5 and (n_res < 150).sum() >= 20
    costs = [complex_cost(r, l) for r, l in shapes]
    assert max(costs) / float(np.median(costs)) > 3.0
    parts = [shard_indices(costs, r, 8) for r in range(8)]
    assert sorted(i for p in parts for i in p) == list(range(363))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) / (sum(loads) / 8) <= 1.05, loads
    assert parts == [shard_indices(costs, r, 8) for r in range(8)]
    # round 5's set (every receptor at 300 residues: bench.py --fixed-receptor) still balances
    costs5 = [complex_cost(300, l) for _, l in shapes]
    loads5 = [sum(costs5[i] for i in shard_indices(costs5, r, 8)) for r in range(8)]
    assert max(loads5) / (sum(loads5) / 8) <= 1.05


def test_product_state_dict_spec_equals_reference_layout():
    from disco_diffdock_amd import synthetic
    a = {k: tuple(v) for k, v in synthetic.score_model_state_dict_spec().items()}
    b = {k: tuple(v) for k, v in smr.state_dict_spec(smr.ScoreModelConfig(latent_vocab=64)).items()}
    assert a == b and len(a) == 171 and sum(int(np.prod(v)) for v in a.values()) == 2107134


def test_graph_cache_roundtrip(tmp_path):
    """Flat graph cache: ragged complexes survive save -> mmap load bit-exactly; corrupt files are refused."""
    from disco_diffdock_amd import graph_cache, synthetic
    from disco_diffdock_amd.data import from_arrays
    cs = [synthetic.make_complex(s, n_res=n, esm_dim=8) for s, n in ((0, 40), (1, 17), (2, 64))]
    synthetic.add_receptor_atoms(cs[1], np.random.default_rng(1))        # one complex carries the all-atom level of the confidence graphs
    path = tmp_path / 'graphs.ddkg'
    assert graph_cache.save_complexes(path, cs) == 3
    back = graph_cache.load_complexes(path)
    assert [c['name'] for c in back] == [c['name'] for c in cs]
    for a, b in zip(cs, back):
        for k in ('lig_x', 'lig_pos', 'bond_index', 'bond_attr', 'edge_mask', 'mask_rotate', 'rec_x', 'rec_pos', 'rec_edge_index', 'original_center'):
            assert np.array_equal(np.asarray(a[k]).reshape(np.asarray(b[k]).shape), b[k]), k
        for k in ('atom_x', 'atom_pos', 'atom_edge_index', 'atom_rec_index'):
            assert (k in a) == (k in b) and (k not in a or np.array_equal(np.asarray(a[k]), b[k])), k
        g = from_arrays(b)
        assert ('atom' in g) == ('atom_x' in a)
        assert g['ligand'].pos.shape == (a['lig_pos'].shape[0], 3) and g['receptor'].x.shape == a['rec_x'].shape
    raw = path.read_bytes()
    (tmp_path / 'bad_magic').write_bytes(b'XXXX' + raw[4:])
    (tmp_path / 'short').write_bytes(raw[:len(raw) // 2])
    for name in ('bad_magic', 'short'):
        with pytest.raises(ValueError):
            graph_cache.load_complexes(tmp_path / name)
    bad = dict(cs[0]); bad['rec_pos'] = bad['rec_pos'][:-1]
    with pytest.raises(ValueError, match='rec_pos'):
        graph_cache.save_complexes(tmp_path / 'x', [bad])


def test_product_confidence_spec_equals_reference_layout():
    """The product-side random-init layout of the confidence model == the oracle's (pinned by the reference's strict load)."""
    from disco_diffdock_amd import synthetic
    from oracle import confidence_ref as cr
    a = {k: tuple(v) for k, v in synthetic.confidence_state_dict_spec().items()}
    b = {k: tuple(v) for k, v in cr.state_dict_spec(cr.ConfidenceModelConfig()).items()}
    assert a == b and len(a) == 430 and sum(int(np.prod(v)) for v in a.values()) == 4773122


def test_collate_accepts_pyg_style_graphs():
    """The reference hands torch_geometric HeteroData to sampling(): stores keep their tensors in a mapping (not in __dict__), edge
    types are 3-tuples.  A minimal stand-in with exactly those accessors must collate to the same batch as our own container."""
    import numpy as np
    import torch
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.data import from_arrays, collate

    class Store:
        def __init__(self, **kw):
            object.__setattr__(self, '_mapping', dict(kw))

        def keys(self):
            return list(self._mapping)

        def __getattr__(self, k):
            try:
                return object.__getattribute__(self, '_mapping')[k]
            except KeyError:
                raise AttributeError(k)

        def __contains__(self, k):
            return k in self._mapping

        @property
        def num_nodes(self):
            return self._mapping['pos'].shape[0]

        @property
        def num_edges(self):
            return self._mapping['edge_index'].shape[1]

    class PygLike:
        def __init__(self, g):
            self._node = {nt: Store(**{k: getattr(g[nt], k) for k in g[nt].keys()}) for nt in g.node_types}
            names = {('ligand', 'ligand'): 'lig_bond', ('receptor', 'receptor'): 'rec_contact'}
            self._edge = {(et[0], names[et], et[1]): Store(**{k: getattr(g[et], k) for k in g[et].keys()}) for et in g.edge_types}

        node_types = property(lambda self: list(self._node))
        edge_types = property(lambda self: list(self._edge))

        def __getitem__(self, key):
            if isinstance(key, tuple):
                for et, st in self._edge.items():
                    if (et[0], et[-1]) == (key[0], key[-1]):
                        return st
                raise KeyError(key)
            return self._node[key]

    c = synthetic.make_complex(5, n_res=30, n_lig=20)
    own = [from_arrays(c) for _ in range(3)]
    for i, g in enumerate(own):
        g['ligand'].pos = g['ligand'].pos + float(i)
    a, b = collate(own), collate([PygLike(g) for g in own])
    assert a.num_graphs == b.num_graphs == 3
    for nt, k in (('ligand', 'pos'), ('ligand', 'x'), ('ligand', 'batch'), ('receptor', 'pos'), ('receptor', 'batch')):
        assert torch.equal(getattr(a[nt], k), getattr(b[nt], k)), (nt, k)
    for et in (('ligand', 'ligand'), ('receptor', 'receptor')):
        assert torch.equal(a[et].edge_index, b[et].edge_index) and a[et].num_edges == b[et].num_edges
    assert a['ligand'].num_nodes == b['ligand'].num_nodes == 3 * len(c['lig_pos'])


def _reversed_ligand(c):
    """the same ligand with its atoms listed in reverse order: every feature SUM is unchanged, topology arrays are not"""
    n = c['lig_x'].shape[0]
    perm = np.arange(n)[::-1].copy()
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    d = dict(c)
    d['lig_x'] = c['lig_x'][perm]
    d['lig_pos'] = c['lig_pos'][perm]
    d['bond_index'] = inv[c['bond_index']]
    d['mask_rotate'] = c['mask_rotate'][:, perm]
    return d


def test_complex_cache_key_is_a_content_hash():
    """the process-global Complex cache key must tell two same-composition ligands on one receptor apart
    (reversed atom order, changed bond types, a mask_rotate override) - sums of features collide."""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.data import collate, from_arrays
    from disco_diffdock_amd.score_model import _fingerprint
    c = synthetic.make_complex(3, n_res=40, n_lig=20)
    key = lambda cc, **kw: _fingerprint(collate([from_arrays(cc) for _ in range(2)]), 2, **kw)
    k0 = key(c)
    assert k0 == key(dict(c))                                        # deterministic
    assert k0 != key(_reversed_ligand(c))                            # equal sums, different atom order / bond_index / mask_rotate
    c2 = dict(c); c2['bond_attr'] = np.roll(c['bond_attr'], 1, axis=1)
    assert k0 != key(c2)
    c3 = dict(c); c3['rec_x'] = c['rec_x'].copy(); c3['rec_x'][5, 700] += 1.0     # ESM feature outside the hashed slice: the checksum sees it
    assert k0 != key(c3)
    mr = np.array(c['mask_rotate'], dtype=bool).copy()
    if mr.size:
        mr[0, 0] = not mr[0, 0]
        assert k0 != key(c, mask_rotate=mr)


@pytest.mark.parametrize('lat', [dict(), dict(latent_dim=2, latent_vocab=1, latent_droprate=0.1)])
def test_strict_state_dict_key_set(lat):
    """load_state_dict(strict=True) validates against the reference key set (DiffDock-S: 171 tensors, DisCo-S: 176)."""
    from types import SimpleNamespace
    from oracle import score_model_ref as smr
    from disco_diffdock_amd.score_model import TensorProductScoreModel
    from disco_diffdock_amd.runtime import DEFAULTS
    cfg = dict(DEFAULTS, **lat)
    spec = TensorProductScoreModel.expected_state_dict_spec(SimpleNamespace(cfg=cfg))
    ref = smr.state_dict_spec(smr.ScoreModelConfig(**(lat or dict(latent_vocab=64))))
    assert {k: tuple(v) for k, v in spec.items()} == {k: tuple(v) for k, v in ref.items()}
    assert len(spec) == (176 if lat else 171)


def test_heterographs_pickle_to_graph_cache(tmp_path):
    """f4: the converter a maintainer runs on the reference side - a pickled list of HeteroData-like graphs
    (datasets_utils/pdbbind.py:101-117) -> DDKG file -> the arrays the device path consumes; score-only and all-atom graphs."""
    import pickle
    from disco_diffdock_amd import synthetic, graph_cache
    from disco_diffdock_amd.data import from_arrays
    cs = [synthetic.make_complex(5, n_res=30, n_lig=17), synthetic.make_complex(6, n_res=25, n_lig=12)]
    synthetic.add_receptor_atoms(cs[1], np.random.default_rng(1))
    graphs = [from_arrays(c) for c in cs]
    pkl, out = tmp_path / 'heterographs.pkl', tmp_path / 'graphs.ddkg'
    with open(pkl, 'wb') as f:
        pickle.dump(graphs, f)
    assert graph_cache.convert_heterographs(str(pkl), str(out)) == 2
    back = graph_cache.load_complexes(str(out))
    for c, r in zip(cs, back):
        keys = [k for k, _ in graph_cache._FIELDS] + ([k for k, _ in graph_cache._ATOM_FIELDS] if 'atom_x' in c else [])
        assert ('atom_x' in r) == ('atom_x' in c)
        for k in keys:
            want = np.asarray(c[k])
            assert np.array_equal(np.asarray(r[k]).astype(want.dtype if want.dtype != np.int64 else np.int32), want.astype(r[k].dtype)), k


def test_state_dict_check_ignores_e3nn_internal_keys():
    """a real checkpoint also carries e3nn's own entries - ``conv_layers.*.tp.output_mask`` / ``.tp.weight`` ([0] with
    shared_weights=False), ``final_tp_tor.weight`` / ``final_tp_tor.output_mask`` of the weightless FullTensorProduct in the torsion head
    (models/score_model.py:152) and ``_compiled_*`` code-generator modules.  They must not count as unexpected under strict=True, while a
    genuinely foreign key, a missing key and a mis-shaped tensor still raise."""
    from types import SimpleNamespace
    from disco_diffdock_amd.score_model import TensorProductScoreModel, check_state_dict
    from disco_diffdock_amd.runtime import DEFAULTS
    spec = TensorProductScoreModel.expected_state_dict_spec(SimpleNamespace(cfg=dict(DEFAULTS)))
    sd = {k: torch.zeros(tuple(v)) for k, v in spec.items()}
    extra = {'final_tp_tor.weight': torch.zeros(0), 'final_tp_tor.output_mask': torch.ones(9),
             'conv_layers.0.tp.weight': torch.zeros(0), 'conv_layers.3.tp.output_mask': torch.ones(84),
             'tor_bond_conv.tp.output_mask': torch.ones(48), 'final_conv.tp._compiled_main_left_right.foo': torch.zeros(1),
             'final_tp_tor._compiled_main_left_right.bar': torch.zeros(1)}
    have, missing, unexpected = check_state_dict(spec, dict(sd, **extra), strict=True)
    assert set(have) == set(spec) and not missing and not unexpected
    with pytest.raises(RuntimeError, match='unexpected keys'):
        check_state_dict(spec, dict(sd, **{'encoder.weight': torch.zeros(3)}), strict=True)
    k0 = next(iter(spec))
    with pytest.raises(RuntimeError, match='missing'):
        check_state_dict(spec, {k: v for k, v in sd.items() if k != k0}, strict=False)
    with pytest.raises(RuntimeError, match='size mismatch'):
        check_state_dict(spec, dict(sd, **{k0: torch.zeros(tuple(spec[k0]) + (2,))}), strict=True)


def test_graph_copies_settle_pending_bookkeeping():
    """a graph with results pending from sampling() (``_lazy``) settles them before a copy / deep copy / pickle, so the
    copy carries values and never the bookkeeping object (which owns a CUDA event); graphs are held weakly by the bookkeeping."""
    import copy
    import pickle
    from disco_diffdock_amd.data import HeteroData

    class FakeBk:
        def __init__(self, g):
            self.g, self.calls = g, 0

        def resolve(self):
            self.calls += 1
            self.g.__dict__.pop('_lazy', None)
            self.g.latent_str = 'L1R2'

    for op in (copy.copy, copy.deepcopy, lambda g: pickle.loads(pickle.dumps(g))):
        g = HeteroData()
        g['ligand'].pos = torch.zeros(4, 3)
        g.name = 'c'
        bk = FakeBk(g)
        g.__dict__['_lazy'] = bk
        h = op(g)
        assert bk.calls == 1 and '_lazy' not in g.__dict__ and '_lazy' not in h.__dict__
        assert h.latent_str == 'L1R2' and h.name == 'c' and tuple(h['ligand'].pos.shape) == (4, 3)


def test_bench_quotes_pmc_traffic_only_for_the_profiled_kernel(tmp_path, monkeypatch):
    """roofline.traffic is a constant read from profiles/; it must turn to null (with the reason) when the kernel's sources
    differ from the ones the profile was taken on."""
    import importlib.util, json as _json
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    prof = tmp_path / 'profiles'
    prof.mkdir()
    sha = b.conv_kernel_source_sha()
    (prof / 'r09_pmc_traffic.json').write_text(_json.dumps({'kernel_source_sha256': sha, 'traffic_bytes_per_launch': 123.0}))
    monkeypatch.setattr(b, 'ROOT', str(tmp_path))
    monkeypatch.setattr(b, 'conv_kernel_source_sha', lambda: sha)
    assert b.pmc_traffic() == (123.0, 'r09_pmc_traffic.json')
    monkeypatch.setattr(b, 'conv_kernel_source_sha', lambda: 'f' * 64)          # "a deliberate edit to k_conv_x.hip without re-profiling"
    t, why = b.pmc_traffic()
    assert t is None and 'other kernel sources' in why
    (prof / 'r09_pmc_traffic.json').write_text(_json.dumps({'traffic_bytes_per_launch': 123.0}))      # a profile without a stamp (rounds 1-3)
    assert b.pmc_traffic()[0] is None


def test_build_stamp_covers_the_public_headers_and_the_product_has_no_variant_kernel():
    """include/ddk.h and ddk_debug.h had dropped out of every object's content stamp (a change of ddk_config would not have
    recompiled anything); VERDICT r05 #7: round 5's opt-in conv kernel lives under tools/variants/ and is not a source of libddk.so."""
    from disco_diffdock_amd import build
    hs = [os.path.normpath(h) for h in build.stamp_headers()]
    for name in ('ddk.h', 'ddk_debug.h'):
        assert os.path.normpath(os.path.join(ROOT, 'include', name)) in hs, name
    assert any(h.endswith('k_conv_x_epi_gen.inc') for h in hs) and all(os.path.exists(h) for h in hs)
    assert 'k_conv_y.hip' not in build.SOURCES and not os.path.exists(os.path.join(ROOT, 'disco_diffdock_amd', 'csrc', 'k_conv_y.hip'))
    assert os.path.exists(os.path.join(ROOT, 'tools', 'variants', 'k_conv_y.hip'))
    src = open(os.path.join(ROOT, 'disco_diffdock_amd', 'csrc', 'k_conv_x.hip')).read()
    assert '#error "X3_ABL_* switches give WRONG RESULTS' in src      # a stray -DX3_ABL_* cannot produce a wrong-answer product library
