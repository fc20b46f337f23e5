"""Flat binary cache of complex graphs: the tensors the reference keeps in
``heterographs.pkl`` (datasets_utils/pdbbind.py:101-117 — PyG ``HeteroData`` pickles that need torch_geometric / rdkit to
load) as one little-endian file of plain arrays that a GPU box can mmap without any of those packages.  It is also the
record layout the ranks agree on for the final pose gather (fixed field order, explicit sizes).

    file   := header  index[n]  blob
    header := b'DDKG' u32 version(=2) u64 n_complexes u64 index_bytes
    index  := per complex 10 x i64: blob_offset, n_lig, n_rec, n_bond_edges, n_rot, n_rec_edges, rec_feat_dim, name_len,
              n_atom, n_atom_edges   (0, 0: no all-atom receptor level)
    blob   := per complex, each array padded to 16 B:
              name[name_len] | lig_x i32[n_lig,16] | lig_pos f32[n_lig,3] | bond_index i32[2,M] | bond_attr f32[M,4] |
              edge_mask u8[M] | mask_rotate u8[n_rot,n_lig] | rec_x f32[n_rec,F] | rec_pos f32[n_rec,3] |
              rec_edge_index i32[2,E] | original_center f32[1,3]
              and, when n_atom > 0 (graphs of the all-atom confidence model, process_mols.py:474-477):
              atom_x i32[n_atom,4] | atom_pos f32[n_atom,3] | atom_edge_index i32[2,E_aa] | atom_rec_index i32[2,n_atom]

Field meaning = the graph tensors of datasets_utils/process_mols.py; the dict returned by
:func:`load_complexes` is what ``data.from_arrays`` / ``runtime.Complex`` / ``include/ddk.h: ddk_complex_desc`` take."""
import struct

import numpy as np

MAGIC, VERSION = b'DDKG', 2
_FIELDS = [('lig_x', np.int32), ('lig_pos', np.float32), ('bond_index', np.int32), ('bond_attr', np.float32),
           ('edge_mask', np.uint8), ('mask_rotate', np.uint8), ('rec_x', np.float32), ('rec_pos', np.float32),
           ('rec_edge_index', np.int32), ('original_center', np.float32)]
_ATOM_FIELDS = [('atom_x', np.int32), ('atom_pos', np.float32), ('atom_edge_index', np.int32), ('atom_rec_index', np.int32)]


def _shapes(n_lig, n_rec, M, R, E, F, n_atom=0, E_aa=0):
    return {'atom_x': (n_atom, 4), 'atom_pos': (n_atom, 3), 'atom_edge_index': (2, E_aa), 'atom_rec_index': (2, n_atom),
            'lig_x': (n_lig, 16), 'lig_pos': (n_lig, 3), 'bond_index': (2, M), 'bond_attr': (M, 4), 'edge_mask': (M,),
            'mask_rotate': (R, n_lig), 'rec_x': (n_rec, F), 'rec_pos': (n_rec, 3), 'rec_edge_index': (2, E),
            'original_center': (1, 3)}


def _pad16(n):
    return (n + 15) // 16 * 16


def save_complexes(path, complexes):
    """complexes: iterable of dicts with the keys of ``synthetic.make_complex`` (numpy arrays or tensors)."""
    recs, blob = [], bytearray()
    for c in complexes:
        a = {k: np.ascontiguousarray(np.asarray(c[k]), dtype=dt) for k, dt in _FIELDS if k != 'original_center'}
        a['original_center'] = np.ascontiguousarray(np.asarray(c.get('original_center', np.zeros((1, 3)))), dtype=np.float32).reshape(1, 3)
        a['mask_rotate'] = a['mask_rotate'].reshape(-1, a['lig_x'].shape[0])
        n_lig, n_rec = a['lig_x'].shape[0], a['rec_x'].shape[0]
        M, R, E, F = a['bond_index'].shape[1], a['mask_rotate'].shape[0], a['rec_edge_index'].shape[1], a['rec_x'].shape[1]
        fields = list(_FIELDS)
        n_atom = E_aa = 0
        if 'atom_x' in c:
            for k, dt in _ATOM_FIELDS:
                a[k] = np.ascontiguousarray(np.asarray(c[k]), dtype=dt)
            n_atom, E_aa = a['atom_x'].shape[0], a['atom_edge_index'].shape[1]
            fields += _ATOM_FIELDS
        want = _shapes(n_lig, n_rec, M, R, E, F, n_atom, E_aa)
        for k, _ in fields:
            if tuple(a[k].shape) != want[k]:
                raise ValueError(f'graph cache: {k} has shape {a[k].shape}, expected {want[k]}')
        if int(a['edge_mask'].sum()) != R:
            raise ValueError('graph cache: edge_mask.sum() must equal the number of rotatable bonds (rows of mask_rotate)')
        name = str(c.get('name', 'complex')).encode()
        recs.append((len(blob), n_lig, n_rec, M, R, E, F, len(name), n_atom, E_aa))
        blob += name + b'\0' * (_pad16(len(name)) - len(name))
        for k, _ in fields:
            b = a[k].tobytes()
            blob += b + b'\0' * (_pad16(len(b)) - len(b))
    index = np.asarray(recs, dtype='<i8').reshape(-1, 10)
    with open(path, 'wb') as f:
        f.write(MAGIC + struct.pack('<IQQ', VERSION, len(recs), index.nbytes))
        f.write(index.tobytes())
        f.write(bytes(blob))
    return len(recs)


def load_complexes(path, mmap=True):
    """-> list of dicts of numpy arrays (views into one memory map when mmap=True: nothing is copied or unpickled)."""
    buf = np.memmap(path, dtype=np.uint8, mode='r') if mmap else np.fromfile(path, dtype=np.uint8)
    if buf.size < 24 or bytes(buf[:4]) != MAGIC:
        raise ValueError(f'{path}: not a ddk graph cache')
    version, n, index_bytes = struct.unpack('<IQQ', bytes(buf[4:24]))
    if version != VERSION:
        raise ValueError(f'{path}: graph cache version {version}, this build reads {VERSION}')
    if index_bytes != n * 80 or buf.size < 24 + index_bytes:
        raise ValueError(f'{path}: truncated graph cache')
    index = np.frombuffer(buf, dtype='<i8', count=n * 10, offset=24).reshape(n, 10)
    base = 24 + index_bytes
    out = []
    for off, n_lig, n_rec, M, R, E, F, name_len, n_atom, E_aa in index.tolist():
        p = base + off
        c = {'name': bytes(buf[p:p + name_len]).decode()}
        p += _pad16(name_len)
        shapes = _shapes(n_lig, n_rec, M, R, E, F, n_atom, E_aa)
        for k, dt in (_FIELDS + _ATOM_FIELDS if n_atom > 0 else _FIELDS):
            cnt = int(np.prod(shapes[k]))
            nbytes = cnt * np.dtype(dt).itemsize
            if p + nbytes > buf.size:
                raise ValueError(f'{path}: truncated graph cache (complex {c["name"]}, field {k})')
            c[k] = np.frombuffer(buf, dtype=dt, count=cnt, offset=p).reshape(shapes[k])
            p += _pad16(nbytes)
        c['edge_mask'] = c['edge_mask'].astype(bool)
        c['mask_rotate'] = c['mask_rotate'].astype(bool)
        out.append(c)
    return out


def complex_from_heterodata(g):
    """One graph of the reference's ``heterographs.pkl`` (datasets_utils/pdbbind.py:101-117: PyG ``HeteroData`` built by
    process_mols.py; any object with the same accessors works, including :class:`disco_diffdock_amd.data.HeteroData`) -> the
    array dict :func:`save_complexes` / ``runtime.Complex`` / ``data.from_arrays`` take.  Runs on the reference side (where
    torch_geometric can unpickle the file); nothing here imports PyG."""
    def arr(t, dt):
        if hasattr(t, 'detach'):
            t = t.detach().cpu().numpy()
        return np.ascontiguousarray(np.asarray(t), dtype=dt)
    lig, rec = g['ligand'], g['receptor']
    mr = lig.mask_rotate
    while isinstance(mr, (list, tuple)):          # batch_size=1 loaders wrap it in a list (utils/sampling.py:57)
        mr = mr[0]
    n_lig = int(arr(lig.x, np.int64).shape[0])
    name = getattr(g, 'name', 'complex')
    while isinstance(name, (list, tuple)):
        name = name[0]
    c = dict(name=str(name),
             lig_x=arr(lig.x, np.int32), lig_pos=arr(lig.pos, np.float32),
             bond_index=arr(g['ligand', 'lig_bond', 'ligand'].edge_index, np.int32),
             bond_attr=arr(g['ligand', 'lig_bond', 'ligand'].edge_attr, np.float32),
             edge_mask=arr(lig.edge_mask, bool), mask_rotate=arr(mr, bool).reshape(-1, n_lig),
             rec_x=arr(rec.x, np.float32), rec_pos=arr(rec.pos, np.float32),
             rec_edge_index=arr(g['receptor', 'rec_contact', 'receptor'].edge_index, np.int32),
             original_center=arr(getattr(g, 'original_center', np.zeros((1, 3))), np.float32).reshape(1, 3))
    order = np.argsort(c['rec_edge_index'][0], kind='stable')      # ddk_complex_create wants the receptor edges grouped by row 0
    c['rec_edge_index'] = np.ascontiguousarray(c['rec_edge_index'][:, order])
    has_atoms = 'atom' in g if hasattr(g, '__contains__') else hasattr(g, 'atom')
    if has_atoms and hasattr(g['atom'], 'x'):       # all-atom graphs of the confidence model (process_mols.py:474-477)
        c.update(atom_x=arr(g['atom'].x, np.int32), atom_pos=arr(g['atom'].pos, np.float32),
                 atom_edge_index=arr(g['atom', 'atom_contact', 'atom'].edge_index, np.int32),
                 atom_rec_index=arr(g['atom', 'atom_rec_contact', 'receptor'].edge_index, np.int32))
    return c


def convert_heterographs(pkl_path, out_path):
    """``heterographs.pkl`` (a pickled list of HeteroData, pdbbind.py:101-117) -> one DDKG file.  Needs the reference's environment
    (torch_geometric) for ``pickle.load`` only."""
    import pickle
    with open(pkl_path, 'rb') as f:
        graphs = pickle.load(f)
    return save_complexes(out_path, [complex_from_heterodata(g) for g in graphs])


if __name__ == '__main__':
    import sys
    if len(sys.argv) != 3:
        raise SystemExit('usage: python -m disco_diffdock_amd.graph_cache <heterographs.pkl> <out.ddkg>')
    print(convert_heterographs(sys.argv[1], sys.argv[2]), 'complexes written to', sys.argv[2])
