"""Host mirror of the reference's utils/sampling.py: same ``sampling(...)`` signature and return value
(sampling.py:49-54,249) and ``randomize_position`` (:12-46).  The 20-step loop itself - score model forward,
SDE perturbation, SE(3)/torsion update, Kabsch re-alignment - runs inside libddk.so (``ddk_sample``) without any
host synchronisation; this file only prepares the per-step host scalars exactly as the reference computes them.
Options outside the accelerated path (visualisation, the oracle latent encoder)
raise instead of silently doing something else."""
import collections

import numpy as np
import torch

from .data import DataLoader
from .diffusion_utils import set_time
from .runtime import h2d_async
from .score_model import complex_for_batch


def is_iterable(arr):
    try:
        iter(arr)
        return True
    except TypeError:
        return False


def randomize_position(data_list, no_torsion, no_random, tr_sigma_max, unbatched=False, ar_args=None, device=None):
    """utils/sampling.py:12-46 with the reference's signature.  The random draws come from the reference's own RNG streams in
    the reference's order; the geometry (torsion rotations, centring, random rotation, translation) runs on the GPU in one
    ``ddk_randomize_position`` launch (:func:`randomize_position_device`).  Limitation (documented in INTEGRATION.md):
    ``data_list`` must hold copies of ONE complex, which is what evaluate.py:232 passes; ``device`` defaults to the current
    CUDA device."""
    from scipy.spatial.transform import Rotation as R
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError('ddk: randomize_position runs on the GPU only (no CPU fallback)')
        device = torch.device('cuda', torch.cuda.current_device())
    randomize_position_device(data_list, no_torsion, no_random, tr_sigma_max, device)
    if ar_args is not None:   # utils/sampling.py:36-46: the pose the AR model sees
        for g in data_list:
            if ar_args.no_randomness:
                ar = torch.from_numpy(np.asarray(g['ligand'].orig_rdkit_pos[0])).float()
                center = torch.mean(ar, dim=0, keepdim=True)
                rot = torch.from_numpy(R.random().as_matrix()).float()
                g['ligand'].ar_pos = ((ar - center) @ rot.T).to(device)
            else:
                g['ligand'].ar_pos = g['ligand'].pos.clone()


def randomize_position_device(data_list, no_torsion, no_random, tr_sigma_max, device):
    """``randomize_position`` (utils/sampling.py:12-34) for the usual case that ``data_list`` holds N copies of ONE complex
    (evaluate.py:232): the random draws are taken on the host from the reference's own RNG streams in the reference's
    order (np.random.uniform per graph, then scipy ``Rotation.random()`` and ``torch.normal`` per graph), the geometry of
    all N copies runs in one ``ddk_randomize_position`` launch, and every ``g['ligand'].pos`` becomes a view of the
    resulting device tensor."""
    from scipy.spatial.transform import Rotation as R
    from .data import collate
    from .score_model import complex_for_batch
    if device is None or torch.device(device).type != 'cuda':
        raise RuntimeError('ddk: randomize_position_device needs a cuda device (no CPU fallback)')
    device = torch.device(device)
    g0, N = data_list[0], len(data_list)
    n_lig = g0['ligand'].pos.shape[0]
    if any(g['ligand'].pos.shape[0] != n_lig or getattr(g, 'name', None) != getattr(g0, 'name', None) for g in data_list):
        raise RuntimeError('ddk: randomize_position_device expects copies of one complex')
    n_rot = int(g0['ligand'].edge_mask.sum())
    tor = None
    if not no_torsion:
        tor = np.stack([np.random.uniform(low=-np.pi, high=np.pi, size=n_rot) for _ in data_list]).astype(np.float32)
    rot = np.empty((N, 3, 3), np.float32)
    tr = None if no_random else torch.empty((N, 3))
    for b in range(N):
        rot[b] = R.random().as_matrix()
        if not no_random:
            tr[b] = torch.normal(mean=0, std=tr_sigma_max, size=(1, 3))[0]
    cx, _ = complex_for_batch(collate([g0]), device, need_model=False)
    pos = cx.randomize_position(torch.as_tensor(np.asarray(g0['ligand'].pos.cpu(), np.float32)).to(device), torch.from_numpy(rot).to(device),
                                None if tor is None or n_rot == 0 else torch.from_numpy(tor).to(device),
                                None if tr is None else tr.to(device))
    for b, g in enumerate(data_list):
        g['ligand'].pos = pos[b]
    return pos


_pending = []      # bookkeeping objects whose device read-back has not been looked at yet


class _Bookkeeping:
    """The host-side results of one sampling() batch that need a device read-back: the latent bookkeeping of utils/sampling.py:205-221
    (``latent_str`` / ``latent_pos`` from the AR picks and the final poses) and the confidence model's edge-capacity flag.  The copies
    are enqueued behind the sampler into pinned memory; nothing waits for them inside sampling(), so the host goes on to the next
    complex while the GPU works.  ``resolve()`` (first access of ``d.latent_str`` / ``d.latent_pos`` on this module's graph
    container, the next sampling() call once the copies have landed, or immediately for foreign containers) fills every graph of the
    batch that is still alive (the graphs are held weakly: a batch nobody kept costs nothing).
    The capacity flag cannot be raised by the ligand-atom edge list any more (round 5: ddk_complex_set_atoms sizes it from the receptor's geometry, a bound no pose
    can exceed); it stays as a guard of the other edge groups' worst-case capacities: a batch that raised it would carry NaN confidences (conf_head_kernel),
    returned as -1000 like the reference's nan_to_num, and the warning below."""

    def __init__(self, graphs, choices, flat, len_lig, latent_dim, conf_cx):
        import weakref
        self.graphs = [weakref.ref(g) for g in graphs]
        self.name = getattr(graphs[0], 'name', None) if graphs else None
        self.len_lig, self.latent_dim = len_lig, latent_dim
        self.done = False
        self._resolving = False
        self.bound = False       # the graphs carry this object as ``_lazy`` (our HeteroData): a graph a later call re-used is skipped
        self.choices = self.flat = None
        self.conf_status = conf_cx.confidence_status_async() if conf_cx is not None else None
        if choices is not None:
            self.choices = torch.empty(choices.shape, dtype=choices.dtype, pin_memory=True)
            self.choices.copy_(choices, non_blocking=True)
            self.flat = torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
            self.flat.copy_(flat.detach(), non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def _check_conf(self):
        st, self.conf_status = self.conf_status, None
        if st is not None and int(st[19]) != 0:
            import warnings
            warnings.warn(f'ddk: ligand-atom edge capacity overflow in a confidence batch of complex {self.name!r}: its confidences were '
                          'returned as -1000 (NaN on the device)', RuntimeWarning)

    def resolve(self):
        if self.done or self._resolving:
            return
        self._resolving = True       # (done is set only when the bookkeeping below has finished: an exception part-way leaves the object retryable)
        try:
            self._resolve()
            self.done = True
        finally:
            self._resolving = False

    def _resolve(self):
        self.event.synchronize()
        if self in _pending:
            _pending.remove(self)
        self._check_conf()
        if self.choices is None:
            return
        ch, n = self.choices.tolist(), self.len_lig
        for i, ref in enumerate(self.graphs):
            d_i = ref()
            if d_i is None or (self.bound and d_i.__dict__.get('_lazy') is not self):      # collected, or re-used by a later sampling() call
                continue
            lat_str, lat_pos = "", []
            center = d_i.original_center.detach().cpu()
            for j in range(self.latent_dim):
                idx = ch[i][j]
                if not 0 <= idx < n + d_i['receptor'].pos.shape[0]:
                    raise RuntimeError(f'ddk: AR pick {idx} outside the graph of complex {self.name!r}')
                if idx < n:
                    lat_str += 'L' + str(idx)
                    lat_pos.append(self.flat[i * n + idx:i * n + idx + 1] + center)
                else:
                    idx -= n
                    lat_str += 'R' + str(idx)
                    lat_pos.append(d_i['receptor'].pos[idx:idx + 1].detach().cpu() + center)
            d_i.__dict__.pop('_lazy', None)
            d_i.latent_str = lat_str
            d_i.latent_pos = torch.cat(lat_pos, dim=0)
        self.choices = self.flat = None      # the pinned buffers go back to the allocator


def _poll_pending():
    """look (without waiting) at the read-backs of earlier calls: whatever has landed is resolved and dropped, so nothing accumulates
    when a caller never reads ``latent_str`` (warm-up calls, re-used graphs)"""
    for bk in list(_pending):
        if bk.event.query():
            bk.resolve()


def draw_noise(inference_steps, b, R_total, R, nc, device):
    """N(0,1) draws of one batch from the device generator, [steps, b, 6 + R_total] (tr xyz, rot xyz, torsions; utils/sampling.py:146-164
    draws them step by step on the host): one launch per contiguous run of steps that use noise - normally ONE for the whole trajectory.
    Steps whose noise coefficients are all zero (no_final_step_noise) draw nothing, like the reference; torsion columns past R stay zero."""
    z = torch.empty((inference_steps, b, 6 + R_total), device=device)
    active = [bool(nc[t].any()) for t in range(inference_steps)]
    t = 0
    while t < inference_steps:
        u = t
        while u < inference_steps and active[u] == active[t]:
            u += 1
        if active[t]:
            z[t:u].normal_(mean=0, std=1)
        else:
            z[t:u].zero_()
        t = u
    if R != R_total:
        z[:, :, 6 + R:] = 0
    return z


def step_coefficients(inference_steps, tr_schedule, rot_schedule, tor_schedule, t_to_sigma, model_args, ode, no_random,
                      no_final_step_noise, temp_sampling, temp_psi, temp_sigma_data):
    """Host scalars of every reverse step, formed with the reference's expressions and dtypes
    (utils/sampling.py:106-111,137-192): perturb = score_coeff*score + noise_coeff*z per tr/rot/tor."""
    if not is_iterable(temp_sampling):
        temp_sampling = [temp_sampling] * 3
    if not is_iterable(temp_psi):
        temp_psi = [temp_psi] * 3
    if not is_iterable(temp_sigma_data):
        temp_sigma_data = [temp_sigma_data] * 3
    assert len(temp_sampling) == 3 and len(temp_psi) == 3 and len(temp_sigma_data) == 3
    scheds = (tr_schedule, rot_schedule, tor_schedule)
    lims = ((model_args.tr_sigma_min, model_args.tr_sigma_max), (model_args.rot_sigma_min, model_args.rot_sigma_max),
            (model_args.tor_sigma_min, model_args.tor_sigma_max))
    t_arr = np.zeros((inference_steps, 3), np.float32)
    sc = np.zeros((inference_steps, 3), np.float32)
    nc = np.zeros((inference_steps, 3), np.float32)
    for t_idx in range(inference_steps):
        ts = [s[t_idx] for s in scheds]
        sig = t_to_sigma(*ts)
        zero = no_random or (no_final_step_noise and t_idx == inference_steps - 1)
        for k in range(3):
            s = scheds[k]
            dt = s[t_idx] - s[t_idx + 1] if t_idx < inference_steps - 1 else s[t_idx]
            lo, hi = lims[k]
            g = sig[k] * torch.sqrt(torch.tensor(2 * np.log(hi / lo)))
            if ode:
                a, b = 0.5 * g ** 2 * dt, 0.0 * g
                if temp_sampling[k] != 1.0:
                    raise RuntimeError('ode=True with temp_sampling != 1 is undefined in the reference (sampling.py:142-144 vs :182)')
            else:
                a, b = g ** 2 * dt, g * np.sqrt(dt)
            if temp_sampling[k] != 1.0:
                sd = np.exp(temp_sigma_data[k] * np.log(hi) + (1 - temp_sigma_data[k]) * np.log(lo))
                lam = (sd + sig[k]) / (sd + sig[k] / temp_sampling[k])
                a = g ** 2 * dt * (lam + temp_sampling[k] * temp_psi[k] / 2)
                b = g * np.sqrt(dt * (1 + temp_psi[k]))
            t_arr[t_idx, k] = ts[k]
            sc[t_idx, k] = float(a)
            nc[t_idx, k] = 0.0 if zero else float(b)
    return t_arr, sc, nc


def _host_single_thread(fn):
    """The host side of a call is a few ms of small tensor bookkeeping that must stay ahead of the GPU; it runs with ONE torch CPU thread:
    on a 128-thread host an OpenMP-parallel torch CPU op (a collate ``torch.cat``, a checksum) was measured to stall for a whole
    reverse-diffusion loop (70-100 ms) while the HIP runtime is busy, which starves the GPU queue (tools/host_calls.py).  The caller's
    thread setting is restored on return."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        n_thr = torch.get_num_threads()
        if n_thr > 1:
            torch.set_num_threads(1)
        try:
            return fn(*args, **kwargs)
        finally:
            if n_thr > 1:
                torch.set_num_threads(n_thr)
    return wrapper


@_host_single_thread
def sampling(data_list, model, inference_steps, tr_schedule, rot_schedule, tor_schedule, device, t_to_sigma, model_args,
             no_random=False, ode=False, visualization_list=None, confidence_model=None, confidence_data_list=None,
             confidence_model_args=None, batch_size=32, no_final_step_noise=False, use_latent=True,
             gumbel_latent_temperature=0.01, ar_model=None, ar_args=None, temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5,
             classifier_free_guidance_weight=0.0, softmax_latent_temperature=1.0, cfg_start=1.0, cfg_end=0.0,
             compute_ar_accuracy=False, noise=None):
    """``noise`` (extra, optional): list with one tensor [steps, b, 6+R] per batch of N(0,1) draws (tr xyz, rot xyz,
    torsions) to replace the device generator - used by the parity tests (the reference never seeds its RNGs)."""
    if visualization_list is not None:
        raise RuntimeError('ddk: visualisation is outside the accelerated hot path')
    confidence, confidence_loader = None, None
    if confidence_model is not None:      # utils/sampling.py:59-62
        cg_conf = getattr(confidence_model, 'score_model', confidence_model)
        if confidence_data_list is None:
            # utils/sampling.py:239-240: a coarse-grained confidence model (TensorProductScoreModel in confidence_mode, reached by evaluate.py's
            # use_original_model_cache / transfer_weights flags) evaluated on the score batch itself
            if not getattr(cg_conf, 'confidence_mode', False):
                raise RuntimeError('ddk: an all-atom confidence model needs confidence_data_list (the score graphs carry no atoms); only a '
                                   'coarse-grained confidence model (get_model(args, ..., confidence_mode=True) without all_atoms) runs on the score batch')
        else:
            cg_conf = None
        confidence_loader = iter(DataLoader(confidence_data_list, batch_size=batch_size)) if confidence_data_list is not None else None
        confidence = []
    conf_checks = []
    latent_model = use_latent and getattr(model_args, 'latent_dim', 0) > 0
    if classifier_free_guidance_weight != 0.0 and not latent_model:
        raise RuntimeError('ddk: classifier-free guidance needs the latent-conditioned model (sampling.py:119-135)')
    if latent_model and (ar_model is None or compute_ar_accuracy):
        raise RuntimeError('ddk: latent-conditioned sampling needs ar_model (the oracle encoder needs the ground-truth pose and is '
                           'outside the hot path)')
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('ddk sampling runs on the GPU only (no CPU fallback)')
    _poll_pending()
    N = len(data_list)
    loader = DataLoader(data_list, batch_size=batch_size)
    score_model = model.module.score_model if hasattr(model, 'module') else getattr(model, 'score_model', model)
    t_arr, sc, nc = step_coefficients(inference_steps, tr_schedule, rot_schedule, tor_schedule, t_to_sigma, model_args, ode,
                                      no_random, no_final_step_noise, temp_sampling, temp_psi, temp_sigma_data)
    with torch.no_grad():
        for batch_id, batch in enumerate(loader):
            b = batch.num_graphs
            if b != min(batch_size, N):
                raise RuntimeError('ragged last batch: the reference draws noise of size min(batch_size, N) (sampling.py:146-153)')
            cx, _ = complex_for_batch(batch, device, ctx=score_model.ctx)
            latent_h = None
            if latent_model:   # utils/sampling.py:69-103: AR decoding on the ar_pos pose, then the latents condition every step
                lig_st = batch['ligand']      # only the poses go to the device: the encoder's score-model copy takes everything else
                lig_st.pos = h2d_async(lig_st.pos, device)      # from the cached ddk_complex (a batch.to(device) moved 61 MB of ESM features)
                if 'ar_pos' in lig_st:
                    lig_st.ar_pos = h2d_async(lig_st.ar_pos, device)
                temp_lig_pos = batch['ligand'].pos
                if 'ar_pos' in batch['ligand']:
                    batch['ligand'].pos = batch['ligand'].ar_pos
                latent_h = ar_model.encode_ar(batch, softmax_latent_temperature)
                batch['ligand'].pos = temp_lig_pos
                batch['ligand'].latent_h, batch['receptor'].latent_h = latent_h
                cx.set_latents(latent_h[0], latent_h[1], 0.0)
                choices = ar_model.last_choices          # [b, latent_dim] picked nodes (device): read back ONCE, after the sampler is enqueued
                cx.set_guidance(classifier_free_guidance_weight, cfg_start, cfg_end)
            elif score_model.cfg['latent_dim'] > 0:
                raise RuntimeError('ddk: a latent-conditioned score model was given but use_latent / model_args.latent_dim disable the latents')
            pos = h2d_async(batch['ligand'].pos.float(), device).reshape(b, -1, 3).contiguous()
            if pos.data_ptr() == batch['ligand'].pos.data_ptr():
                pos = pos.clone()       # ddk_sample updates in place; the caller's start poses stay intact (the reference rebinds, never mutates)
            R = cx.R if not model_args.no_torsion else 0
            if noise is not None:
                z = noise[batch_id].to(device)
            elif no_random or ode:
                z = None
            else:
                z = draw_noise(inference_steps, b, cx.R, R, nc, device)
            cx.sample(pos, t_arr, sc, nc, z)
            if confidence_model is not None and cg_conf is not None:
                # utils/sampling.py:239-240: the score batch itself at the final poses; its times are those of the LAST EXECUTED step
                # (t_idx = inference_steps - 1, utils/sampling.py:105-111: the reference resets them only in the confidence_data_list branch),
                # which confidence_mode reads as sigmas.  evaluate.py:269 passes the full schedule with inference_steps = actual_steps, so this
                # is schedule[inference_steps - 1], not schedule[-1]
                last = inference_steps - 1
                out = cg_conf.confidence(batch, pos, (float(tr_schedule[last]), float(rot_schedule[last]), float(tor_schedule[last])))
                confidence.append(out)
            elif confidence_model is not None:   # utils/sampling.py:230-243: final poses into the all-atom graphs, t = 0
                cbatch = next(confidence_loader)
                cbatch['ligand'].pos = pos.reshape(-1, 3)
                set_time(cbatch, 0, 0, 0, b, confidence_model_args.all_atoms if confidence_model_args is not None else True, device)
                from .confidence import ConfidenceModel
                if isinstance(confidence_model, ConfidenceModel):
                    out = confidence_model(cbatch, check=False)       # the capacity flag is read once per call, below
                    conf_checks.append(confidence_model.last_complex)
                else:
                    out = confidence_model(cbatch)
                confidence.append(out[0] if type(out) is tuple else out)
            len_lig = pos.shape[1]
            flat = pos.reshape(-1, 3)
            graphs = [data_list[batch_id * batch_size + i] for i in range(b)]
            for i, d_i in enumerate(graphs):
                d_i['ligand'].pos = flat[i * len_lig:len_lig * (i + 1)]
            if latent_model or conf_checks:
                # latent bookkeeping of utils/sampling.py:205-221 and the confidence model's capacity flag: ONE read-back, enqueued behind the
                # sampler and not awaited here (the reference synchronises 6x per pose)
                from .data import HeteroData
                bk = _Bookkeeping(graphs, choices if latent_model else None, flat, len_lig, getattr(model_args, 'latent_dim', 0),
                                  conf_checks.pop() if conf_checks else None)
                if latent_model and all(isinstance(d_i, HeteroData) for d_i in graphs):
                    for d_i in graphs:
                        d_i.__dict__.pop('latent_str', None)
                        d_i.__dict__.pop('latent_pos', None)
                        d_i.__dict__['_lazy'] = bk
                    bk.bound = True
                    _pending.append(bk)
                elif latent_model:
                    bk.resolve()              # foreign graph containers (PyG) cannot fill attributes lazily: wait now
                else:
                    _pending.append(bk)
    if confidence_model is not None:          # utils/sampling.py:245-247
        confidence = torch.nan_to_num(torch.cat(confidence, dim=0), nan=-1000)
    return data_list, confidence
